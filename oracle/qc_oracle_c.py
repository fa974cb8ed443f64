"""ctypes loader of oracle/libqc_oracle.so (the C restatement; test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libqc_oracle.so")
_dp = C.POINTER(C.c_double)


class qco_problem(C.Structure):
    _fields_ = [("N", C.c_int), ("m", C.c_int), ("T", C.c_longlong), ("zdim", C.c_int), ("off_U", C.c_int),
                ("off_a", C.c_int), ("off_dt", C.c_int), ("dt_fixed", C.c_double), ("integrator", C.c_int),
                ("order", C.c_int), ("n_deriv", C.c_int), ("x_off", C.c_int * 8), ("dx_off", C.c_int * 8),
                ("ddim", C.c_int * 8), ("G_drift", _dp), ("G_drives", _dp), ("ncol", C.c_int)]


def load(build: bool = True) -> C.CDLL:
    if not os.path.exists(LIB):
        if not build:
            raise FileNotFoundError(LIB)
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lib = C.CDLL(LIB)
    lib.qco_eval_F_jac.restype = C.c_int
    lib.qco_eval_F_jac.argtypes = [C.POINTER(qco_problem), _dp, _dp, _dp, C.c_longlong, C.c_longlong, C.c_int]
    lib.qco_eval_hess.restype = C.c_int
    lib.qco_eval_hess.argtypes = [C.POINTER(qco_problem), _dp, _dp, _dp, C.c_longlong, C.c_longlong, C.c_int]
    for f in ("qco_ddim", "qco_jac_nnz", "qco_hess_nnz"):
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [C.POINTER(qco_problem)]
    lib.qco_max_threads.restype = C.c_int
    return lib


class COracle:
    """Same Problem object as oracle/qc_oracle.py; evaluates with the C library."""

    def __init__(self, prob, threads: int = 0):
        self.lib = load()
        self.prob = prob
        self.threads = threads
        p = qco_problem()
        p.N, p.m, p.T, p.zdim = prob.N, prob.m, prob.T, prob.zdim
        p.off_U, p.off_a, p.off_dt, p.dt_fixed = prob.off_U, prob.off_a, prob.off_dt, prob.dt_fixed
        p.integrator, p.order, p.n_deriv = prob.integrator, prob.order, len(prob.derivs)
        for i, d in enumerate(prob.derivs):
            p.x_off[i], p.dx_off[i], p.ddim[i] = d.x_off, d.dx_off, d.dim
        self._G0 = np.asfortranarray(prob.G_drift, dtype=np.float64)
        self._Gd = np.ascontiguousarray(np.stack([g.reshape(-1, order="F") for g in prob.G_drives])) if prob.m else np.zeros((1, 1))
        p.G_drift = self._G0.ctypes.data_as(_dp)
        p.G_drives = self._Gd.ctypes.data_as(_dp)
        p.ncol = getattr(prob, "ncol", 0)
        self.p = p
        self.ddim = self.lib.qco_ddim(C.byref(p))
        self.jac_nnz = self.lib.qco_jac_nnz(C.byref(p))
        self.hess_nnz = self.lib.qco_hess_nnz(C.byref(p))      # own values per interval (the C code knows no padding)
        a = max(1, getattr(prob, "hess_align", 1))
        self.hess_pad = (-self.hess_nnz) % a if self.hess_nnz else 0
        self.row_stride = getattr(prob, "row_stride", self.ddim)
        self.row_map = np.asarray(getattr(prob, "row_map", np.arange(self.ddim)))
        self._plain_rows = self.row_stride == self.ddim and np.array_equal(self.row_map, np.arange(self.ddim))

    def F_dF(self, Z, t_begin=0, t_end=None, want_F=True, want_J=True):
        t_end = self.prob.T - 1 if t_end is None else t_end
        k = t_end - t_begin
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        F = np.empty(k * self.ddim) if want_F else None
        J = np.empty(k * self.jac_nnz) if want_J else None
        rc = self.lib.qco_eval_F_jac(C.byref(self.p), Z.ctypes.data_as(_dp), F.ctypes.data_as(_dp) if want_F else None,
                                     J.ctypes.data_as(_dp) if want_J else None, t_begin, t_end, self.threads)
        assert rc == 0, rc
        if want_F and not self._plain_rows:       # rows at their state component's position (Problem.row_map)
            Fs = np.zeros((k, self.row_stride))
            Fs[:, self.row_map] = F.reshape(k, self.ddim)
            F = Fs.reshape(-1)
        return F, J

    def mu_d2F(self, Z, mu, t_begin=0, t_end=None):
        t_end = self.prob.T - 1 if t_end is None else t_end
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        if not self._plain_rows:
            mu = np.ascontiguousarray(mu.reshape(-1, self.row_stride)[:, self.row_map]).reshape(-1)
        H = np.empty((t_end - t_begin) * self.hess_nnz)
        rc = self.lib.qco_eval_hess(C.byref(self.p), Z.ctypes.data_as(_dp), mu.ctypes.data_as(_dp), H.ctypes.data_as(_dp),
                                    t_begin, t_end, self.threads)
        assert rc == 0, rc
        if self.hess_pad:                          # explicit zeros after each interval's values (Problem.hess_align)
            Hp = np.zeros((t_end - t_begin, self.hess_nnz + self.hess_pad))
            Hp[:, :self.hess_nnz] = H.reshape(t_end - t_begin, self.hess_nnz)
            H = Hp.reshape(-1)
        return H
