"""Final-knot fidelity terms ("next" row 1), evaluated on the GPU through
`qc_fidelity_*` (include/qcolloc.h):

    iso_vec_unitary_fidelity(U_T, U_G; subspace)                      unitary_minimum_time_problem.jl:77
    UnitaryInfidelityObjective(state_name, traj, Q; subspace)         unitary_smooth_pulse_problem.jl:133-137
    FinalUnitaryFidelityConstraint(state_name, val, traj; subspace)   unitary_minimum_time_problem.jl:80-84

Loss per the reference docstring (unitary_smooth_pulse_problem.jl:23-28): l = |1 - |tr(U_goal' U_T)| / N|.
Only the last knot's state enters; gradients/Hessians are returned on those `2N^2` variables together with
their global indices.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .named_trajectory import NamedTrajectory


class _Fidelity:
    def __init__(self, goal_iso: np.ndarray, subspace: Optional[Sequence[int]] = None, device: int = 0):
        goal_iso = np.ascontiguousarray(goal_iso, dtype=np.float64)
        self.N = int(round((goal_iso.size / 2) ** 0.5))
        self.s = 2 * self.N * self.N
        if goal_iso.size != self.s:
            raise ValueError("goal must be an iso-vec of length 2 N^2")
        sub = None if subspace is None else np.ascontiguousarray(subspace, dtype=np.int32)
        self._h = C.c_void_p()
        rc = _lib.lib.qc_fidelity_create(self.N, _lib.dptr(goal_iso), None if sub is None else sub.ctypes.data_as(C.POINTER(C.c_int32)),
                                         0 if sub is None else sub.size, device, C.byref(self._h))
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_fidelity_last_error(None).decode())

    def eval(self, u: np.ndarray, grad: bool = True, hess: bool = True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if u.size != self.s:
            raise ValueError(f"state has length {u.size}, expected {self.s}")
        F, L = C.c_double(), C.c_double()
        g = np.empty(self.s) if grad else None
        H = np.empty(self.s * (self.s + 1) // 2) if hess else None
        rc = _lib.lib.qc_fidelity_eval(self._h, _lib.dptr(u), C.byref(F), C.byref(L), _lib.dptr(g) if grad else None,
                                       _lib.dptr(H) if hess else None)
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_fidelity_last_error(self._h).decode())
        return F.value, L.value, g, H

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib.qc_fidelity_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def iso_vec_unitary_fidelity(U_T: np.ndarray, U_G: np.ndarray, subspace: Optional[Sequence[int]] = None, device: int = 0) -> float:
    f = _Fidelity(U_G, subspace, device)
    try:
        return f.eval(U_T, grad=False, hess=False)[0]
    finally:
        f.close()


class _FinalKnotTerm:
    _ALIASES = {}

    def __init__(self, state_name: str, traj: NamedTrajectory, subspace, device):
        self.traj = traj
        self.s = len(traj.components[state_name])
        self.first = (traj.T - 1) * traj.dim + traj.offset(state_name)     # 0-based global index of the final state
        goal = traj.goal.get(state_name)
        if goal is None:
            raise ValueError(f"trajectory has no goal for {state_name}")
        self._f = _Fidelity(np.asarray(goal, dtype=np.float64), subspace, device)
        self.state_indices = np.arange(self.first, self.first + self.s)
        r, c = np.triu_indices(self.s)
        # column-major upper triangle: entry (i <= j) at j(j+1)/2 + i
        order = np.lexsort((r, c))
        self.hess_structure = (self.first + r[order], self.first + c[order])

    def _u(self, Z):
        Z = np.asarray(Z, dtype=np.float64)
        return Z[self.first:self.first + self.s]

    def __getattr__(self, name):
        al = type(self)._ALIASES
        if name in al:
            return getattr(self, al[name])
        raise AttributeError(name)

    def close(self):
        self._f.close()


class UnitaryInfidelityObjective(_FinalKnotTerm):
    """Q * |1 - F(U~_T)|.  `L(Z)`, `grad_L(Z)` (values on `state_indices`), `hess_L(Z)` (values on `hess_structure`);
    `getattr(obj, "∇L")` / `"∂²L"` resolve to the same members."""
    _ALIASES = {"∇L": "grad_L", "∂²L": "hess_L", "∂²L_structure": "hess_structure"}

    def __init__(self, state_name: str, traj: NamedTrajectory, Q: float = 100.0, subspace=None, device: int = 0):
        super().__init__(state_name, traj, subspace, device)
        self.Q = float(Q)

    def L(self, Z) -> float:
        return self.Q * self._f.eval(self._u(Z), grad=False, hess=False)[1]

    def grad_L(self, Z) -> np.ndarray:
        F, _, g, _ = self._f.eval(self._u(Z), grad=True, hess=False)
        return -np.sign(1.0 - F) * self.Q * g if F != 1.0 else -self.Q * g

    def hess_L(self, Z) -> np.ndarray:
        F, _, _, H = self._f.eval(self._u(Z), grad=False, hess=True)
        return -(1.0 if 1.0 - F >= 0 else -1.0) * self.Q * H


class FinalUnitaryFidelityConstraint(_FinalKnotTerm):
    """g(Z) = F(U~_T) - value >= 0 (one row).  `g`, `dg` (values on `state_indices`), `mu_d2g(Z, mu)`."""
    _ALIASES = {"∂g": "dg", "μ∂²g": "mu_d2g", "μ∂²g_structure": "hess_structure"}

    def __init__(self, state_name: str, value: float, traj: NamedTrajectory, subspace=None, device: int = 0):
        super().__init__(state_name, traj, subspace, device)
        self.value = float(value)
        self.dim = 1

    def g(self, Z) -> np.ndarray:
        return np.array([self._f.eval(self._u(Z), grad=False, hess=False)[0] - self.value])

    def dg(self, Z) -> np.ndarray:
        return self._f.eval(self._u(Z), grad=True, hess=False)[2]

    def mu_d2g(self, Z, mu) -> np.ndarray:
        return float(np.asarray(mu).ravel()[0]) * self._f.eval(self._u(Z), grad=False, hess=True)[3]
