"""Final-knot fidelity terms ("next" row 1), evaluated on the GPU through
`qc_fidelity_*` (include/qcolloc.h):

    iso_vec_unitary_fidelity(U_T, U_G; subspace)                      unitary_minimum_time_problem.jl:77
    UnitaryInfidelityObjective(state_name, traj, Q; subspace)         unitary_smooth_pulse_problem.jl:133-137
    FinalUnitaryFidelityConstraint(state_name, val, traj; subspace)   unitary_minimum_time_problem.jl:80-84

Loss per the reference docstring (unitary_smooth_pulse_problem.jl:23-28): l = |1 - |tr(U_goal' U_T)| / N|.
Only the last knot's state enters; gradients/Hessians are returned on those `2N^2` variables together with
their global indices.

Whole-trajectory terms through `qc_terms_*`: `QuadraticRegularizer`, `MinimumTimeObjective`
(summed into one `TrajectoryObjective`) and the constant `TimeStepsAllEqualConstraint`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .named_trajectory import NamedTrajectory


class _Fidelity:
    def __init__(self, goal_iso: np.ndarray, subspace: Optional[Sequence[int]] = None, device: int = 0, kind: str = "unitary",
                 form: str = "abs", phase_operators: Optional[Sequence[np.ndarray]] = None):
        goal_iso = np.ascontiguousarray(goal_iso, dtype=np.float64)
        self._h = C.c_void_p()
        self.K = 0
        if kind == "unitary":
            self.N = int(round((goal_iso.size / 2) ** 0.5))
            self.s = 2 * self.N * self.N
            if goal_iso.size != self.s:
                raise ValueError("goal must be an iso-vec of length 2 N^2")
            sub = None if subspace is None else np.ascontiguousarray(subspace, dtype=np.int32)
            d = _lib.qc_fidelity_desc()
            d.kind, d.N, d.goal_iso = _lib.QC_FID_UNITARY, self.N, _lib.dptr(goal_iso)
            d.subspace = None if sub is None else sub.ctypes.data_as(C.POINTER(C.c_int32))
            d.n_sub = 0 if sub is None else sub.size
            d.form = {"abs": _lib.QC_FID_FORM_ABS, "abs2": _lib.QC_FID_FORM_ABS2}[form]
            d.device = device
            keep = []
            if phase_operators is not None and len(phase_operators):
                ops = [np.asarray(Op, dtype=complex) for Op in phase_operators]
                dims = np.ascontiguousarray([Op.shape[0] for Op in ops], dtype=np.int32)
                planes = np.ascontiguousarray(np.concatenate([np.concatenate([Op.real.reshape(-1, order="F"), Op.imag.reshape(-1, order="F")])
                                                              for Op in ops]))
                d.n_phases = self.K = len(ops)
                d.phase_dims = dims.ctypes.data_as(C.POINTER(C.c_int32))
                d.phase_ops = _lib.dptr(planes)
                keep = [dims, planes]
            rc = _lib.lib.qc_fidelity_create_desc(C.byref(d), C.byref(self._h))
            del keep
        else:   # "ket": state psi~ (2N);  "density": state rho~ (2N^2) against the pure goal |psi_goal><psi_goal|
            if subspace is not None or phase_operators is not None or form != "abs":
                raise ValueError("subspace, form and free phases apply to unitary fidelities only")
            if goal_iso.size % 2:
                raise ValueError("the goal ket must be an iso-vec [Re psi; Im psi]")
            self.N = goal_iso.size // 2
            self.s = 2 * self.N if kind == "ket" else 2 * self.N * self.N
            rc = _lib.lib.qc_fidelity_create_kind(_lib.QC_FID_KET if kind == "ket" else _lib.QC_FID_DENSITY, self.N, _lib.dptr(goal_iso),
                                                  device, C.byref(self._h))
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_fidelity_last_error(None).decode())
        self.P = self.s + self.K          # input length: [state ; free phases]

    def eval(self, u: np.ndarray, grad: bool = True, hess: bool = True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if u.size != self.P:
            raise ValueError(f"input has length {u.size}, expected {self.P} (state{' + phases' if self.K else ''})")
        F, L = C.c_double(), C.c_double()
        g = np.empty(self.P) if grad else None
        H = np.empty(self.P * (self.P + 1) // 2) if hess else None
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


def iso_vec_unitary_fidelity(U_T: np.ndarray, U_G: np.ndarray, subspace: Optional[Sequence[int]] = None, device: int = 0,
                             form: str = "abs") -> float:
    """`form`: "abs" = |tr(U_G' U_T)| / n (the reference's docstring), "abs2" = |tr|^2 / n^2 (INTEGRATION.md, table of choices)."""
    f = _Fidelity(U_G, subspace, device, form=form)
    try:
        return f.eval(U_T, grad=False, hess=False)[0]
    finally:
        f.close()


def iso_vec_unitary_free_phase_fidelity(U_T: np.ndarray, U_G: np.ndarray, phases, phase_operators, subspace=None, device: int = 0,
                                        form: str = "abs") -> float:
    """|tr(U_G' R(phi) U_T)| / n with R(phi) = kron_k exp(i phi_k Op_k) (reference unitary_minimum_time_problem.jl:86-90)."""
    f = _Fidelity(U_G, subspace, device, form=form, phase_operators=phase_operators)
    try:
        return f.eval(np.concatenate([np.asarray(U_T, dtype=np.float64), np.asarray(phases, dtype=np.float64).ravel()]),
                      grad=False, hess=False)[0]
    finally:
        f.close()


class _FinalKnotTerm:
    _ALIASES = {}
    _KIND = "unitary"

    def __init__(self, state_name: str, traj: NamedTrajectory, subspace, device, goal=None, form: str = "abs",
                 phase_name: Optional[str] = None, phase_operators=None):
        self.traj = traj
        self.s = len(traj.components[state_name])
        self.first = (traj.T - 1) * traj.dim + traj.offset(state_name)     # 0-based global index of the final state
        goal = traj.goal.get(state_name) if goal is None else goal
        if goal is None:
            raise ValueError(f"trajectory has no goal for {state_name}")
        self._f = _Fidelity(np.asarray(goal, dtype=np.float64), subspace, device, kind=type(self)._KIND, form=form,
                            phase_operators=phase_operators)
        if self._f.s != self.s:
            raise ValueError(f"component {state_name} has length {self.s}, the goal implies {self._f.s}")
        idx = np.arange(self.first, self.first + self.s)
        if self._f.K:
            # the free phases are global variables behind the knots: Z = [vec(data) ; global_data...]  (trajectory_initialization.jl:370-380)
            off = traj.T * traj.dim
            for name, v in traj.global_data.items():
                if name == phase_name:
                    break
                off += v.size
            else:
                raise ValueError(f"trajectory has no global component {phase_name}")
            if traj.global_data[phase_name].size != self._f.K:
                raise ValueError("one phase per phase operator")
            idx = np.concatenate([idx, np.arange(off, off + self._f.K)])
        self.state_indices = idx           # variables of the term: the final state (and the free phases)
        P = idx.size
        r, c = np.triu_indices(P)
        # column-major upper triangle: entry (i <= j) at j(j+1)/2 + i
        order = np.lexsort((r, c))
        self.hess_structure = (idx[r[order]], idx[c[order]])

    def _u(self, Z):
        Z = np.asarray(Z, dtype=np.float64)
        return Z[self.state_indices]

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

    def __init__(self, state_name: str, traj: NamedTrajectory, Q: float = 100.0, subspace=None, device: int = 0, form: str = "abs"):
        super().__init__(state_name, traj, subspace, device, form=form)
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

    def __init__(self, state_name: str, value: float, traj: NamedTrajectory, subspace=None, device: int = 0, form: str = "abs"):
        super().__init__(state_name, traj, subspace, device, form=form)
        self.value = float(value)
        self.dim = 1

    def g(self, Z) -> np.ndarray:
        return np.array([self._f.eval(self._u(Z), grad=False, hess=False)[0] - self.value])

    def dg(self, Z) -> np.ndarray:
        return self._f.eval(self._u(Z), grad=True, hess=False)[2]

    def mu_d2g(self, Z, mu) -> np.ndarray:
        return float(np.asarray(mu).ravel()[0]) * self._f.eval(self._u(Z), grad=False, hess=True)[3]


class UnitaryFreePhaseInfidelityObjective(UnitaryInfidelityObjective):
    """Q * |1 - F(U~_T, phi)| with free phases (reference unitary_smooth_pulse_problem.jl:138-143): variables = the final state
    and the K global phases `traj.global_data[phase_name]`."""

    def __init__(self, state_name: str, phase_name: str, phase_operators, traj: NamedTrajectory, Q: float = 100.0, subspace=None,
                 device: int = 0, form: str = "abs"):
        _FinalKnotTerm.__init__(self, state_name, traj, subspace, device, form=form, phase_name=phase_name, phase_operators=phase_operators)
        self.Q = float(Q)


class FinalUnitaryFreePhaseFidelityConstraint(FinalUnitaryFidelityConstraint):
    """g(Z) = F(U~_T, phi) - value >= 0 (reference unitary_minimum_time_problem.jl:95-100)."""

    def __init__(self, state_name: str, phase_name: str, phase_operators, value: float, traj: NamedTrajectory, subspace=None,
                 device: int = 0, form: str = "abs"):
        _FinalKnotTerm.__init__(self, state_name, traj, subspace, device, form=form, phase_name=phase_name, phase_operators=phase_operators)
        self.value = float(value)
        self.dim = 1


def iso_fidelity(psi_iso: np.ndarray, psi_goal_iso: np.ndarray, device: int = 0) -> float:
    """|<psi_goal|psi>|^2 on ket iso-vecs (reference quantum_state_minimum_time_problem.jl:50)."""
    f = _Fidelity(psi_goal_iso, None, device, kind="ket")
    try:
        return f.eval(psi_iso, grad=False, hess=False)[0]
    finally:
        f.close()


class QuantumStateObjective(UnitaryInfidelityObjective):
    """Q * |1 - |<psi_goal|psi_T>|^2| on the final ket (reference quantum_state_smooth_pulse_problem.jl:133)."""
    _KIND = "ket"

    def __init__(self, state_name: str, traj: NamedTrajectory, Q: float = 100.0, device: int = 0):
        _FinalKnotTerm.__init__(self, state_name, traj, None, device)
        self.Q = float(Q)


class FinalQuantumStateFidelityConstraint(FinalUnitaryFidelityConstraint):
    """g(Z) = |<psi_goal|psi_T>|^2 - value >= 0 (reference quantum_state_minimum_time_problem.jl:55-62)."""
    _KIND = "ket"

    def __init__(self, state_name: str, value: float, traj: NamedTrajectory, device: int = 0):
        _FinalKnotTerm.__init__(self, state_name, traj, None, device)
        self.value = float(value)
        self.dim = 1


class DensityOperatorPureStateInfidelityObjective(UnitaryInfidelityObjective):
    """Q * |1 - psi_goal' rho_T psi_goal| on the final density iso-vec (reference density_operator_smooth_pulse_problem.jl:55)."""
    _KIND = "density"

    def __init__(self, state_name: str, psi_goal: np.ndarray, traj: NamedTrajectory, Q: float = 100.0, device: int = 0):
        psi_goal = np.asarray(psi_goal, dtype=complex)
        _FinalKnotTerm.__init__(self, state_name, traj, None, device, goal=np.concatenate([psi_goal.real, psi_goal.imag]))
        self.Q = float(Q)


# ---------------------------------------------------------------------------------------------------------------
#  Whole-trajectory cost terms through `qc_terms_*`
# ---------------------------------------------------------------------------------------------------------------
class QuadraticRegularizer:
    """`QuadraticRegularizer(name, traj, R; baseline, timestep_name)` (reference call sites
    unitary_smooth_pulse_problem.jl:151-153): 1/2 sum_t sum_i R_i (dt_t (x_ti - b_ti))^2.  A description only; add it
    to a `TrajectoryObjective` to evaluate it."""

    def __init__(self, name: str, traj: NamedTrajectory, R, baseline: Optional[np.ndarray] = None, timestep_name: Optional[str] = None):
        self.name = name
        self.dim = len(traj.components[name])
        R = np.asarray(R, dtype=np.float64)
        self.R = np.full(self.dim, float(R)) if R.ndim == 0 else R.copy()
        if self.R.shape != (self.dim,):
            raise ValueError(f"R has shape {self.R.shape}, expected ({self.dim},)")
        self.baseline = None if baseline is None else np.asarray(baseline, dtype=np.float64).reshape(self.dim, traj.T)
        self.timestep_name = timestep_name if timestep_name is not None else (traj.timestep if isinstance(traj.timestep, str) else None)

    def __add__(self, other):
        return TrajectoryObjectiveSpec([self]) + other


class MinimumTimeObjective:
    """`MinimumTimeObjective(traj; D)` (reference unitary_minimum_time_problem.jl:67-69): D * sum_{t=1}^{T-1} dt_t."""

    def __init__(self, traj: NamedTrajectory, D: float = 1.0, timestep_name: Optional[str] = None):
        self.D = float(D)
        self.timestep_name = timestep_name if timestep_name is not None else traj.timestep
        if not isinstance(self.timestep_name, str):
            raise ValueError("a minimum-time objective needs a free timestep component")

    def __add__(self, other):
        return TrajectoryObjectiveSpec([self]) + other


class TrajectoryObjectiveSpec:
    def __init__(self, terms):
        self.terms = list(terms)

    def __add__(self, other):
        more = other.terms if isinstance(other, TrajectoryObjectiveSpec) else [other]
        return TrajectoryObjectiveSpec(self.terms + list(more))


class TrajectoryObjective:
    """Sum of `QuadraticRegularizer` / `MinimumTimeObjective` terms evaluated in one pass over the knots on the GPU.
    `L(Z)`, `grad_L(Z)` (dense, length `len(Z)`), `hess_L(Z)` (values on `hess_structure`); `"∇L"`, `"∂²L"`,
    `"∂²L_structure"` resolve to the same members."""
    _ALIASES = {"∇L": "grad_L", "∂²L": "hess_L", "∂²L_structure": "hess_structure"}

    def __init__(self, terms, traj: NamedTrajectory, dt_scaled: bool = True, device: int = 0):
        """dt_scaled=True (default): 1/2 sum_t R (dt_t x_t)^2, the weighting the templates' `timestep_name=` argument implies
        (unitary_smooth_pulse_problem.jl:151-153); False: the docstring's 1/2 sum_t R x_t^2 (QC_REG_PLAIN)."""
        if isinstance(terms, TrajectoryObjectiveSpec):
            terms = terms.terms
        elif isinstance(terms, (QuadraticRegularizer, MinimumTimeObjective)):
            terms = [terms]
        self.traj = traj
        w = np.zeros(traj.dim)
        used = np.zeros(traj.dim, dtype=bool)
        base = np.zeros((traj.dim, traj.T))
        any_base = False
        D = 0.0
        ts_names = set()
        for term in terms:
            if isinstance(term, QuadraticRegularizer):
                idx = np.asarray(traj.components[term.name])
                w[idx] += term.R
                if term.baseline is not None:
                    if used[idx].any():
                        raise ValueError(f"{term.name}: a baseline cannot be combined with another regulariser on the same component")
                    base[idx, :] = term.baseline
                    any_base = True
                used[idx] = True
                if term.timestep_name is not None:
                    ts_names.add(term.timestep_name)
            elif isinstance(term, MinimumTimeObjective):
                D += term.D
                ts_names.add(term.timestep_name)
            else:
                raise TypeError(f"unsupported term {type(term).__name__}")
        if len(ts_names) > 1:
            raise ValueError(f"terms disagree on the timestep component: {sorted(ts_names)}")
        free = isinstance(traj.timestep, str)
        self._index = np.ascontiguousarray(np.nonzero(used)[0], dtype=np.int32)
        self._R = np.ascontiguousarray(w[self._index])
        self._base = np.ascontiguousarray(base[self._index, :].T) if any_base else None      # (T, n_reg): entry-fastest
        d = _lib.qc_terms_desc()
        d.T = traj.T
        d.zdim = traj.dim
        d.off_dt = traj.offset(traj.timestep) if free else -1
        d.global_dim = traj.global_dim
        d.dt_fixed = 0.0 if free else float(traj.timestep)
        d.n_reg = self._index.size
        d.weighting = _lib.QC_REG_DT_SCALED if dt_scaled else _lib.QC_REG_PLAIN
        d.reg_index = self._index.ctypes.data_as(C.POINTER(C.c_int32))
        d.reg_R = _lib.dptr(self._R) if self._R.size else None
        d.reg_baseline = _lib.dptr(self._base) if self._base is not None else None
        d.min_time_D = D
        d.min_time_knots = traj.T - 1 if D != 0.0 else 0
        d.device = device
        self._desc = d
        self.Z_len = traj.T * traj.dim + traj.global_dim
        self._h = C.c_void_p()
        rc = _lib.lib.qc_terms_create(C.byref(d), C.byref(self._h))
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_terms_last_error(None).decode())
        nnz = C.c_int64()
        _lib.lib.qc_terms_hess_nnz(self._h, C.byref(nnz))
        self.hess_nnz = nnz.value
        r = np.empty(self.hess_nnz, dtype=np.int64)
        c = np.empty(self.hess_nnz, dtype=np.int64)
        _lib.lib.qc_terms_hess_structure(self._h, _lib.iptr(r), _lib.iptr(c), 0)
        self.hess_structure = (r, c)

    def _eval(self, Z, grad: bool, hess: bool):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.size != self.Z_len:
            raise ValueError(f"Z has length {Z.size}, expected {self.Z_len}")
        J = C.c_double()
        g = np.empty(self.Z_len) if grad else None
        H = np.empty(self.hess_nnz) if hess else None
        rc = _lib.lib.qc_terms_eval(self._h, _lib.dptr(Z), C.byref(J), _lib.dptr(g) if grad else None, _lib.dptr(H) if hess else None)
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_terms_last_error(self._h).decode())
        return J.value, g, H

    def L(self, Z) -> float:
        return self._eval(Z, False, False)[0]

    def grad_L(self, Z) -> np.ndarray:
        return self._eval(Z, True, False)[1]

    def hess_L(self, Z) -> np.ndarray:
        return self._eval(Z, False, True)[2]

    def L_grad_hess(self, Z):
        return self._eval(Z, True, True)

    def eval_device(self, dZ, dJ, dgrad=None, dhess=None, stream=None):
        """Device-resident evaluation on torch CUDA tensors (float64), asynchronous on `stream`."""
        s = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        rc = _lib.lib.qc_terms_eval_dev(self._h, dZ.data_ptr(), dJ.data_ptr(), dgrad.data_ptr() if dgrad is not None else None,
                                        dhess.data_ptr() if dhess is not None else None, s)
        if rc != _lib.QC_OK:
            raise _lib.QCollocError(rc, _lib.lib.qc_terms_last_error(self._h).decode())

    def __getattr__(self, name):
        al = type(self)._ALIASES
        if name in al:
            return getattr(self, al[name])
        raise AttributeError(name)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib.qc_terms_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TimeStepsAllEqualConstraint:
    """`TimeStepsAllEqualConstraint(timestep_name, traj)` (reference _problem_templates.jl:57-63): the T-1 linear rows
    dt_t - dt_T = 0.  Constant Jacobian (+1, -1), no Hessian: nothing to evaluate on a device; provided so that the
    whole constraint set of a problem template can be described on this side of the boundary."""

    def __init__(self, timestep_name: str, traj: NamedTrajectory):
        off = traj.offset(timestep_name)
        self.dim = traj.T - 1
        self.indices = np.arange(traj.T, dtype=np.int64) * traj.dim + off
        rows = np.repeat(np.arange(self.dim, dtype=np.int64), 2)
        cols = np.stack([self.indices[:-1], np.full(self.dim, self.indices[-1])], axis=1).ravel()
        self.jac_structure = (rows, cols)
        self.jac_values = np.tile([1.0, -1.0], self.dim)

    def g(self, Z) -> np.ndarray:
        Z = np.asarray(Z, dtype=np.float64)
        return Z[self.indices[:-1]] - Z[self.indices[-1]]

    def dg(self, Z=None) -> np.ndarray:
        return self.jac_values
