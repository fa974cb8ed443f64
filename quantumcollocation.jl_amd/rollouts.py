"""Rollouts on the GPU through `qc_rollout`:

    unitary_rollout(Ũ⃗_init, controls, Δt, system)          trajectory_initialization.jl:426
    rollout(ψ̃_init, controls, Δt, system)                   trajectory_initialization.jl:493
    open_rollout(ρ⃗̃_init, controls, Δt, system)             trajectory_initialization.jl:547
    unitary_rollout_fidelity(traj, system; subspace)        unitary_smooth_pulse_problem.jl:218

x_{t+1} = exp(Δt_t G(a_t)) x_t; the result has one column per knot.  The functions taking raw controls build a
minimal trajectory layout [state, a, Δt] for the descriptor; `QuantumDynamics.rollout` reuses an existing handle.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .isomorphisms import operator_to_iso_vec
from .named_trajectory import NamedTrajectory
from .objectives import iso_vec_unitary_fidelity


def _rollout(init: np.ndarray, controls: np.ndarray, dts, system, cols: int, device: int = 0) -> np.ndarray:
    controls = np.asarray(controls, dtype=np.float64)
    if controls.ndim != 2 or controls.shape[0] != system.n_drives:
        raise ValueError("controls must be n_drives x T")
    m, T = controls.shape
    dts = np.full(T, float(dts)) if np.ndim(dts) == 0 else np.asarray(dts, dtype=np.float64).ravel()
    if dts.size != T:
        raise ValueError("one timestep per knot expected")
    init = np.ascontiguousarray(init, dtype=np.float64).ravel()
    n = 2 * system.state_levels
    if init.size != n * cols:
        raise ValueError(f"initial state has length {init.size}, expected {n * cols}")
    if T < 2:
        return init[:, None].copy()
    s = init.size
    d = _lib.qc_desc()
    d.N, d.m, d.T, d.zdim, d.global_dim = system.state_levels, m, T, s + m + 1, 0
    d.off_U, d.off_a, d.off_dt, d.dt_fixed = 0, s, s + m, 0.0
    d.integrator, d.pade_order, d.n_deriv = _lib.QC_EXPONENTIAL, 0, 0
    d.state_cols = 0 if cols == system.state_levels else cols
    G0 = np.asfortranarray(system.G_drift, dtype=np.float64)
    Gd = np.ascontiguousarray(np.stack([np.asarray(G, dtype=np.float64).reshape(-1, order="F") for G in system.G_drives])
                              if m else np.zeros((1, n * n)))
    d.G_drift, d.G_drives = _lib.dptr(G0), _lib.dptr(Gd)
    d.device, d.kernel = device, _lib.QC_KERNEL_LDS
    h = C.c_void_p()
    _lib.check(_lib.lib.qc_create(C.byref(d), C.byref(h)))
    try:
        Z = np.zeros((T, s + m + 1))
        Z[:, s:s + m] = controls.T
        Z[:, s + m] = dts
        out = np.empty((T, s))
        _lib.check(_lib.lib.qc_rollout(h, _lib.dptr(Z), _lib.dptr(init), _lib.dptr(out)), h)
    finally:
        _lib.lib.qc_destroy(h)
    return np.ascontiguousarray(out.T)


def unitary_rollout(U_iso_init: np.ndarray, controls: np.ndarray, dts, system, device: int = 0) -> np.ndarray:
    """Ũ⃗ trajectory (2N^2 x T)."""
    return _rollout(U_iso_init, controls, dts, system, system.levels, device)


def rollout(psi_iso_init: np.ndarray, controls: np.ndarray, dts, system, device: int = 0) -> np.ndarray:
    """ψ̃ trajectory (2N x T) of one ket."""
    return _rollout(psi_iso_init, controls, dts, system, 1, device)


def open_rollout(rho_iso_init: np.ndarray, controls: np.ndarray, dts, system, device: int = 0) -> np.ndarray:
    """ρ⃗̃ trajectory (2N^2 x T) under the Lindblad generators of an `OpenQuantumSystem`."""
    return _rollout(rho_iso_init, controls, dts, system, 1, device)


def unitary_rollout_fidelity(traj: NamedTrajectory, system, state_name: str = "Ũ⃗", control_name: str = "a",
                             subspace: Optional[Sequence[int]] = None, device: int = 0) -> float:
    """Fidelity of the rolled-out final unitary with `traj.goal[state_name]`."""
    init = traj.initial.get(state_name) if getattr(traj, "initial", None) else None
    if init is None:
        init = operator_to_iso_vec(np.eye(system.levels, dtype=complex))
    dts = traj[traj.timestep].ravel() if isinstance(traj.timestep, str) else float(traj.timestep)
    U = unitary_rollout(np.asarray(init, dtype=np.float64), traj[control_name], dts, system, device)
    return iso_vec_unitary_fidelity(U[:, -1], np.asarray(traj.goal[state_name], dtype=np.float64), subspace, device)


def rollout_fidelity(traj: NamedTrajectory, system, state_name: str = "ψ̃", control_name: str = "a", device: int = 0) -> float:
    """`rollout_fidelity(traj, system; state_name)` for a ket component (reference quantum_state_smooth_pulse_problem.jl:247-249,
    quantum_state_sampling_problem.jl:187-189): the ket rolled out from its first knot under the trajectory's controls, |<goal|psi_T>|^2
    with `traj.goal[state_name]`."""
    from .objectives import iso_fidelity
    init = traj.initial.get(state_name) if getattr(traj, "initial", None) else None
    if init is None:
        init = traj[state_name][:, 0]
    dts = traj[traj.timestep].ravel() if isinstance(traj.timestep, str) else float(traj.timestep)
    psi = rollout(np.ascontiguousarray(init, dtype=np.float64), traj[control_name], dts, system, device)
    return iso_fidelity(psi[:, -1], np.asarray(traj.goal[state_name], dtype=np.float64), device)
