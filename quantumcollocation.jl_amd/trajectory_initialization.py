"""Layout contract of the reference's trajectory initialisation (the hot path's INPUT side):
component order [states..., a, da, dda, (dt)] (reference trajectory_initialization.jl:357-382),
geodesic state guess (:140-166), random controls with zero end points (:194-223).  Host-side,
one-off, numpy/scipy; nothing here is on the timed path."""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import scipy.linalg as sla

from .isomorphisms import operator_to_iso_vec
from .named_trajectory import NamedTrajectory


def unitary_geodesic(U_init: np.ndarray, U_goal: np.ndarray, times: Union[int, Sequence[float]],
                     return_generator: bool = False):
    """Iso-vecs (2N^2 x T) of U(t) = exp(-i H (t - t0)) U_init with H = i log(U_goal U_init') / T
    (reference trajectory_initialization.jl:140-166)."""
    if np.isscalar(times):
        times = np.linspace(0.0, 1.0, int(times))
    times = np.asarray(times, dtype=float)
    t0, span = times[0], times[-1] - times[0]
    H = 1j * sla.logm(U_goal @ U_init.conj().T) / span
    cols = [operator_to_iso_vec(sla.expm(-1j * H * (t - t0)) @ U_init) for t in times]
    out = np.stack(cols, axis=1)
    return (out, H) if return_generator else out


def unitary_linear_interpolation(U_init: np.ndarray, U_goal: np.ndarray, samples: int) -> np.ndarray:
    """Iso-vecs on the straight line between the two iso-vecs (reference trajectory_initialization.jl:35-45: `geodesic=false`)."""
    u0, u1 = operator_to_iso_vec(U_init), operator_to_iso_vec(U_goal)
    lam = np.linspace(0.0, 1.0, int(samples))
    return u0[:, None] + (u1 - u0)[:, None] * lam[None, :]


def control_derivatives_from_guess(a: np.ndarray, dts, n_derivatives: int):
    """[a, da, dda, ...] from a control guess (reference trajectory_initialization.jl:225-244): every derivative by differences of the
    one before it such that the DerivativeIntegrator rows x_{t+1} - x_t - dt_t dx_t vanish at the initial point -- what the reference's
    fix-up of the last column is there for (":to avoid constraint violation error at initial iteration")."""
    a = np.array(a, dtype=float)
    T = a.shape[1]
    dts = np.full(T, float(dts)) if np.isscalar(dts) else np.asarray(dts, dtype=float).ravel()
    out = [a]
    for _ in range(n_derivatives):
        x = out[-1]
        dx = np.zeros_like(x)
        dx[:, :T - 1] = (x[:, 1:] - x[:, :T - 1]) / dts[None, :T - 1]
        dx[:, T - 1] = dx[:, T - 2] if T > 1 else 0.0
        out.append(dx)
    return out


def initialize_control_trajectory(n_drives: int, n_derivatives: int, T: int, bounds: Sequence[float],
                                  drive_derivative_sigma: float, rng: np.random.Generator):
    """a: zeros at both ends, Uniform(-b, b) inside; derivatives N(0, sigma^2)
    (reference trajectory_initialization.jl:194-223)."""
    a = np.zeros((n_drives, T))
    for i in range(n_drives):
        a[i, 1:T - 1] = rng.uniform(-bounds[i], bounds[i], size=T - 2)
    out = [a]
    for _ in range(n_derivatives):
        out.append(rng.standard_normal((n_drives, T)) * drive_derivative_sigma)
    return out


def initialize_trajectory(U_goal: np.ndarray, T: int, dt: float, n_drives: int,
                          control_bounds: Sequence[Sequence[float]], *, free_time: bool = True,
                          state_name: str = "Ũ⃗", control_name: str = "a", timestep_name: str = "Δt",
                          dt_bounds: Optional[tuple] = None, drive_derivative_sigma: float = 0.1,
                          state_noise: float = 0.0, rng: Optional[np.random.Generator] = None,
                          U_init: Optional[np.ndarray] = None, a_guess: Optional[np.ndarray] = None, system=None,
                          geodesic: bool = True, device: int = 0) -> NamedTrajectory:
    """Unitary trajectory in the reference's component order [U~, a, da, dda, dt]
    (reference trajectory_initialization.jl:357-382,389-444).  With `a_guess` (and its `system`) the states are the ROLLOUT of the
    guess, `unitary_rollout(U~_init, a_guess, timesteps, system)` (:422-426) -- on the GPU (qc_rollout) -- and the control
    derivatives its differences (:225-244); `geodesic=False` interpolates the iso-vecs linearly (:176-188)."""
    rng = rng if rng is not None else np.random.default_rng()
    N = U_goal.shape[0]
    U_init = np.eye(N, dtype=complex) if U_init is None else U_init
    n_deriv = len(control_bounds) - 1
    if a_guess is not None:
        if system is None:
            raise ValueError("System must be provided if a_guess is provided.")
        from .rollouts import unitary_rollout
        a_guess = np.asarray(a_guess, dtype=float)
        if a_guess.shape != (n_drives, T):
            raise ValueError(f"a_guess has shape {a_guess.shape}, expected ({n_drives}, {T})")
        states = unitary_rollout(operator_to_iso_vec(U_init), a_guess, np.full(T, float(dt)), system, device=device)
        if state_noise:
            states = states + rng.standard_normal(states.shape) * state_noise
        ctrl = control_derivatives_from_guess(a_guess, dt, n_deriv)
    else:
        states = unitary_geodesic(U_init, U_goal, T) if geodesic else unitary_linear_interpolation(U_init, U_goal, T)
        if state_noise:     # (drawn BEFORE the controls: the seeded synthetic inputs of bench.py and of the tests depend on the order)
            states = states + rng.standard_normal(states.shape) * state_noise
        ctrl = initialize_control_trajectory(n_drives, n_deriv, T, control_bounds[0], drive_derivative_sigma, rng)
    names = [control_name] + ["d" * i + control_name for i in range(1, n_deriv + 1)]
    comps = {state_name: states}
    for nm, c in zip(names, ctrl):
        comps[nm] = c
    bounds = {nm: (-np.asarray(b, dtype=float), np.asarray(b, dtype=float)) for nm, b in zip(names, control_bounds)}
    if free_time:
        comps[timestep_name] = np.full((1, T), dt)
        bounds[timestep_name] = dt_bounds if dt_bounds is not None else (0.5 * dt, 1.5 * dt)
        timestep: Union[str, float] = timestep_name
        controls = (names[-1], timestep_name)
    else:
        timestep = float(dt)
        controls = (names[-1],)
    return NamedTrajectory(
        comps, controls=controls, timestep=timestep, bounds=bounds,
        initial={state_name: operator_to_iso_vec(U_init), control_name: np.zeros(n_drives)},
        final={control_name: np.zeros(n_drives)},
        goal={state_name: operator_to_iso_vec(U_goal)},
    )
