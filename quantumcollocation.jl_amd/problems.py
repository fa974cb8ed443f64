"""Inputs to the hot path for the BASELINE.json configurations, built the way
`UnitarySmoothPulseProblem` builds them (reference unitary_smooth_pulse_problem.jl:70-191): trajectory
via `initialize_trajectory`, then [unitary integrator, DerivativeIntegrator(a, da),
DerivativeIntegrator(da, dda)] (:163-179).  Objectives, bounds handling, Ipopt and the problem
struct stay on the reference side and are out of scope.

Synthetic inputs follow SURVEY.md section 8(d): seed 20250218, drift 0.1 * sum Z_i Z_{i+1}
(1 qubit: 0.1 Z as in reference test/test_utils.jl:123), drives X_i, Y_i per qubit, geodesic states +
N(0, 1e-2) noise, a in U(-1,1) with zero ends, da/dda ~ N(0, 0.1^2), dt = 0.2.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .gates import GATES, operator_from_string
from .integrators import (DensityOperatorExponentialIntegrator, DerivativeIntegrator, QuantumStateExponentialIntegrator, QuantumStatePadeIntegrator,
                          UnitaryExponentialIntegrator, UnitaryPadeIntegrator)
from .named_trajectory import NamedTrajectory
from .quantum_systems import QuantumSystem
from .trajectory_initialization import initialize_trajectory

SEED = 20250218


@dataclass
class ConfigSpec:
    name: str
    n_qubits: int
    gate: str
    T: int
    description: str


CONFIGS = {
    1: ConfigSpec("config1", 1, "H", 50, "1-qubit Hadamard UnitarySmoothPulseProblem, T=50, dt=0.2, X/Y drives"),
    2: ConfigSpec("config2", 2, "CNOT", 200, "2-qubit CNOT UnitarySmoothPulseProblem, T=200, 4th-order Pade"),
    3: ConfigSpec("config3", 3, "TOFFOLI", 1000, "3-qubit Toffoli UnitarySmoothPulseProblem, T=1000"),
    4: ConfigSpec("config4", 3, "TOFFOLI", 8000, "3-qubit Toffoli T=8000, knot-sharded across GPUs"),
    5: ConfigSpec("config5", 4, "QFT16", 500, "4-qubit QFT UnitaryMinimumTimeProblem, T=500, free dt"),
}


def multi_qubit_system(n_qubits: int, zz: float = 0.1) -> QuantumSystem:
    def term(op: str, i: int) -> np.ndarray:
        return operator_from_string("".join(op if k == i else "I" for k in range(n_qubits)))

    if n_qubits == 1:
        H_drift = zz * operator_from_string("Z")
    else:
        H_drift = sum(zz * operator_from_string("".join("Z" if k in (i, i + 1) else "I" for k in range(n_qubits)))
                      for i in range(n_qubits - 1))
    H_drives = []
    for i in range(n_qubits):
        H_drives += [term("X", i), term("Y", i)]
    return QuantumSystem(H_drift, H_drives)


@dataclass
class HotPathInputs:
    system: QuantumSystem
    traj: NamedTrajectory
    integrators: List
    spec: Optional[ConfigSpec] = None


def unitary_smooth_pulse_inputs(system: QuantumSystem, U_goal: np.ndarray, T: int, dt: float = 0.2, *,
                                free_time: bool = True, integrator: str = "pade", pade_order: int = 4,
                                a_bound: float = 1.0, dda_bound: float = 1.0, state_noise: float = 1e-2,
                                seed: int = SEED) -> HotPathInputs:
    rng = np.random.default_rng(seed)
    m = system.n_drives
    traj = initialize_trajectory(
        U_goal, T, dt, m, ([a_bound] * m, [np.inf] * m, [dda_bound] * m),
        free_time=free_time, state_noise=state_noise, rng=rng)
    if integrator == "pade":
        U_int = UnitaryPadeIntegrator("Ũ⃗", "a", system, traj, order=pade_order)
    elif integrator == "exponential":
        U_int = UnitaryExponentialIntegrator("Ũ⃗", "a", system, traj)
    else:
        raise ValueError("integrator must be one of ('pade', 'exponential')")
    integrators = [U_int, DerivativeIntegrator("a", "da", traj), DerivativeIntegrator("da", "dda", traj)]
    return HotPathInputs(system, traj, integrators)


def config_inputs(cfg: int, T: Optional[int] = None, **kw) -> HotPathInputs:
    spec = CONFIGS[cfg]
    system = multi_qubit_system(spec.n_qubits)
    out = unitary_smooth_pulse_inputs(system, GATES[spec.gate], T if T is not None else spec.T, **kw)
    out.spec = spec
    return out


def quantum_state_smooth_pulse_inputs(system: QuantumSystem, psi_inits, psi_goals, T: int, dt: float = 0.2, *,
                                      free_time: bool = True, integrator: str = "pade", pade_order: int = 4,
                                      seed: int = SEED) -> HotPathInputs:
    """Inputs of `QuantumStateSmoothPulseProblem` (reference quantum_state_smooth_pulse_problem.jl:57-208): one
    trajectory component and one integrator per ket (:146-152), then the two derivative integrators.  States are
    linear interpolations init -> goal plus noise (the hot path does not care how the guess was made)."""
    from .named_trajectory import NamedTrajectory
    rng = np.random.default_rng(seed)
    m, N = system.n_drives, system.levels
    comps = {}
    names = []
    inits, goals = {}, {}
    for k, (p0, p1) in enumerate(zip(psi_inits, psi_goals)):
        p0 = np.asarray(p0, dtype=complex)
        p1 = np.asarray(p1, dtype=complex)
        lam = np.linspace(0.0, 1.0, T)[None, :]
        iso0 = np.concatenate([p0.real, p0.imag])[:, None]
        iso1 = np.concatenate([p1.real, p1.imag])[:, None]
        name = f"ψ̃{k + 1}" if len(psi_inits) > 1 else "ψ̃"
        comps[name] = iso0 * (1 - lam) + iso1 * lam + 1e-2 * rng.standard_normal((2 * N, T))
        names.append(name)
        inits[name], goals[name] = iso0[:, 0], iso1[:, 0]
    a = np.zeros((m, T))
    a[:, 1:T - 1] = rng.uniform(-1, 1, size=(m, T - 2))
    comps["a"] = a
    comps["da"] = 0.1 * rng.standard_normal((m, T))
    comps["dda"] = 0.1 * rng.standard_normal((m, T))
    if free_time:
        comps["Δt"] = np.full((1, T), dt)
    traj = NamedTrajectory(comps, controls=("dda", "Δt") if free_time else ("dda",), timestep="Δt" if free_time else dt,
                           initial=inits, goal=goals)
    cls = QuantumStatePadeIntegrator if integrator == "pade" else QuantumStateExponentialIntegrator
    kw = {"order": pade_order} if integrator == "pade" else {}
    integrators = [cls(nm, "a", system, traj, **kw) for nm in names]
    integrators += [DerivativeIntegrator("a", "da", traj), DerivativeIntegrator("da", "dda", traj)]
    return HotPathInputs(system, traj, integrators)


def density_operator_smooth_pulse_inputs(system, rho_init: np.ndarray, psi_goal: np.ndarray, T: int, dt: float = 0.2, *,
                                         free_time: bool = True, seed: int = SEED) -> HotPathInputs:
    """Inputs of `DensityOperatorSmoothPulseProblem` (reference density_operator_smooth_pulse_problem.jl:3-124): state
    component `ρ⃗̃` = iso-vec of vec(rho), the density-operator exponential integrator (:104-106) followed by the two
    derivative integrators (:108-112).  The guess interpolates rho_init -> |psi_goal><psi_goal| linearly plus noise."""
    from .isomorphisms import density_to_iso_vec
    from .named_trajectory import NamedTrajectory
    rng = np.random.default_rng(seed)
    m, N = system.n_drives, system.levels
    psi_goal = np.asarray(psi_goal, dtype=complex)
    r0 = density_to_iso_vec(rho_init)[:, None]
    r1 = density_to_iso_vec(np.outer(psi_goal, psi_goal.conj()))[:, None]
    lam = np.linspace(0.0, 1.0, T)[None, :]
    comps = {"ρ⃗̃": r0 * (1 - lam) + r1 * lam + 1e-2 * rng.standard_normal((2 * N * N, T))}
    a = np.zeros((m, T))
    a[:, 1:T - 1] = rng.uniform(-1, 1, size=(m, T - 2))
    comps["a"] = a
    comps["da"] = 0.1 * rng.standard_normal((m, T))
    comps["dda"] = 0.1 * rng.standard_normal((m, T))
    if free_time:
        comps["Δt"] = np.full((1, T), dt)
    traj = NamedTrajectory(comps, controls=("dda", "Δt") if free_time else ("dda",), timestep="Δt" if free_time else dt,
                           goal={"ρ⃗̃": r1[:, 0]})
    integrators = [DensityOperatorExponentialIntegrator("ρ⃗̃", "a", system, traj),
                   DerivativeIntegrator("a", "da", traj), DerivativeIntegrator("da", "dda", traj)]
    return HotPathInputs(system, traj, integrators)


def unitary_sampling_inputs(systems, U_goal: np.ndarray, T: int, dt: float = 0.2, *, free_time: bool = True,
                            integrator: str = "pade", pade_order: int = 4, seed: int = SEED) -> HotPathInputs:
    """Inputs of `UnitarySamplingProblem` (reference unitary_sampling_problem.jl:44-167): K systems share the
    controls of one merged trajectory with components `Ũ⃗_system_k` (:103-107), one unitary integrator per system
    followed by the two derivative integrators (:134-155)."""
    from .named_trajectory import NamedTrajectory
    from .trajectory_initialization import unitary_geodesic
    rng = np.random.default_rng(seed)
    m = systems[0].n_drives
    N = systems[0].levels
    comps = {}
    names = []
    for k in range(len(systems)):
        name = f"Ũ⃗_system_{k + 1}"
        comps[name] = unitary_geodesic(np.eye(N, dtype=complex), U_goal, T) + 1e-2 * rng.standard_normal((2 * N * N, T))
        names.append(name)
    a = np.zeros((m, T))
    a[:, 1:T - 1] = rng.uniform(-1, 1, size=(m, T - 2))
    comps["a"] = a
    comps["da"] = 0.1 * rng.standard_normal((m, T))
    comps["dda"] = 0.1 * rng.standard_normal((m, T))
    if free_time:
        comps["Δt"] = np.full((1, T), dt)
    from .isomorphisms import operator_to_iso_vec
    traj = NamedTrajectory(comps, controls=("dda", "Δt") if free_time else ("dda",), timestep="Δt" if free_time else dt,
                           goal={nm: operator_to_iso_vec(U_goal) for nm in names})
    cls = UnitaryPadeIntegrator if integrator == "pade" else UnitaryExponentialIntegrator
    kw = {"order": pade_order} if integrator == "pade" else {}
    integrators = [cls(nm, "a", sys_, traj, **kw) for nm, sys_ in zip(names, systems)]
    integrators += [DerivativeIntegrator("a", "da", traj), DerivativeIntegrator("da", "dda", traj)]
    return HotPathInputs(systems[0], traj, integrators)


def quantum_state_sampling_inputs(systems, psi_inits, psi_goals, T: int, dt: float = 0.2, *, free_time: bool = True,
                                  integrator: str = "pade", pade_order: int = 4, seed: int = SEED) -> HotPathInputs:
    """Inputs of `QuantumStateSamplingProblem` (reference quantum_state_sampling_problem.jl:5-120): K systems share the
    controls; system j carries its own copies of the kets, components `ψ̃{i}_system_{j}` (:40-43), one ket integrator per
    (state, system) in system-major order (:98-110), then the two derivative integrators."""
    from .named_trajectory import NamedTrajectory
    rng = np.random.default_rng(seed)
    m, N = systems[0].n_drives, systems[0].levels
    comps, names = {}, []
    lam = np.linspace(0.0, 1.0, T)[None, :]
    for jsys in range(len(systems)):
        row = []
        for i, (p0, p1) in enumerate(zip(psi_inits, psi_goals)):
            p0, p1 = np.asarray(p0, dtype=complex), np.asarray(p1, dtype=complex)
            iso0 = np.concatenate([p0.real, p0.imag])[:, None]
            iso1 = np.concatenate([p1.real, p1.imag])[:, None]
            name = f"ψ̃{i + 1}_system_{jsys + 1}"
            comps[name] = iso0 * (1 - lam) + iso1 * lam + 1e-2 * rng.standard_normal((2 * N, T))
            row.append(name)
        names.append(row)
    a = np.zeros((m, T))
    a[:, 1:T - 1] = rng.uniform(-1, 1, size=(m, T - 2))
    comps["a"] = a
    comps["da"] = 0.1 * rng.standard_normal((m, T))
    comps["dda"] = 0.1 * rng.standard_normal((m, T))
    if free_time:
        comps["Δt"] = np.full((1, T), dt)
    traj = NamedTrajectory(comps, controls=("dda", "Δt") if free_time else ("dda",), timestep="Δt" if free_time else dt)
    cls = QuantumStatePadeIntegrator if integrator == "pade" else QuantumStateExponentialIntegrator
    kw = {"order": pade_order} if integrator == "pade" else {}
    integrators = [cls(nm, "a", sys_, traj, **kw) for row, sys_ in zip(names, systems) for nm in row]
    integrators += [DerivativeIntegrator("a", "da", traj), DerivativeIntegrator("da", "dda", traj)]
    return HotPathInputs(systems[0], traj, integrators)


def unitary_bang_bang_inputs(system: QuantumSystem, U_goal: np.ndarray, T: int, dt: float = 0.2, *, free_time: bool = True,
                             integrator: str = "pade", pade_order: int = 4, control_name: str = "a",
                             seed: int = SEED) -> HotPathInputs:
    """Inputs of `UnitaryBangBangProblem` (reference unitary_bang_bang_problem.jl:73-188): ONE control derivative
    (`initialize_trajectory(..., (a_bounds, da_bounds))`, :102-121), the unitary integrator and a single
    `DerivativeIntegrator(a, da)` (:163-175).  The L1 regulariser on `da` appends two slack components of the size of `da`
    to the trajectory (`L1Regularizer!(constraints, control_names[2], traj, ...)`, :149-152; their names come from
    QuantumCollocationCore and are a choice of this build): the dynamics never read them, they only widen the knots.
    The reference's own test runs this template with `pade_order=12` and `control_name=:u` (:205-215)."""
    rng = np.random.default_rng(seed)
    m = system.n_drives
    base = initialize_trajectory(U_goal, T, dt, m, ([1.0] * m, [1.0] * m), free_time=free_time, rng=rng)
    a, da = control_name, "d" + control_name
    names = {"a": a, "da": da}
    comps = {}
    for nm in base.names:
        comps[names.get(nm, nm)] = np.array(base[nm])
    comps[f"s1_{da}"] = np.maximum(comps[da], 0.0)
    comps[f"s2_{da}"] = np.maximum(-comps[da], 0.0)
    controls = tuple(names.get(c, c) for c in base.controls) + (f"s1_{da}", f"s2_{da}")
    tstep = base.timestep
    traj = NamedTrajectory(comps, controls=controls, timestep=tstep,
                           goal={names.get(k, k): v for k, v in base.goal.items()})
    if integrator == "pade":
        U_int = UnitaryPadeIntegrator("Ũ⃗", a, system, traj, order=pade_order)
    elif integrator == "exponential":
        U_int = UnitaryExponentialIntegrator("Ũ⃗", a, system, traj)
    else:
        raise ValueError("integrator must be one of ('pade', 'exponential')")
    return HotPathInputs(system, traj, [U_int, DerivativeIntegrator(a, da, traj)])


def unitary_direct_sum_inputs(parts, labels=None) -> HotPathInputs:
    """Inputs of `UnitaryDirectSumProblem` (reference unitary_direct_sum_problem.jl:48-186) from the hot-path inputs of its
    member problems (smooth-pulse problems only, :72): the members' trajectories merged with their labels as suffixes
    (`merge([add_suffix(p.trajectory, l) ...])`, :104) and every member's integrators re-pointed at the merged
    trajectory, member after member (:127-130) -- [U_1, D, D, U_2, D, D, ...]: every member keeps its OWN controls, unlike
    the sampling problem.  The reference's test builds the members with `free_time=false` (:196: a constant timestep, no
    `Δt` rows); with free time this build keeps ONE shared, un-suffixed timestep component after the members' components
    (what the merge does with several timestep rows is NamedTrajectories' business and unverified here)."""
    from .integrators import _UnitaryIntegrator
    labels = [str(i + 1) for i in range(len(parts))] if labels is None else [str(l) for l in labels]
    if len(labels) != len(parts) or len(parts) < 2:
        raise ValueError("at least two problems, one label each")
    T = parts[0].traj.T
    free = [isinstance(p.traj.timestep, str) for p in parts]
    if any(free) and not all(free):
        raise ValueError("members must all have a free timestep or all a fixed one")
    if not free[0] and len({float(p.traj.timestep) for p in parts}) != 1:
        raise ValueError("members with fixed timesteps must share the timestep")
    comps, controls, goal = {}, [], {}
    for p, l in zip(parts, labels):
        if p.traj.T != T:
            raise ValueError("the member trajectories must have the same number of knots")
        if "dda" not in p.traj.names:
            raise ValueError("Only smooth pulse problems are supported.")
        for nm in p.traj.names:
            if nm != p.traj.timestep:
                comps[nm + l] = np.array(p.traj[nm])
        controls += [c + l for c in p.traj.controls if c != p.traj.timestep]
        goal.update({k + l: v for k, v in p.traj.goal.items()})
    if free[0]:
        tname = parts[0].traj.timestep
        comps[tname] = np.array(parts[0].traj[tname])
        controls.append(tname)
        tstep = tname
    else:
        tstep = float(parts[0].traj.timestep)
    traj = NamedTrajectory(comps, controls=tuple(controls), timestep=tstep, goal=goal)
    integrators = []
    for p, l in zip(parts, labels):
        for I in p.integrators:
            if isinstance(I, _UnitaryIntegrator):
                kw = {"order": I.order} if isinstance(I, UnitaryPadeIntegrator) else {}
                integrators.append(type(I)(I.state_name + l, I.control_name + l, I.system, traj, **kw))
            elif isinstance(I, DerivativeIntegrator):
                integrators.append(DerivativeIntegrator(I.x + l, I.dx + l, traj))
            else:
                raise NotImplementedError("direct sums of unitary smooth-pulse problems")
    return HotPathInputs(parts[0].system, traj, integrators)
