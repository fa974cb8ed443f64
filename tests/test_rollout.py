"""Rollouts.  CPU: the oracle's rollout against scipy's expm and unitarity / trace invariants.
GPU: `qc_rollout` (three-level scan of propagators) against the oracle for unitaries, kets and density operators,
across chunk boundaries, and the rollout fidelity."""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle_bridge import problem_from_inputs


def test_oracle_rollout_matches_scipy_and_stays_unitary(qc, oracle):
    inp = qc.config_inputs(2, T=12)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    N = inp.system.levels
    init = qc.operator_to_iso_vec(np.eye(N, dtype=complex))
    R = oracle.rollout(prob, Z, init)
    assert R.shape == (2 * N * N, 12)
    U = np.eye(N, dtype=complex)
    Zm = Z.reshape(12, -1)
    for t in range(11):
        a = Zm[t, prob.off_a:prob.off_a + prob.m]
        H = inp.system.H_drift + sum(x * Hk for x, Hk in zip(a, inp.system.H_drives))
        U = sla.expm(-1j * Zm[t, prob.off_dt] * H) @ U
        np.testing.assert_allclose(qc.iso_vec_to_operator(R[:, t + 1]), U, atol=1e-12)
    np.testing.assert_allclose(U.conj().T @ U, np.eye(N), atol=1e-12)
    # the rolled-out states satisfy the exponential-integrator constraint exactly
    inp_e = qc.unitary_smooth_pulse_inputs(inp.system, qc.GATES["CX"], 12, integrator="exponential")
    pe = problem_from_inputs(inp_e)
    Ze = inp_e.traj.datavec.copy().reshape(12, -1)
    Re = oracle.rollout(pe, Ze.ravel(), init)
    Ze[:, pe.off_U:pe.off_U + pe.s] = Re.T
    F = oracle.F(pe, Ze.ravel()).reshape(11, -1)
    assert np.abs(F[:, :pe.s]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T", [(1, 2), (1, 5), (1, 50), (2, 37), (3, 101), (5, 9)])
def test_unitary_rollout_kernel(qc, oracle, cfg, T):
    inp = qc.config_inputs(cfg, T=T)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    N = inp.system.levels
    rng = np.random.default_rng(T)
    init = qc.operator_to_iso_vec(sla.expm(1j * (lambda A: (A + A.conj().T) / 2)(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)))))
    ref = oracle.rollout(prob, Z, init)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)          # a Pade handle: the rollout is exponential regardless
    got = dyn.rollout(Z, init)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-11)
    dyn.close()
    a = inp.traj["a"]
    dts = inp.traj["Δt"].ravel()
    got2 = qc.unitary_rollout(init, a, dts, inp.system)
    np.testing.assert_allclose(got2, ref, rtol=1e-10, atol=1e-11)


@pytest.mark.gpu
def test_rollout_large_generator_norm_and_fixed_time(qc, oracle):
    """Strong drives (several squarings) and a scalar timestep."""
    base = qc.multi_qubit_system(2)
    sys_ = qc.QuantumSystem(40.0 * base.H_drift, [25.0 * H for H in base.H_drives])
    rng = np.random.default_rng(5)
    T = 30
    a = rng.uniform(-1, 1, (sys_.n_drives, T))
    init = qc.operator_to_iso_vec(np.eye(4, dtype=complex))
    got = qc.unitary_rollout(init, a, 0.37, sys_)
    U = np.eye(4, dtype=complex)
    for t in range(T - 1):
        H = sys_.H_drift + sum(x * Hk for x, Hk in zip(a[:, t], sys_.H_drives))
        U = sla.expm(-1j * 0.37 * H) @ U
        np.testing.assert_allclose(qc.iso_vec_to_operator(got[:, t + 1]), U, atol=1e-9)


@pytest.mark.gpu
def test_ket_and_density_rollouts(qc, oracle):
    import test_density
    base = qc.multi_qubit_system(2)
    rng = np.random.default_rng(2)
    T = 26
    a = rng.uniform(-1, 1, (base.n_drives, T))
    dts = rng.uniform(0.1, 0.3, T)
    psi = rng.standard_normal(4) + 1j * rng.standard_normal(4)
    psi /= np.linalg.norm(psi)
    got = qc.rollout(np.concatenate([psi.real, psi.imag]), a, dts, base)
    v = psi.copy()
    for t in range(T - 1):
        H = base.H_drift + sum(x * Hk for x, Hk in zip(a[:, t], base.H_drives))
        v = sla.expm(-1j * dts[t] * H) @ v
    np.testing.assert_allclose(got[:4, -1] + 1j * got[4:, -1], v, atol=1e-11)
    osys = test_density.open_system(qc, 2, gamma=0.1)
    rho0 = np.outer(psi, psi.conj())
    R = qc.open_rollout(qc.density_to_iso_vec(rho0), a, dts, osys)
    x = qc.density_to_iso_vec(rho0)
    for t in range(T - 1):
        x = sla.expm(dts[t] * osys.G(a[:, t])) @ x
    np.testing.assert_allclose(R[:, -1], x, atol=1e-11)
    rho = qc.iso_vec_to_density(R[:, -1])
    assert abs(np.trace(rho) - 1) < 1e-11 and np.linalg.eigvalsh((rho + rho.conj().T) / 2).min() > -1e-10


@pytest.mark.gpu
def test_unitary_rollout_fidelity(qc, oracle):
    inp = qc.config_inputs(1, T=40)
    f = qc.unitary_rollout_fidelity(inp.traj, inp.system)
    prob = problem_from_inputs(inp)
    R = oracle.rollout(prob, inp.traj.datavec, qc.operator_to_iso_vec(np.eye(2, dtype=complex)))
    assert abs(f - oracle.iso_vec_unitary_fidelity(R[:, -1], inp.traj.goal["Ũ⃗"])) < 1e-12
    assert 0.0 <= f <= 1.0 + 1e-12


@pytest.mark.gpu
def test_rollout_is_bit_reproducible(qc, oracle):
    """Every knot's state is written by ONE workgroup: until round 4 the first knot of a scan chunk was written twice (by the chunk that
    starts there and by the one that ends there, products in different orders, one ulp apart) and repeated calls differed in the last
    bit at the chunk boundaries."""
    for cfg, T in [(2, 9), (2, 50), (3, 101), (1, 17)]:
        inp = qc.config_inputs(cfg, T=T)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        N = inp.system.levels
        init = qc.operator_to_iso_vec(np.eye(N, dtype=complex))
        first = dyn.rollout(inp.traj.datavec, init)
        for _ in range(6):
            np.testing.assert_array_equal(dyn.rollout(inp.traj.datavec, init), first)
        dyn.close()


def test_control_guess_derivatives_and_linear_interpolation(qc):
    """Host side of `initialize_trajectory(...; a_guess, geodesic=false)` (reference trajectory_initialization.jl:176-188,225-244)."""
    from qcolloc_amd.trajectory_initialization import control_derivatives_from_guess, unitary_linear_interpolation
    rng = np.random.default_rng(2)
    a = rng.standard_normal((3, 9))
    dts = rng.uniform(0.1, 0.3, 9)
    a0, da, dda = control_derivatives_from_guess(a, dts, 2)
    np.testing.assert_array_equal(a0, a)
    np.testing.assert_allclose(a[:, 1:] - a[:, :-1] - dts[:-1] * da[:, :-1], 0.0, atol=1e-15)         # DerivativeIntegrator(a, da) rows
    np.testing.assert_allclose(da[:, 1:] - da[:, :-1] - dts[:-1] * dda[:, :-1], 0.0, atol=1e-14)     # DerivativeIntegrator(da, dda)
    U = qc.GATES["CNOT"]
    lin = unitary_linear_interpolation(np.eye(4, dtype=complex), U, 5)
    np.testing.assert_array_equal(lin[:, 0], qc.operator_to_iso_vec(np.eye(4, dtype=complex)))
    np.testing.assert_allclose(lin[:, -1], qc.operator_to_iso_vec(U), atol=1e-15)
    np.testing.assert_allclose(lin[:, 2], 0.5 * (lin[:, 0] + lin[:, -1]), atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("nq,T,free_time", [(1, 20, True), (2, 33, False), (3, 50, True)])
def test_trajectory_from_a_control_guess_satisfies_the_exponential_dynamics(qc, oracle, nq, T, free_time):
    """`initialize_trajectory(U_goal, T, dt, ...; a_guess, system)` rolls the guess out (reference trajectory_initialization.jl:422-426):
    the rollout kernel's states are a zero of the exponential integrator's residual kernel (x_{t+1} = exp(dt G(a_t)) x_t on both
    sides, two independent implementations: scan of propagators vs per-interval expm), and the differenced controls a zero of the
    derivative rows."""
    from qcolloc_amd.trajectory_initialization import initialize_trajectory
    system = qc.multi_qubit_system(nq)
    m = system.n_drives
    rng = np.random.default_rng(nq)
    a_guess = 0.5 * np.sin(np.linspace(0, 3, T))[None, :] * rng.uniform(0.5, 1.0, (m, 1))
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI"}[nq]
    traj = initialize_trajectory(qc.GATES[gate], T, 0.2, m, ([1.0] * m, [np.inf] * m, [1.0] * m), free_time=free_time,
                                 a_guess=a_guess, system=system)
    integ = [qc.UnitaryExponentialIntegrator("Ũ⃗", "a", system, traj), qc.DerivativeIntegrator("a", "da", traj), qc.DerivativeIntegrator("da", "dda", traj)]
    dyn = qc.QuantumDynamics(integ, traj, eval_hessian=False)
    F = dyn.F(traj.datavec)
    assert np.abs(F).max() < 5e-13, np.abs(F).max()
    dyn.close()
    # ... and of the order-12 Pade residual to its truncation error, of the order-4 residual only to O(dt^5)
    integ12 = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", system, traj, order=12)] + integ[1:]
    dyn12 = qc.QuantumDynamics(integ12, traj)
    assert np.abs(dyn12.F(traj.datavec)).max() < 1e-12
    dyn12.close()
    with pytest.raises(ValueError):
        initialize_trajectory(qc.GATES[gate], T, 0.2, m, ([1.0] * m, [np.inf] * m, [1.0] * m), a_guess=a_guess)     # no system


@pytest.mark.gpu
def test_ket_rollout_fidelity(qc, oracle):
    """`rollout_fidelity(prob.trajectory, sys; state_name)` (reference quantum_state_smooth_pulse_problem.jl:247-249,
    quantum_state_sampling_problem.jl:187-189) against a scipy rollout of the same controls."""
    system = qc.multi_qubit_system(2)
    basis = np.eye(4, dtype=complex)
    inp = qc.quantum_state_smooth_pulse_inputs(system, [basis[:, 0], basis[:, 1]], [basis[:, 3], (basis[:, 0] + 1j * basis[:, 2]) / np.sqrt(2)], 31)
    a, dts = inp.traj["a"], inp.traj["Δt"].ravel()
    for k, name in enumerate(("ψ̃1", "ψ̃2")):
        v = basis[:, k].copy()
        for t in range(inp.traj.T - 1):
            H = system.H_drift + sum(x * Hk for x, Hk in zip(a[:, t], system.H_drives))
            v = sla.expm(-1j * dts[t] * H) @ v
        goal = inp.traj.goal[name]
        want = abs(np.vdot(goal[:4] + 1j * goal[4:], v)) ** 2
        got = qc.rollout_fidelity(inp.traj, system, state_name=name)
        assert abs(got - want) < 1e-12, (name, got, want)
