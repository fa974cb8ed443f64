"""Density-operator integrator: the Lindblad superoperator of `OpenQuantumSystem` against a direct
integration of the master equation (CPU), descriptor / structure against the oracle (CPU), kernel parity (GPU)."""
import numpy as np
import pytest
import scipy.integrate as si
import scipy.linalg as sla

from oracle_bridge import problem_from_inputs

RTOL = 1e-10   # north_star tolerance on residual / Jacobian entries


def open_system(qc, nq, gamma=0.05):
    base = qc.multi_qubit_system(nq)
    N = base.levels
    sm = np.array([[0, 1], [0, 0]], dtype=complex)
    diss = []
    for q in range(nq):
        ops = [np.eye(2, dtype=complex)] * nq
        ops[q] = sm
        L = ops[0]
        for o in ops[1:]:
            L = np.kron(L, o)
        diss.append(np.sqrt(gamma * (q + 1)) * L)
    return qc.OpenQuantumSystem(base.H_drift, base.H_drives, diss)


def test_lindblad_superoperator_matches_master_equation(qc):
    rng = np.random.default_rng(0)
    sys_ = open_system(qc, 1, gamma=0.3)
    a = rng.uniform(-1, 1, sys_.n_drives)
    H = sys_.H_drift + sum(ak * Hk for ak, Hk in zip(a, sys_.H_drives))
    A = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
    rho0 = A @ A.conj().T
    rho0 /= np.trace(rho0)

    def rhs(t, y):
        rho = y.reshape(2, 2)
        d = -1j * (H @ rho - rho @ H)
        for L in sys_.dissipation_operators:
            LdL = L.conj().T @ L
            d = d + L @ rho @ L.conj().T - 0.5 * (LdL @ rho + rho @ LdL)
        return d.reshape(-1)

    tf = 0.7
    sol = si.solve_ivp(rhs, (0, tf), rho0.reshape(-1).astype(complex), rtol=1e-11, atol=1e-13)
    rho_ode = sol.y[:, -1].reshape(2, 2)
    v = sla.expm(tf * sys_.G(a)) @ qc.density_to_iso_vec(rho0)
    rho_exp = qc.iso_vec_to_density(v)
    np.testing.assert_allclose(rho_exp, rho_ode, atol=1e-9)
    assert abs(np.trace(rho_exp) - 1) < 1e-12 and np.allclose(rho_exp, rho_exp.conj().T, atol=1e-12)
    # no dissipators: the superoperator reproduces unitary conjugation
    closed = qc.OpenQuantumSystem(sys_.H_drift, sys_.H_drives, [])
    U = sla.expm(-1j * tf * H)
    np.testing.assert_allclose(qc.iso_vec_to_density(sla.expm(tf * closed.G(a)) @ qc.density_to_iso_vec(rho0)), U @ rho0 @ U.conj().T, atol=1e-12)
    # iso helpers
    np.testing.assert_array_equal(qc.iso_vec_to_density(qc.density_to_iso_vec(rho0)), rho0)
    np.testing.assert_allclose(qc.iso_operator(-1j * H), qc.iso_generator(H))


@pytest.mark.parametrize("nq,free_time", [(1, True), (2, False)])
def test_density_descriptor_and_structure(qc, oracle, nq, free_time):
    sys_ = open_system(qc, nq)
    N = sys_.levels
    psi = np.zeros(N, dtype=complex)
    psi[-1] = 1
    inp = qc.density_operator_smooth_pulse_inputs(sys_, np.eye(N) / N, psi, 6, free_time=free_time)
    desc, keep = qc.make_desc(inp.integrators, inp.traj)
    assert desc.N == N * N and desc.state_cols == 1 and desc.integrator == qc._lib.QC_EXPONENTIAL
    prob = problem_from_inputs(inp)
    dims = qc.desc_dims(desc)
    assert dims.ddim == prob.ddim == 2 * N * N + 2 * sys_.n_drives
    jr, jc, hr, hc = qc.desc_structures(desc)
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    # the Lindblad step is an exponential integrator with one state column: its Hessian has no entry at knot t+1
    orr, oc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    assert dims.hess_nnz_interval == oracle.hess_nnz_interval(prob) > 0
    # the oracle's residual vanishes on an exact Lindblad propagation
    Z = inp.traj.datavec.copy().reshape(inp.traj.T, inp.traj.dim)
    off = inp.traj.offset("ρ⃗̃")
    for t in range(inp.traj.T - 1):
        a = Z[t, inp.traj.offset("a"):inp.traj.offset("a") + sys_.n_drives]
        h = Z[t, inp.traj.offset("Δt")] if free_time else float(inp.traj.timestep)
        Z[t + 1, off:off + 2 * N * N] = sla.expm(h * sys_.G(a)) @ Z[t, off:off + 2 * N * N]
    F = oracle.F(prob, Z.ravel()).reshape(inp.traj.T - 1, prob.ddim)
    assert np.abs(F[:, :2 * N * N]).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("nq,T,free_time", [(1, 9, True), (1, 5, False), (2, 6, True)])
def test_density_kernel_matches_oracle(qc, oracle, nq, T, free_time):
    sys_ = open_system(qc, nq)
    N = sys_.levels
    psi = np.zeros(N, dtype=complex)
    psi[0] = 1
    inp = qc.density_operator_smooth_pulse_inputs(sys_, np.eye(N) / N, psi, T, free_time=free_time)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert dyn.kernel == "mfma"      # 1 qubit: N^2 = 4 levels, padded 2N = 16 tile; 2 qubits: 16 levels, the 2N = 32 kernel
    F, J = dyn.F_dF(Z)
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    jr, jc = dyn.dF_structure
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    # mu_d2F of the Lindblad step (density_operator_smooth_pulse_problem.jl:104-106 builds the integrator; :68 passes eval_hessian on)
    mu = np.random.default_rng(nq).standard_normal(prob.n_rows)
    Hr = oracle.mu_d2F(prob, Z, mu)
    np.testing.assert_allclose(dyn.mu_d2F(Z, mu), Hr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Hr).max()))
    dyn.close()
