"""Final-knot fidelity: the oracle against finite differences and known answers (CPU), the
`qc_fidelity_*` kernel against the oracle (GPU)."""
import numpy as np
import pytest
import scipy.linalg as sla


def rand_unitary(N, rng):
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    return sla.expm(1j * (A + A.conj().T) / 2)


def test_fidelity_known_answers(oracle):
    rng = np.random.default_rng(0)
    for N in (2, 4, 8):
        U = rand_unitary(N, rng)
        u = oracle.operator_to_iso_vec(U)
        assert abs(oracle.iso_vec_unitary_fidelity(u, u) - 1.0) < 1e-14
        # global phase invariance
        assert abs(oracle.iso_vec_unitary_fidelity(oracle.operator_to_iso_vec(np.exp(0.7j) * U), u) - 1.0) < 1e-14
        # orthogonal gate: X vs I on one qubit has zero overlap
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    assert oracle.iso_vec_unitary_fidelity(oracle.operator_to_iso_vec(X), oracle.operator_to_iso_vec(np.eye(2))) < 1e-15
    # subspace: a 3-level gate that is H on levels {0,1} and anything on level 2
    H3 = np.eye(3, dtype=complex)
    H3[:2, :2] = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    V = H3.copy()
    V[2, 2] = np.exp(1.3j)
    f_full = oracle.iso_vec_unitary_fidelity(oracle.operator_to_iso_vec(V), oracle.operator_to_iso_vec(H3))
    f_sub = oracle.iso_vec_unitary_fidelity(oracle.operator_to_iso_vec(V), oracle.operator_to_iso_vec(H3), subspace=[0, 1])
    assert f_full < 1.0 - 1e-3 and abs(f_sub - 1.0) < 1e-14


@pytest.mark.parametrize("N,sub", [(2, None), (3, [0, 1]), (4, None), (4, [0, 2, 3])])
def test_fidelity_derivatives_vs_finite_differences(oracle, N, sub):
    rng = np.random.default_rng(N)
    goal = oracle.operator_to_iso_vec(rand_unitary(N, rng))
    u = oracle.operator_to_iso_vec(rand_unitary(N, rng)) + 0.05 * rng.standard_normal(2 * N * N)
    F, g, Hm = oracle.fidelity_value_grad_hess(u, goal, sub)
    eps = 1e-6
    gfd = np.array([(oracle.iso_vec_unitary_fidelity(u + eps * e, goal, sub) - oracle.iso_vec_unitary_fidelity(u - eps * e, goal, sub)) / (2 * eps)
                    for e in np.eye(u.size)])
    np.testing.assert_allclose(g, gfd, rtol=1e-7, atol=1e-9)
    Hfd = np.array([(oracle.fidelity_value_grad_hess(u + eps * e, goal, sub)[1] - oracle.fidelity_value_grad_hess(u - eps * e, goal, sub)[1]) / (2 * eps)
                    for e in np.eye(u.size)])
    np.testing.assert_allclose(Hm, Hfd, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(Hm, Hm.T, atol=1e-15)
    l, gl, Hl = oracle.infidelity_value_grad_hess(u, goal, sub)
    assert abs(l - abs(1 - F)) < 1e-15
    np.testing.assert_allclose(gl, -np.sign(1 - F) * g)


@pytest.mark.gpu
@pytest.mark.parametrize("N,sub", [(2, None), (3, [0, 1]), (4, None), (8, None), (8, [0, 1, 2, 3]), (16, None)])
def test_fidelity_kernel_matches_oracle(qc, oracle, N, sub):
    from qcolloc_amd.objectives import _Fidelity
    rng = np.random.default_rng(10 + N)
    goal = oracle.operator_to_iso_vec(rand_unitary(N, rng))
    u = oracle.operator_to_iso_vec(rand_unitary(N, rng)) + 0.05 * rng.standard_normal(2 * N * N)
    f = _Fidelity(goal, sub)
    F, L, g, H = f.eval(u)
    Fr, gr, Hr = oracle.fidelity_value_grad_hess(u, goal, sub)
    assert abs(F - Fr) < 1e-13 and abs(L - abs(1 - Fr)) < 1e-13
    np.testing.assert_allclose(g, gr, rtol=1e-11, atol=1e-13)
    s = u.size
    Hd = np.zeros((s, s))
    jj = np.repeat(np.arange(s), np.arange(1, s + 1))
    ii = np.concatenate([np.arange(j + 1) for j in range(s)])
    Hd[ii, jj] = H
    np.testing.assert_allclose(Hd, np.triu(Hr), rtol=1e-10, atol=1e-12)
    assert abs(qc.iso_vec_unitary_fidelity(u, goal, sub) - Fr) < 1e-13
    f.close()


@pytest.mark.gpu
def test_objective_and_constraint_mirror(qc, oracle):
    inp = qc.config_inputs(2, T=12)
    traj = inp.traj
    Z = traj.datavec
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=100.0)
    con = qc.FinalUnitaryFidelityConstraint("Ũ⃗", 0.99, traj)
    uT = traj["Ũ⃗"][:, -1]
    goal = traj.goal["Ũ⃗"]
    l, gl, Hl = oracle.infidelity_value_grad_hess(uT, goal)
    Fr, gF, HF = oracle.fidelity_value_grad_hess(uT, goal)
    assert abs(obj.L(Z) - 100.0 * l) < 1e-11
    np.testing.assert_allclose(obj.grad_L(Z), 100.0 * gl, rtol=1e-10, atol=1e-12)
    assert obj.state_indices[0] == 11 * traj.dim and obj.state_indices.size == 32
    r, c = obj.hess_structure
    assert (r <= c).all() and r.size == 32 * 33 // 2
    Hd = np.zeros((Z.size, Z.size))
    Hd[r, c] = getattr(obj, "∂²L")(Z)
    first = obj.first
    np.testing.assert_allclose(Hd[first:first + 32, first:first + 32], np.triu(100.0 * Hl), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(con.g(Z), [Fr - 0.99], atol=1e-13)
    np.testing.assert_allclose(con.dg(Z), gF, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(con.mu_d2g(Z, [2.5]), 2.5 * HF[np.triu_indices(32)][np.lexsort((np.triu_indices(32)[0], np.triu_indices(32)[1]))], rtol=1e-9, atol=1e-11)
    obj.close()
    con.close()
