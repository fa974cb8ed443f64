"""The C restatement (oracle/qc_oracle.c, bench.py's cpu_baseline) against the numpy oracle, and
both against the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle_bridge import random_problem

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("N,m,order", [(1, 1, 4), (2, 2, 2), (2, 3, 4), (3, 2, 6), (4, 4, 4), (2, 2, 12), (8, 6, 4)])
@pytest.mark.parametrize("free_time", [True, False])
def test_c_oracle_matches_numpy_oracle(oracle, coracle, N, m, order, free_time):
    prob, Z = random_problem(oracle, N=N, m=m, T=4, order=order, free_time=free_time, seed=N * 31 + order)
    co = coracle.COracle(prob, threads=2)
    F, J = co.F_dF(Z)
    np.testing.assert_allclose(F, oracle.F(prob, Z), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=1e-12, atol=1e-13)
    assert co.jac_nnz == oracle.jac_nnz_interval(prob) and co.hess_nnz + co.hess_pad == oracle.hess_nnz_interval(prob)
    mu = np.random.default_rng(1).standard_normal(prob.n_rows)
    np.testing.assert_allclose(co.mu_d2F(Z, mu), oracle.mu_d2F(prob, Z, mu), rtol=1e-11, atol=1e-12)


def test_c_oracle_layouts_and_exponential(oracle, coracle):
    for layout in ("shuffled", "script"):
        prob, Z = random_problem(oracle, N=2, m=2, T=4, layout=layout, seed=3)
        F, J = coracle.COracle(prob).F_dF(Z)
        np.testing.assert_allclose(F, oracle.F(prob, Z), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=1e-12, atol=1e-13)
    for ft in (True, False):
        prob, Z = random_problem(oracle, N=2, m=2, T=4, free_time=ft, integrator=oracle.EXPONENTIAL, seed=4)
        F, J = coracle.COracle(prob).F_dF(Z)
        np.testing.assert_allclose(F, oracle.F(prob, Z), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("N,m", [(1, 1), (2, 2), (2, 3), (3, 2), (4, 4), (8, 6)])
@pytest.mark.parametrize("free_time", [True, False])
def test_c_oracle_exponential_hessian(oracle, coracle, N, m, free_time):
    """Forward-mode chains through scaling and squaring (C) against the 3n x 3n block-triangular exponential (numpy)."""
    T = 3 if N == 8 else 4
    for layout in (("standard",) if N > 2 else ("standard", "shuffled", "script")):
        prob, Z = random_problem(oracle, N=N, m=m, T=T, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=70 + N + m, layout=layout)
        co = coracle.COracle(prob, threads=2)
        assert co.hess_nnz + co.hess_pad == oracle.hess_nnz_interval(prob) > 0
        mu = np.random.default_rng(5).standard_normal(prob.n_rows)
        np.testing.assert_allclose(co.mu_d2F(Z, mu), oracle.mu_d2F(prob, Z, mu), rtol=1e-11, atol=1e-12)
    # a large step (several squarings) and non-Hermitian Hamiltonians
    prob, Z = random_problem(oracle, N=N, m=m, T=3, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=71, hermitian=False)
    if free_time:
        Z[prob.off_dt::prob.zdim] = 1.3
    else:
        prob.dt_fixed = 1.3
    mu = np.random.default_rng(6).standard_normal(prob.n_rows)
    ref = oracle.mu_d2F(prob, Z, mu)
    np.testing.assert_allclose(coracle.COracle(prob).mu_d2F(Z, mu), ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max())


def test_oracles_reproduce_the_golden_vectors(oracle, coracle):
    import json
    fx = json.load(open(os.path.join(GOLD, "named_trajectory_type_1.json")))
    gold = np.load(os.path.join(GOLD, "fixture_outputs.npz"))
    data = np.array(fx["data"])
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    Y = np.array([[0, -1j], [1j, 0]])
    Zp = np.array([[1, 0], [0, -1]], dtype=complex)
    prob = oracle.Problem(N=2, m=2, T=5, zdim=15, off_U=0, off_a=8, off_dt=14, G_drift=oracle.generator(0.1 * Zp),
                          G_drives=np.array([oracle.generator(X), oracle.generator(Y)]), order=4,
                          derivs=[oracle.DerivSpec(8, 10, 2), oracle.DerivSpec(10, 12, 2)])
    Zv = data.reshape(-1, order="F")
    mu = np.ones(prob.n_rows)
    np.testing.assert_allclose(oracle.F(prob, Zv), gold["F"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(oracle.dF(prob, Zv), gold["dF"], rtol=1e-14, atol=1e-16)
    # the golden Hessian vector is the default layout (exactly the structural entries); hess_align = 16 pads every interval's 58 values to 64
    np.testing.assert_allclose(oracle.mu_d2F(prob, Zv, mu), gold["mu_d2F"], rtol=1e-13, atol=1e-16)
    prob.hess_align = 16
    Hp = oracle.mu_d2F(prob, Zv, mu).reshape(prob.T - 1, -1)
    own = len(oracle.hess_structure_local(prob))
    assert Hp.shape[1] == 64 and own == 58 and not Hp[:, own:].any()
    np.testing.assert_allclose(Hp[:, :own].reshape(-1), gold["mu_d2F"], rtol=1e-13, atol=1e-16)
    hr, hc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr.reshape(prob.T - 1, -1)[:, :own].reshape(-1), gold["mu_d2F_rows"])
    np.testing.assert_array_equal(hr.reshape(prob.T - 1, -1)[:, own:], np.repeat(hr.reshape(prob.T - 1, -1)[:, :1], 6, axis=1))
    prob.hess_align = 1
    np.testing.assert_allclose(oracle.mu_d2F(prob, Zv, mu), gold["mu_d2F"], rtol=1e-13, atol=1e-16)
    r, c = oracle.jac_structure(prob)
    np.testing.assert_array_equal(r, gold["dF_rows"])
    np.testing.assert_array_equal(c, gold["dF_cols"])
    co = coracle.COracle(prob)
    F, J = co.F_dF(Zv)
    np.testing.assert_allclose(F, gold["F"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(J, gold["dF"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(co.mu_d2F(Zv, mu), gold["mu_d2F"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("K", [1, 2, 5])
def test_c_oracle_kets(oracle, coracle, K):
    for integ in (oracle.PADE, oracle.EXPONENTIAL):
        prob, Z = random_problem(oracle, N=3, m=2, T=4, order=4, seed=60 + K, integrator=integ, ncol=K)
        co = coracle.COracle(prob)
        F, J = co.F_dF(Z)
        np.testing.assert_allclose(F, oracle.F(prob, Z), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=1e-11, atol=1e-12)
        assert co.jac_nnz == oracle.jac_nnz_interval(prob)
        mu = np.random.default_rng(3).standard_normal(prob.n_rows)
        np.testing.assert_allclose(co.mu_d2F(Z, mu), oracle.mu_d2F(prob, Z, mu), rtol=1e-11, atol=1e-12)


def test_oracle_reproduces_the_8f_golden_vectors(oracle):
    """tests/golden/fixture_outputs_8f.npz (build-oracle outputs on the reference's data fixture for the §8f rows:
    exponential integrator, rollout, fidelity, regularisers + minimum-time term) is reproducible from the oracle."""
    import json
    fx = json.load(open(os.path.join(GOLD, "named_trajectory_type_1.json")))
    gold = np.load(os.path.join(GOLD, "fixture_outputs_8f.npz"))
    Zv = np.array(fx["data"]).reshape(-1, order="F")
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    Y = np.array([[0, -1j], [1j, 0]])
    Zp = np.array([[1, 0], [0, -1]], dtype=complex)
    pe = oracle.Problem(N=2, m=2, T=5, zdim=15, off_U=0, off_a=8, off_dt=14, G_drift=oracle.generator(0.1 * Zp),
                        G_drives=np.array([oracle.generator(X), oracle.generator(Y)]), integrator=oracle.EXPONENTIAL,
                        derivs=[oracle.DerivSpec(8, 10, 2), oracle.DerivSpec(10, 12, 2)])
    np.testing.assert_allclose(oracle.F(pe, Zv), gold["exp_F"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(oracle.dF(pe, Zv), gold["exp_dF"], rtol=1e-13, atol=1e-15)
    roll = oracle.rollout(pe, Zv, gold["init"])
    np.testing.assert_allclose(roll, gold["rollout"], rtol=1e-13, atol=1e-15)
    f, g, H = oracle.fidelity_value_grad_hess(roll[:, -1], gold["goal"])
    assert abs(f - float(gold["fidelity"])) < 1e-14
    np.testing.assert_allclose(g, gold["fidelity_grad"], rtol=1e-12, atol=1e-14)
    tm = oracle.Terms(T=5, zdim=15, off_dt=14, reg_index=gold["terms_index"], reg_R=gold["terms_R"], D=1.5, n_mt=4, dt_scaled=True)   # the golden vectors were made with the dt-scaled weighting
    assert abs(oracle.terms_value(tm, Zv) - float(gold["terms_J"])) < 1e-15
    np.testing.assert_allclose(oracle.terms_grad(tm, Zv), gold["terms_grad"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(oracle.terms_hess(tm, Zv), gold["terms_hess"], rtol=1e-14, atol=1e-16)
