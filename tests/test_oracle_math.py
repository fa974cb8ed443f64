"""Pins the CPU oracle against MATHEMATICS (the reference holds no numbers for this path:
SURVEY.md section 8c): complex-step derivatives, scipy expm, mpmath 50-digit spot checks, Pade order
of accuracy, and the one data fixture the reference's tests ship (test/test_utils.jl:54-70)."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest
import scipy.linalg as sla

from oracle_bridge import random_problem

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def complex_step_jac(f, x, h=1e-30):
    cols = []
    for i in range(x.size):
        xp = x.astype(complex)
        xp[i] += 1j * h
        cols.append(np.imag(f(xp)) / h)
    return np.stack(cols, axis=1)


def test_pade_coefficients(oracle):
    # SURVEY A.2 table
    expect = {
        4: ["1/2", "1/12"],
        6: ["1/2", "1/10", "1/120"],
        8: ["1/2", "3/28", "1/84", "1/1680"],
        10: ["1/2", "1/9", "1/72", "1/1008", "1/30240"],
        12: ["1/2", "5/44", "1/66", "1/792", "1/15840", "1/665280"],
    }
    for order, fr in expect.items():
        c = oracle.pade_coeffs(order)
        assert c[0] == 1.0
        np.testing.assert_allclose(c[1:], [float(Fraction(f)) for f in fr], rtol=1e-15)


def test_isomorphism(oracle):
    rng = np.random.default_rng(1)
    N = 4
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    H = (A + A.conj().T) / 2
    G = oracle.generator(H)
    np.testing.assert_allclose(G.T, -G, atol=1e-15)           # Hermitian H  =>  antisymmetric G
    U = sla.expm(-1j * 0.3 * H)
    iso = np.vstack([U.real, U.imag])
    np.testing.assert_allclose(sla.expm(0.3 * G) @ np.vstack([np.eye(N), np.zeros((N, N))]), iso, atol=1e-14)
    v = oracle.operator_to_iso_vec(U)
    np.testing.assert_allclose(v.reshape(2 * N, N, order="F"), iso, atol=0)
    np.testing.assert_allclose(oracle.iso_vec_to_operator(v), U, atol=0)
    # vec(B X) = (I_N (x) B) vec X, column-major
    B = rng.standard_normal((2 * N, 2 * N))
    np.testing.assert_allclose(np.kron(np.eye(N), B) @ v, (B @ iso).reshape(-1, order="F"), atol=1e-14)


@pytest.mark.parametrize("order", [2, 4, 6, 8])
def test_pade_order_of_accuracy(oracle, order):
    prob, Z = random_problem(oracle, N=2, m=2, T=2, order=order, seed=3)
    zd = prob.zdim
    z0 = Z[:zd].copy()
    errs = []
    hs = [0.4, 0.2, 0.1]
    for h in hs:
        z0[prob.off_dt] = h
        G = prob.G_drift + np.tensordot(z0[prob.off_a:prob.off_a + prob.m], prob.G_drives, axes=(0, 0))
        U0 = z0[prob.off_U:prob.off_U + prob.s].reshape(prob.n, prob.N, order="F")
        z1 = Z[zd:2 * zd].copy()
        z1[prob.off_U:prob.off_U + prob.s] = (sla.expm(h * G) @ U0).reshape(-1, order="F")
        errs.append(np.linalg.norm(oracle.interval_residual(prob, z0, z1)[:prob.s]))
    slope = np.polyfit(np.log(hs), np.log(errs), 1)[0]
    assert abs(slope - (order + 1)) < 0.35, (slope, errs)


def test_pade12_residual_at_roundoff(oracle):
    prob, Z = random_problem(oracle, N=2, m=2, T=2, order=12, seed=4)
    zd = prob.zdim
    z0, z1 = Z[:zd].copy(), Z[zd:].copy()
    z0[prob.off_dt] = 0.2
    G = prob.G_drift + np.tensordot(z0[prob.off_a:prob.off_a + prob.m], prob.G_drives, axes=(0, 0))
    U0 = z0[prob.off_U:prob.off_U + prob.s].reshape(prob.n, prob.N, order="F")
    z1[prob.off_U:prob.off_U + prob.s] = (sla.expm(0.2 * G) @ U0).reshape(-1, order="F")
    assert np.linalg.norm(oracle.interval_residual(prob, z0, z1)[:prob.s]) < 1e-13


@pytest.mark.parametrize("order", [2, 4, 6, 10, 12])
@pytest.mark.parametrize("free_time", [True, False])
@pytest.mark.parametrize("layout", ["standard", "shuffled", "script"])
def test_jacobian_vs_complex_step(oracle, order, free_time, layout):
    prob, Z = random_problem(oracle, N=2, m=3, T=2, order=order, free_time=free_time, seed=5, layout=layout)
    zd = prob.zdim
    zz = Z[:2 * zd]
    J = oracle.interval_jacobian_dense(prob, zz[:zd], zz[zd:])
    Jcs = complex_step_jac(lambda x: oracle.interval_residual(prob, x[:zd], x[zd:]), zz)
    np.testing.assert_allclose(J, Jcs, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("order", [2, 4, 6, 8, 12])
@pytest.mark.parametrize("free_time", [True, False])
@pytest.mark.parametrize("layout", ["standard", "shuffled"])
def test_hessian_vs_complex_step(oracle, order, free_time, layout):
    prob, Z = random_problem(oracle, N=2, m=3, T=2, order=order, free_time=free_time, seed=6, layout=layout)
    zd = prob.zdim
    zz = Z[:2 * zd]
    mu = np.random.default_rng(7).standard_normal(prob.ddim)
    Hd = oracle.interval_hessian_dense(prob, zz[:zd], zz[zd:], mu)
    Hcs = complex_step_jac(lambda x: oracle.interval_jacobian_dense(prob, x[:zd], x[zd:]).T @ mu, zz)
    np.testing.assert_allclose(Hd, Hcs, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(Hd, Hd.T, atol=0)
    # U-U blocks vanish identically
    iU0 = slice(prob.off_U, prob.off_U + prob.s)
    iU1 = slice(zd + prob.off_U, zd + prob.off_U + prob.s)
    assert not Hd[iU0, iU0].any() and not Hd[iU1, iU1].any() and not Hd[iU0, iU1].any()


def test_hessian_non_antisymmetric_generator(oracle):
    # formulas must not silently assume G^T = -G (non-Hermitian effective Hamiltonians)
    prob, Z = random_problem(oracle, N=2, m=2, T=2, order=6, seed=8, hermitian=False)
    zd = prob.zdim
    zz = Z[:2 * zd]
    mu = np.random.default_rng(9).standard_normal(prob.ddim)
    Hd = oracle.interval_hessian_dense(prob, zz[:zd], zz[zd:], mu)
    Hcs = complex_step_jac(lambda x: oracle.interval_jacobian_dense(prob, x[:zd], x[zd:]).T @ mu, zz)
    np.testing.assert_allclose(Hd, Hcs, rtol=1e-11, atol=1e-12)


def test_expm_and_frechet_vs_scipy(oracle):
    rng = np.random.default_rng(10)
    for scale in (0.05, 1.0, 7.0):
        X = rng.standard_normal((6, 6)) * scale
        E = rng.standard_normal((6, 6))
        np.testing.assert_allclose(oracle.expm_taylor(X), sla.expm(X), rtol=1e-12, atol=1e-12 * np.exp(np.linalg.norm(X, 2)))
        eX, L = oracle.expm_frechet_block(X, E)
        eXs, Ls = sla.expm_frechet(X, E)
        np.testing.assert_allclose(L, Ls, rtol=1e-11, atol=1e-11 * np.exp(np.linalg.norm(X, 2)))


@pytest.mark.parametrize("free_time", [True, False])
def test_exponential_integrator(oracle, free_time):
    prob, Z = random_problem(oracle, N=2, m=2, T=2, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=11)
    zd = prob.zdim
    zz = Z[:2 * zd]
    U0, U1, a, h = oracle._split(prob, zz[:zd], zz[zd:])
    G = prob.G_drift + np.tensordot(a, prob.G_drives, axes=(0, 0))
    r = oracle.interval_residual(prob, zz[:zd], zz[zd:])
    np.testing.assert_allclose(r[:prob.s], (U1 - sla.expm(h * G) @ U0).reshape(-1, order="F"), atol=1e-13)
    J = oracle.interval_jacobian_dense(prob, zz[:zd], zz[zd:])
    Jcs = complex_step_jac(lambda x: oracle.interval_residual(prob, x[:zd], x[zd:]), zz)
    np.testing.assert_allclose(J, Jcs, rtol=1e-11, atol=1e-12)
    # orthogonality of the step for antisymmetric G
    E = oracle.expm_taylor(h * G)
    np.testing.assert_allclose(E @ E.T, np.eye(prob.n), atol=1e-13)


@pytest.mark.parametrize("free_time", [True, False])
@pytest.mark.parametrize("layout", ["standard", "shuffled", "script"])
@pytest.mark.parametrize("hermitian", [True, False])
def test_exponential_hessian_vs_complex_step(oracle, free_time, layout, hermitian):
    """mu_d2F of the exponential integrator (the reference solves :exponential problems with eval_hessian left on,
    unitary_smooth_pulse_problem.jl:224-266): the second Frechet derivative through the 3n x 3n block matrix against a
    complex step of the Jacobian, whose own Frechet derivatives come from the 2n x 2n block matrix."""
    prob, Z = random_problem(oracle, N=2, m=3, T=2, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=21, layout=layout,
                             hermitian=hermitian)
    zd = prob.zdim
    zz = Z[:2 * zd]
    mu = np.random.default_rng(22).standard_normal(prob.ddim)
    Hd = oracle.interval_hessian_dense(prob, zz[:zd], zz[zd:], mu)
    Hcs = complex_step_jac(lambda x: oracle.interval_jacobian_dense(prob, x[:zd], x[zd:]).T @ mu, zz)
    np.testing.assert_allclose(Hd, Hcs, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(Hd, Hd.T, atol=0)
    # delta is linear in U_{t+1}: every block that touches knot t+1 vanishes; so does (U_t, U_t)
    assert not Hd[zd:, :].any() and not Hd[:, zd:].any()
    iU0 = slice(prob.off_U, prob.off_U + prob.s)
    assert not Hd[iU0, iU0].any()
    # the structure covers every non-zero of the dense block
    loc = np.array(oracle.hess_structure_local(prob))
    mask = np.zeros_like(Hd, dtype=bool)
    mask[loc[:, 0], loc[:, 1]] = True
    assert not np.triu(Hd)[~mask].any()
    s, m = prob.s, prob.m
    assert len(loc) == s * m + m * (m + 1) // 2 + ((s + m + 1 + sum(d.dim for d in prob.derivs)) if free_time else 0)


def test_second_frechet_derivative_vs_finite_differences_of_scipy(oracle):
    """L2_exp(X; A, B) against a central second difference of scipy's expm_frechet (an independent implementation)."""
    rng = np.random.default_rng(23)
    for scale in (0.1, 1.0, 4.0):
        X = rng.standard_normal((5, 5)) * scale
        A, B = rng.standard_normal((5, 5)), rng.standard_normal((5, 5))
        L2 = oracle.expm_frechet2_block(X, A, B)
        np.testing.assert_allclose(L2, oracle.expm_frechet2_block(X, B, A), rtol=1e-12, atol=1e-13 * np.abs(L2).max())
        eps = 1e-5
        fd = (sla.expm_frechet(X + eps * B, A)[1] - sla.expm_frechet(X - eps * B, A)[1]) / (2 * eps)
        np.testing.assert_allclose(L2, fd, rtol=2e-7, atol=2e-8 * np.abs(L2).max())


def test_exponential_hessian_mpmath_spot_check(oracle):
    """50-digit evaluation of the (a_i, a_j), (a_j, h) and (h, h) entries from the Taylor series of exp
    (sum over ordered insertions of the directions) certifies the float64 oracle to 1e-12."""
    import mpmath as mp

    mp.mp.dps = 50
    prob, Z = random_problem(oracle, N=2, m=2, T=2, integrator=oracle.EXPONENTIAL, seed=24)
    zd = prob.zdim
    z0, z1 = Z[:zd], Z[zd:]
    n, N, s, m = prob.n, prob.N, prob.s, prob.m
    mu = np.random.default_rng(25).standard_normal(prob.ddim)
    Hd = oracle.interval_hessian_dense(prob, z0, z1, mu)
    tom = lambda A: mp.matrix(A.tolist())
    Gk = [tom(prob.G_drives[j]) for j in range(m)]
    G = tom(prob.G_drift)
    for j in range(m):
        G = G + mp.mpf(float(z0[prob.off_a + j])) * Gk[j]
    h = mp.mpf(float(z0[prob.off_dt]))
    U0 = tom(z0[prob.off_U:prob.off_U + s].reshape(n, N, order="F"))
    Mm = tom(mu[:s].reshape(n, N, order="F"))
    inner = lambda A, B: sum(A[i, j] * B[i, j] for i in range(A.rows) for j in range(A.cols))
    K = 40
    X = h * G
    Xp = [mp.eye(n)]
    for k in range(K + 1):
        Xp.append(Xp[-1] * X)
    # exp, first and second directional derivatives of the series sum_k X^k / k!
    def d1(A):
        out = mp.zeros(n)
        for k in range(1, K):
            out += sum((Xp[i] * A * Xp[k - 1 - i] for i in range(k)), mp.zeros(n)) / mp.factorial(k)
        return out
    def d2(A, B):
        out = mp.zeros(n)
        for k in range(2, K):
            acc = mp.zeros(n)
            for al in range(k - 1):
                for be in range(k - 1 - al):
                    ga = k - 2 - al - be
                    acc += Xp[al] * A * Xp[be] * B * Xp[ga] + Xp[al] * B * Xp[be] * A * Xp[ga]
            out += acc / mp.factorial(k)
        return out
    E = mp.expm(X)
    for i in range(m):
        for j in range(i, m):
            ref = -inner(Mm, d2(h * Gk[i], h * Gk[j]) * U0)
            assert abs(float(ref) - Hd[prob.off_a + i, prob.off_a + j]) <= 1e-12 * max(1.0, abs(float(ref)))
    for j in range(m):
        ref = -inner(Mm, (Gk[j] * E + G * d1(h * Gk[j])) * U0)
        assert abs(float(ref) - Hd[prob.off_a + j, prob.off_dt]) <= 1e-12 * max(1.0, abs(float(ref)))
    ref = -inner(Mm, G * G * E * U0)
    assert abs(float(ref) - Hd[prob.off_dt, prob.off_dt]) <= 1e-12 * max(1.0, abs(float(ref)))


def test_nnz_formulas_match_survey_table(oracle):
    # (N, m) -> (zdim, ddim, jac nnz, hess nnz)
    table = {(2, 2): (15, 12, 104, 58), (4, 4): (45, 40, 704, 343), (8, 6): (147, 140, 5040, 1832), (16, 8): (537, 528, 37440, 9277)}
    for (N, m), (zdim, ddim, jn, hn) in table.items():
        n, s = 2 * N, 2 * N * N
        prob = oracle.Problem(N=N, m=m, T=3, zdim=zdim, off_U=0, off_a=s, off_dt=s + 3 * m,
                              G_drift=np.zeros((n, n)), G_drives=np.zeros((m, n, n)),
                              derivs=[oracle.DerivSpec(s, s + m, m), oracle.DerivSpec(s + m, s + 2 * m, m)])
        assert prob.zdim == s + 3 * m + 1 and prob.ddim == ddim
        assert oracle.jac_nnz_interval(prob) == jn == 2 * N * n * n + s * m + s + 8 * m
        assert len(oracle.hess_structure_local(prob)) == hn == m * (m + 1) // 2 + 2 * s * m + m + 2 * s + 1 + 2 * m
        assert oracle.hess_nnz_interval(prob) == hn                      # default layout: exactly SURVEY's structural entries
        prob.hess_align = 16
        assert oracle.hess_nnz_interval(prob) == -(-hn // 16) * 16       # line-aligned layout: padded to whole 128-byte lines


@pytest.mark.parametrize("integrator", ["pade", "exp"])
def test_coo_assembly_equals_dense(oracle, integrator):
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=2, m=2, T=4, seed=12, integrator=integ)
    vals = oracle.dF(prob, Z)
    rows, cols = oracle.jac_structure(prob)
    assert len(set(zip(rows.tolist(), cols.tolist()))) == rows.size       # no duplicate entries
    Jd = oracle.dense_from_coo(vals, rows, cols, (prob.n_rows, prob.n_vars))
    zd, dd = prob.zdim, prob.ddim
    ref = np.zeros_like(Jd)
    for t in range(prob.T - 1):
        ref[t * dd:(t + 1) * dd, t * zd:(t + 2) * zd] = oracle.interval_jacobian_dense(prob, Z[t * zd:(t + 1) * zd], Z[(t + 1) * zd:(t + 2) * zd])
    np.testing.assert_array_equal(Jd, ref)   # structure covers every structural non-zero
    if True:   # both integrators have an analytic Hessian (round 6)
        mu = np.random.default_rng(13).standard_normal(prob.n_rows)
        hv = oracle.mu_d2F(prob, Z, mu)
        hr, hc = oracle.hess_structure(prob)
        assert (hr <= hc).all()
        # duplicate-free except for the alignment padding (explicit zeros repeating an interval's first entry)
        assert len(set(zip(hr.tolist(), hc.tolist()))) == hr.size - oracle.hess_pad(prob) * (prob.T - 1)
        Hd =oracle.dense_from_coo(hv, hr, hc, (prob.n_vars, prob.n_vars), symmetric=True)
        refH = np.zeros_like(Hd)
        for t in range(prob.T - 1):
            refH[t * zd:(t + 2) * zd, t * zd:(t + 2) * zd] += oracle.interval_hessian_dense(
                prob, Z[t * zd:(t + 1) * zd], Z[(t + 1) * zd:(t + 2) * zd], mu[t * dd:(t + 1) * dd])
        np.testing.assert_allclose(Hd, refH, atol=1e-15)


def test_mpmath_spot_check(oracle):
    """50-digit evaluation of the order-4 residual and its a/dt derivatives on one knot certifies that
    float64 evaluation in the oracle is good to ~1e-14 relative, far inside the 1e-10 target."""
    import mpmath as mp

    mp.mp.dps = 50
    prob, Z = random_problem(oracle, N=2, m=2, T=2, order=4, seed=14)
    zd = prob.zdim
    z0, z1 = Z[:zd], Z[zd:]
    n, N = prob.n, prob.N
    tom = lambda A: mp.matrix(A.tolist())
    G = tom(prob.G_drift)
    for j in range(prob.m):
        G = G + mp.mpf(float(z0[prob.off_a + j])) * tom(prob.G_drives[j])
    h = mp.mpf(float(z0[prob.off_dt]))
    U0 = tom(z0[prob.off_U:prob.off_U + prob.s].reshape(n, N, order="F"))
    U1 = tom(z1[prob.off_U:prob.off_U + prob.s].reshape(n, N, order="F"))
    I = mp.eye(n)
    G2 = G * G
    B = I - h / 2 * G + h ** 2 / 12 * G2
    Fm = I + h / 2 * G + h ** 2 / 12 * G2
    delta = B * U1 - Fm * U0
    ref = np.array([[float(delta[i, j]) for j in range(N)] for i in range(n)]).reshape(-1, order="F")
    got = oracle.interval_residual(prob, z0, z1)[:prob.s]
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-15)
    J = oracle.interval_jacobian_dense(prob, z0, z1)
    S, D = U1 + U0, U1 - U0
    dh = -mp.mpf(1) / 2 * G * S + h / 6 * G2 * D
    ref = np.array([[float(dh[i, j]) for j in range(N)] for i in range(n)]).reshape(-1, order="F")
    np.testing.assert_allclose(J[:prob.s, prob.off_dt], ref, rtol=1e-13, atol=1e-15)
    for j in range(prob.m):
        Gj = tom(prob.G_drives[j])
        da = -h / 2 * Gj * S + h ** 2 / 12 * (Gj * G + G * Gj) * D
        ref = np.array([[float(da[i, k]) for k in range(N)] for i in range(n)]).reshape(-1, order="F")
        np.testing.assert_allclose(J[:prob.s, prob.off_a + j], ref, rtol=1e-13, atol=1e-15)


def test_reference_fixture_layout(oracle):
    """The reference's 15x5 fixture (test/test_utils.jl:54-70): its state rows are the Hadamard geodesic
    in the iso-vec layout to print precision, zdim = 15, dt = 0.2."""
    with open(os.path.join(GOLD, "named_trajectory_type_1.json")) as f:
        fx = json.load(f)
    data = np.array(fx["data"])
    assert data.shape == (15, 5)
    Hgate = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
    Hgen = 1j * sla.logm(Hgate.astype(complex))
    for k, sfrac in enumerate([0, 0.25, 0.5, 0.75, 1.0]):
        U = sla.expm(-1j * Hgen * sfrac)
        np.testing.assert_allclose(data[0:8, k], oracle.operator_to_iso_vec(U), atol=2e-6)
    np.testing.assert_allclose(fx["initial_U"], oracle.operator_to_iso_vec(np.eye(2)))
    np.testing.assert_allclose(fx["goal_U"], oracle.operator_to_iso_vec(np.array([[0, 1], [1, 0]])))  # file says X
    np.testing.assert_array_equal(data[14], 0.2)


@pytest.mark.parametrize("K", [1, 3])
@pytest.mark.parametrize("integrator", ["pade", "exp"])
def test_ket_problem_derivatives(oracle, K, integrator):
    """K kets = an n x K iso state: same formulas, K columns (reference quantum_state_smooth_pulse_problem.jl:146-152)."""
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=3, m=2, T=2, order=6, seed=40 + K, integrator=integ, ncol=K)
    zd = prob.zdim
    zz = Z[:2 * zd]
    assert prob.s == 6 * K and prob.ddim == 6 * K + 4
    J = oracle.interval_jacobian_dense(prob, zz[:zd], zz[zd:])
    Jcs = complex_step_jac(lambda x: oracle.interval_residual(prob, x[:zd], x[zd:]), zz)
    np.testing.assert_allclose(J, Jcs, rtol=1e-11, atol=1e-12)
    assert oracle.jac_nnz_interval(prob) == (2 * K * 36 if integ == oracle.PADE else K * 36 + 6 * K) + 6 * K * 2 + 6 * K + 16
    if True:   # both integrators
        mu = np.random.default_rng(1).standard_normal(prob.ddim)
        Hd = oracle.interval_hessian_dense(prob, zz[:zd], zz[zd:], mu)
        Hcs = complex_step_jac(lambda x: oracle.interval_jacobian_dense(prob, x[:zd], x[zd:]).T @ mu, zz)
        np.testing.assert_allclose(Hd, Hcs, rtol=1e-11, atol=1e-12)
