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
    np.testing.assert_allclose(Hd[first:first + 32, first:first + 32], np.triu(100.0 * Hl), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(con.g(Z), [Fr - 0.99], atol=1e-13)
    np.testing.assert_allclose(con.dg(Z), gF, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(con.mu_d2g(Z, [2.5]), 2.5 * HF[np.triu_indices(32)][np.lexsort((np.triu_indices(32)[0], np.triu_indices(32)[1]))], rtol=1e-10, atol=1e-11)
    obj.close()
    con.close()


def test_ket_and_density_fidelity_known_answers(oracle):
    rng = np.random.default_rng(3)
    for N in (2, 3, 8):
        g = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        g /= np.linalg.norm(g)
        p = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        p /= np.linalg.norm(p)
        giso, piso = np.concatenate([g.real, g.imag]), np.concatenate([p.real, p.imag])
        F, grad, H = oracle.ket_fidelity_value_grad_hess(piso, giso)
        assert abs(F - abs(np.vdot(g, p)) ** 2) < 1e-14
        assert abs(oracle.ket_fidelity_value_grad_hess(giso, giso)[0] - 1.0) < 1e-14
        eps = 1e-6
        gfd = np.array([(oracle.ket_fidelity_value_grad_hess(piso + eps * e, giso)[0] - oracle.ket_fidelity_value_grad_hess(piso - eps * e, giso)[0]) / (2 * eps)
                        for e in np.eye(2 * N)])
        np.testing.assert_allclose(grad, gfd, rtol=1e-7, atol=1e-9)
        Hfd = np.array([(oracle.ket_fidelity_value_grad_hess(piso + eps * e, giso)[1] - oracle.ket_fidelity_value_grad_hess(piso - eps * e, giso)[1]) / (2 * eps)
                        for e in np.eye(2 * N)])
        np.testing.assert_allclose(H, Hfd, rtol=1e-6, atol=1e-8)
        # density: psi' rho psi for a mixed state, and 1 for rho = |g><g|
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        rho = A @ A.conj().T
        rho /= np.trace(rho)
        riso = np.concatenate([rho.real.reshape(-1, order="F"), rho.imag.reshape(-1, order="F")])
        Fd, gd = oracle.density_fidelity_value_grad(riso, giso)
        assert abs(Fd - np.real(np.vdot(g, rho @ g))) < 1e-14
        P = np.outer(g, g.conj())
        assert abs(oracle.density_fidelity_value_grad(np.concatenate([P.real.reshape(-1, order="F"), P.imag.reshape(-1, order="F")]), giso)[0] - 1) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 3, 8, 16])
def test_ket_and_density_fidelity_kernels(qc, oracle, N):
    from qcolloc_amd.objectives import _Fidelity
    rng = np.random.default_rng(20 + N)
    g = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    g /= np.linalg.norm(g)
    giso = np.concatenate([g.real, g.imag])
    p = rng.standard_normal(2 * N)
    f = _Fidelity(giso, kind="ket")
    F, L, grad, H = f.eval(p)
    Fr, gr, Hr = oracle.ket_fidelity_value_grad_hess(p, giso)
    assert abs(F - Fr) < 1e-13 * max(1, abs(Fr)) and abs(L - abs(1 - Fr)) < 1e-13 * max(1, abs(Fr))
    np.testing.assert_allclose(grad, gr, rtol=1e-13, atol=1e-14)
    r, c = np.triu_indices(2 * N)
    order = np.lexsort((r, c))
    np.testing.assert_allclose(H, Hr[r[order], c[order]], rtol=1e-13, atol=1e-14)
    assert abs(qc.iso_fidelity(p, giso) - Fr) < 1e-13 * max(1, abs(Fr))
    f.close()
    rho = rng.standard_normal(2 * N * N)
    fd = _Fidelity(giso, kind="density")
    F, L, grad, H = fd.eval(rho)
    Fr, gr = oracle.density_fidelity_value_grad(rho, giso)
    assert abs(F - Fr) < 1e-13 * max(1, abs(Fr))
    np.testing.assert_allclose(grad, gr, rtol=1e-14, atol=1e-15)
    assert not H.any()
    fd.close()


@pytest.mark.gpu
def test_ket_and_density_objectives_through_the_host_mirror(qc, oracle):
    sys_ = qc.multi_qubit_system(2)
    psi0, psi1 = np.eye(4)[:, 0].astype(complex), (np.eye(4)[:, 1] + 1j * np.eye(4)[:, 2]) / np.sqrt(2)
    inp = qc.quantum_state_smooth_pulse_inputs(sys_, [psi0], [psi1], 7)
    traj = inp.traj
    name = [n for n in traj.names if n.startswith("ψ̃")][0]
    traj.goal[name] = np.concatenate([psi1.real, psi1.imag])
    obj = qc.QuantumStateObjective(name, traj, Q=50.0)
    con = qc.FinalQuantumStateFidelityConstraint(name, 0.9, traj)
    Z = traj.datavec
    u = Z[obj.state_indices]
    Fr, gr, Hr = oracle.ket_fidelity_value_grad_hess(u, traj.goal[name])
    assert abs(obj.L(Z) - 50.0 * abs(1 - Fr)) < 1e-12
    np.testing.assert_allclose(obj.grad_L(Z), -np.sign(1 - Fr) * 50.0 * gr, rtol=1e-12, atol=1e-13)
    assert abs(con.g(Z)[0] - (Fr - 0.9)) < 1e-13
    np.testing.assert_allclose(con.dg(Z), gr, rtol=1e-12, atol=1e-13)
    import test_density
    osys = test_density.open_system(qc, 1)
    psi = np.array([0.6, 0.8j])
    dinp = qc.density_operator_smooth_pulse_inputs(osys, np.eye(2) / 2, psi, 6)
    dobj = qc.DensityOperatorPureStateInfidelityObjective("ρ⃗̃", psi, dinp.traj, Q=10.0)
    Zd = dinp.traj.datavec
    Fd, gd = oracle.density_fidelity_value_grad(Zd[dobj.state_indices], np.concatenate([psi.real, psi.imag]))
    assert abs(dobj.L(Zd) - 10.0 * abs(1 - Fd)) < 1e-12
    np.testing.assert_allclose(dobj.grad_L(Zd), -np.sign(1 - Fd) * 10.0 * gd, rtol=1e-12, atol=1e-13)
    for o in (obj, con, dobj):
        o.close()


# ------------------------------------------------------------------------------------------------
#  Free-phase fidelity (unitary_minimum_time_problem.jl:86-100) and the |tr|^2 / n^2 form
# ------------------------------------------------------------------------------------------------
ZP = np.array([[1, 0], [0, -1]], dtype=complex)
XP = np.array([[0, 1], [1, 0]], dtype=complex)
FREE_PHASE_CASES = [(2, None, [ZP]), (4, None, [ZP, ZP]), (4, None, [ZP, 0.3 * XP + 0.5 * ZP]), (9, [0, 1, 3, 4], [ZP, ZP]),
                    (16, None, [ZP, ZP, ZP, ZP]), (8, None, [np.diag([0.0, 1.0, 2.0, 3.0]).astype(complex), ZP])]


def test_hermitian_eig_helper(qc):
    """qc_hermitian_eig (host-side set-up of the phase operators): A = V diag(w) V', V unitary."""
    L = qc._lib
    rng = np.random.default_rng(0)
    for d in (1, 2, 3, 4, 7, 16):
        A = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        A = (A + A.conj().T) / 2
        for M in (A, np.diag(rng.standard_normal(d)).astype(complex)):
            Ar, Ai = np.asfortranarray(M.real), np.asfortranarray(M.imag)
            w, Vr, Vi = np.empty(d), np.empty((d, d), order="F"), np.empty((d, d), order="F")
            assert L.lib.qc_hermitian_eig(d, L.dptr(Ar), L.dptr(Ai), L.dptr(w), L.dptr(Vr), L.dptr(Vi)) == 0
            V = Vr + 1j * Vi
            assert np.abs(V @ np.diag(w) @ V.conj().T - M).max() < 1e-13 and np.abs(V.conj().T @ V - np.eye(d)).max() < 1e-13
            np.testing.assert_allclose(np.sort(w), np.linalg.eigvalsh(M), atol=1e-13)


@pytest.mark.parametrize("case", range(4))
@pytest.mark.parametrize("form", ["abs", "abs2"])
def test_free_phase_fidelity_oracle(oracle, case, form):
    """Oracle: value against the operator-level statement, derivatives against central differences, known answers:
    a Y gate is an X gate followed by a virtual Z rotation (the reference's own test, unitary_smooth_pulse_problem.jl:342-374)."""
    N, sub, ops = FREE_PHASE_CASES[case]
    rng = np.random.default_rng(case)
    Q = rand_unitary(N, rng)
    U = Q @ (np.eye(N) + 0.1 * (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))))
    goal, u = oracle.operator_to_iso_vec(Q), oracle.operator_to_iso_vec(U)
    phi = rng.standard_normal(len(ops))
    x = np.concatenate([u, phi])
    F, g, H = oracle.free_phase_fidelity_value_grad_hess(x, goal, ops, sub, form)
    assert abs(F - oracle.unitary_free_phase_fidelity(U, Q, phi, ops, sub, form)) < 1e-14
    eps = 1e-6
    f = lambda y: oracle.free_phase_fidelity_value_grad_hess(y, goal, ops, sub, form)
    gfd = np.array([(f(x + eps * e)[0] - f(x - eps * e)[0]) / (2 * eps) for e in np.eye(x.size)])
    Hfd = np.array([(f(x + eps * e)[1] - f(x - eps * e)[1]) / (2 * eps) for e in np.eye(x.size)])
    np.testing.assert_allclose(g, gfd, rtol=1e-7, atol=2e-9)
    np.testing.assert_allclose(H, Hfd, rtol=1e-6, atol=2e-8)
    np.testing.assert_allclose(H, H.T, atol=1e-15)
    if case == 0:
        Y = np.array([[0, -1j], [1j, 0]])
        fy = lambda ph: oracle.unitary_free_phase_fidelity(XP, Y, [ph], [ZP])
        assert fy(0.0) < 1e-15 and abs(fy(np.pi / 2) - 1.0) < 1e-14
        # no phase operators: the plain fidelity
        F0, g0, H0 = oracle.free_phase_fidelity_value_grad_hess(u, goal, [], sub, "abs")
        Fp, gp, Hp = oracle.fidelity_value_grad_hess(u, goal, sub)
        assert abs(F0 - Fp) < 1e-15
        np.testing.assert_allclose(g0, gp, atol=1e-15)
        np.testing.assert_allclose(H0, Hp, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(FREE_PHASE_CASES)))
@pytest.mark.parametrize("form", ["abs", "abs2"])
def test_free_phase_fidelity_kernel(qc, oracle, case, form):
    """qc_fidelity_create_desc with free phases (eigenbasis form on the device) against the oracle's dense Kronecker /
    matrix-exponential form: value, gradient and Hessian over [U~ ; phi]."""
    from qcolloc_amd.objectives import _Fidelity
    N, sub, ops = FREE_PHASE_CASES[case]
    rng = np.random.default_rng(30 + case)
    Q = rand_unitary(N, rng)
    goal = oracle.operator_to_iso_vec(Q)
    u = oracle.operator_to_iso_vec(rand_unitary(N, rng)) + 0.05 * rng.standard_normal(2 * N * N)
    x = np.concatenate([u, rng.standard_normal(len(ops))])
    f = _Fidelity(goal, sub, form=form, phase_operators=ops)
    assert f.P == x.size == qc._lib.lib.qc_fidelity_input_len(f._h)
    F, L, grad, H = f.eval(x)
    Fr, gr, Hr = oracle.free_phase_fidelity_value_grad_hess(x, goal, ops, sub, form)
    assert abs(F - Fr) < 1e-12 * max(1, abs(Fr)) and abs(L - abs(1 - Fr)) < 1e-12
    np.testing.assert_allclose(grad, gr, rtol=1e-10, atol=1e-12)
    r, c = np.triu_indices(x.size)
    order = np.lexsort((r, c))
    np.testing.assert_allclose(H, Hr[r[order], c[order]], rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Hr).max()))
    assert abs(qc.iso_vec_unitary_free_phase_fidelity(u, goal, x[u.size:], ops, subspace=sub, form=form) - Fr) < 1e-12
    f.close()
    # the squared form without phases, and the plain form through the descriptor entry point
    fp = _Fidelity(goal, sub, form=form)
    F, L, grad, H = fp.eval(u)
    Fr, gr, Hr = oracle.free_phase_fidelity_value_grad_hess(u, goal, [], sub, form)
    assert abs(F - Fr) < 1e-13
    np.testing.assert_allclose(grad, gr, rtol=1e-11, atol=1e-13)
    r, c = np.triu_indices(u.size)
    order = np.lexsort((r, c))
    np.testing.assert_allclose(H, Hr[r[order], c[order]], rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Hr).max()))
    fp.close()


@pytest.mark.gpu
def test_free_phase_descriptor_errors(qc, oracle):
    L = qc._lib
    import ctypes as C
    goal = oracle.operator_to_iso_vec(np.eye(4, dtype=complex))
    d = L.qc_fidelity_desc()
    d.kind, d.N, d.goal_iso, d.n_phases = L.QC_FID_UNITARY, 4, L.dptr(goal), 1
    dims = np.array([2], dtype=np.int32)                     # 2 != subspace size 4
    ops = np.concatenate([ZP.real.reshape(-1), ZP.imag.reshape(-1)])
    d.phase_dims, d.phase_ops = dims.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(ops)
    h = C.c_void_p()
    assert L.lib.qc_fidelity_create_desc(C.byref(d), C.byref(h)) == L.QC_ERR_INVALID
    dims[0] = 4
    bad = np.arange(32, dtype=np.float64)                    # not Hermitian
    d.phase_ops = L.dptr(bad)
    assert L.lib.qc_fidelity_create_desc(C.byref(d), C.byref(h)) == L.QC_ERR_INVALID
    d.n_phases, d.form = 0, 7
    assert L.lib.qc_fidelity_create_desc(C.byref(d), C.byref(h)) == L.QC_ERR_INVALID
