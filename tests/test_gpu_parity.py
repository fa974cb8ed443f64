"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Tolerance: BASELINE.json north_star asks for 1e-10 relative on residual / Jacobian
entries and bit-exact sparsity structure; we assert RTOL = 1e-10 per entry with an absolute floor of
1e-12 x (largest entry of the block), and structures with array_equal."""
import json
import os

import numpy as np
import pytest
import torch

from oracle_bridge import assert_same_hessian_values, problem_from_inputs, random_problem, sparse_drive_problem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-10


def assert_close(got, ref, what=""):
    scale = max(1.0, float(np.max(np.abs(ref)))) if ref.size else 1.0
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-12 * scale, err_msg=what)


def dynamics_from_problem(qc, prob, kernel="auto", t_range=None):
    """Build a handle straight from an oracle Problem through the raw C descriptor (no host mirror)."""
    import ctypes as C
    L = qc._lib
    d = L.qc_desc()
    d.N, d.m, d.T, d.zdim, d.global_dim = prob.N, prob.m, prob.T, prob.zdim, prob.global_dim
    d.off_U, d.off_a, d.off_dt, d.dt_fixed = prob.off_U, prob.off_a, prob.off_dt, prob.dt_fixed
    d.integrator, d.pade_order = prob.integrator, prob.order
    d.n_deriv = len(prob.derivs)
    for i, dv in enumerate(prob.derivs):
        d.deriv_x_off[i], d.deriv_dx_off[i], d.deriv_dim[i] = dv.x_off, dv.dx_off, dv.dim
    G0 = np.asfortranarray(prob.G_drift)
    Gd = np.ascontiguousarray(np.stack([g.reshape(-1, order="F") for g in prob.G_drives])) if prob.m else np.zeros((1, 1))
    d.G_drift, d.G_drives = L.dptr(G0), L.dptr(Gd)
    d.state_cols = getattr(prob, "ncol", 0)
    d.device = 0
    d.kernel = {"auto": L.QC_KERNEL_AUTO, "lds": L.QC_KERNEL_LDS, "mfma": L.QC_KERNEL_MFMA}[kernel]
    if t_range:
        d.t_begin, d.t_end = t_range
    h = C.c_void_p()
    L.check(L.lib.qc_create(C.byref(d), C.byref(h)))
    dims = L.qc_dims_t()
    L.check(L.lib.qc_dims(h, C.byref(dims)), h)
    return h, dims, (G0, Gd)


class RawHandle:
    def __init__(self, qc, prob, **kw):
        self.qc, self.L = qc, qc._lib
        self.h, self.dims, self._keep = dynamics_from_problem(qc, prob, **kw)

    def F_jac(self, Z):
        F = np.empty(self.dims.F_len)
        J = np.empty(self.dims.jac_nnz)
        self.L.check(self.L.lib.qc_eval_F_jac(self.h, self.L.dptr(Z), self.L.dptr(F), self.L.dptr(J)), self.h)
        return F, J

    def F(self, Z):
        F = np.empty(self.dims.F_len)
        self.L.check(self.L.lib.qc_eval_F(self.h, self.L.dptr(Z), self.L.dptr(F)), self.h)
        return F

    def hess(self, Z, mu):
        H = np.empty(self.dims.hess_nnz)
        self.L.check(self.L.lib.qc_eval_hess(self.h, self.L.dptr(Z), self.L.dptr(mu), self.L.dptr(H)), self.h)
        return H

    def structure(self):
        jr = np.empty(self.dims.jac_nnz, dtype=np.int64)
        jc = np.empty(self.dims.jac_nnz, dtype=np.int64)
        self.L.check(self.L.lib.qc_jac_structure(self.h, self.L.iptr(jr), self.L.iptr(jc), 0), self.h)
        return jr, jc

    def close(self):
        self.L.lib.qc_destroy(self.h)


def kernels_for(qc, prob):
    ks = ["lds"]
    if prob.integrator == 0 and prob.order == 4 and ((prob.n <= 16 and prob.nc <= 8) or (16 < prob.n <= 32 and prob.nc <= 16)
                                                     or (32 < prob.n <= 64 and prob.nc <= 32)):
        ks.append("mfma")
    if prob.integrator == 0 and prob.order != 4 and prob.n <= 16 and prob.nc <= 8:      # any-order kernel (qc_mfma_padeP.hip)
        ks.append("mfma")
    if prob.integrator == 1 and prob.m <= 8 and ((prob.n <= 16 and prob.nc <= 8) or (16 < prob.n <= 32 and prob.nc <= 16)):
        ks.append("mfma")
    return ks


# ------------------------------------------------------------------------------------------------
#  BASELINE.json configurations (oracle-sized T), through the host mirror
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,T", [(1, 50), (2, 200), (3, 64), (5, 9)])
def test_config_parity(qc, oracle, cfg, T):
    inp = qc.config_inputs(cfg, T=T)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    F_ref, J_ref = oracle.F(prob, Z), oracle.dF(prob, Z)
    rr, rc = oracle.jac_structure(prob)
    for kernel in ["auto", "lds"]:
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        F, J = dyn.F_dF(Z)
        assert_close(F, F_ref, f"F cfg{cfg} {dyn.kernel}")
        assert_close(J, J_ref, f"dF cfg{cfg} {dyn.kernel}")
        assert_close(dyn.F(Z), F_ref)
        assert_close(dyn.dF(Z), J_ref)
        jr, jc = dyn.dF_structure
        np.testing.assert_array_equal(jr, rr)
        np.testing.assert_array_equal(jc, rc)
        assert dyn.structure_tuples("∂F")[0] == (int(rr[0]) + 1, int(rc[0]) + 1)
        assert getattr(dyn, "∂F") == dyn.dF
        dyn.close()


def test_reference_fixture_input_against_golden(qc, oracle):
    """Input = the reference's own 15x5 fixture (test/test_utils.jl:54-70); expected outputs = golden
    file generated by the build's oracle (tests/golden/make_golden.py), NOT reference output."""
    with open(os.path.join(GOLD, "named_trajectory_type_1.json")) as f:
        fx = json.load(f)
    gold = np.load(os.path.join(GOLD, "fixture_outputs.npz"))
    data = np.array(fx["data"])
    comps = {"Ũ⃗": data[0:8], "a": data[8:10], "da": data[10:12], "dda": data[12:14], "Δt": data[14:15]}
    traj = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    sys_ = qc.QuantumSystem(0.1 * qc.PAULIS["Z"], [qc.PAULIS["X"], qc.PAULIS["Y"]])    # test_utils.jl:123
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", sys_, traj, order=4), qc.DerivativeIntegrator("a", "da", traj),
             qc.DerivativeIntegrator("da", "dda", traj)]
    dyn = qc.QuantumDynamics(integ, traj)
    F, J = dyn.F_dF(traj.datavec)
    assert_close(F, gold["F"])
    assert_close(J, gold["dF"])
    jr, jc = dyn.dF_structure
    np.testing.assert_array_equal(jr, gold["dF_rows"])
    np.testing.assert_array_equal(jc, gold["dF_cols"])
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Random problems through the raw C descriptor: dimensions, orders, layouts, edge cases
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,m", [(1, 1), (2, 2), (3, 2), (4, 4), (5, 3), (8, 6), (16, 2)])
@pytest.mark.parametrize("order", [2, 4, 6, 12])
def test_random_problem_parity(qc, oracle, N, m, order):
    prob, Z = random_problem(oracle, N=N, m=m, T=5, order=order, seed=100 + N + order)
    F_ref, J_ref = oracle.F(prob, Z), oracle.dF(prob, Z)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        F, J = h.F_jac(Z)
        assert_close(F, F_ref, f"F {kernel}")
        assert_close(J, J_ref, f"dF {kernel}")
        h.close()


@pytest.mark.parametrize("layout", ["standard", "shuffled", "script"])
@pytest.mark.parametrize("free_time", [True, False])
def test_layouts_and_fixed_time(qc, oracle, layout, free_time):
    prob, Z = random_problem(oracle, N=2, m=3, T=6, order=4, free_time=free_time, layout=layout, seed=7)
    F_ref, J_ref = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    assert_close(F, F_ref)
    assert_close(J, J_ref)
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    h.close()


def test_edge_cases(qc, oracle):
    # T = 2 (a single interval), m = 0 (no drives), no derivative integrators, non-Hermitian generators
    prob, Z = random_problem(oracle, N=2, m=2, T=2, seed=1)
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    assert_close(F, oracle.F(prob, Z))
    assert_close(J, oracle.dF(prob, Z))
    h.close()
    prob, Z = random_problem(oracle, N=3, m=2, T=4, seed=2, hermitian=False)
    prob.derivs = []
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    assert_close(F, oracle.F(prob, Z))
    assert_close(J, oracle.dF(prob, Z))
    h.close()
    # non-finite input is evaluated, not rejected
    prob, Z = random_problem(oracle, N=2, m=2, T=4, seed=3)
    Zn = Z.copy()
    Zn[prob.zdim + prob.off_a] = np.nan
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Zn)
    nn = prob.ddim
    # a_1 = NaN poisons interval 1 and the a-derivative row of interval 0 (a_1 - a_0 - h da_0), nothing else
    assert np.isfinite(F[:prob.s]).all() and np.isnan(F[nn:2 * nn]).any() and np.isfinite(F[2 * nn:]).all()
    h.close()


def test_shards_concatenate_to_the_full_evaluation(qc, oracle):
    prob, Z = random_problem(oracle, N=4, m=3, T=23, seed=4)
    full = RawHandle(qc, prob)
    F, J = full.F_jac(Z)
    full.close()
    Fs, Js = [], []
    for a, b in ((0, 7), (7, 8), (8, 22)):
        h = RawHandle(qc, prob, t_range=(a, b))
        f, j = h.F_jac(Z)
        Fs.append(f)
        Js.append(j)
        h.close()
    np.testing.assert_array_equal(np.concatenate(Fs), F)       # same kernel, same inputs: bit-identical
    np.testing.assert_array_equal(np.concatenate(Js), J)


@pytest.mark.parametrize("order", [6, 12])
def test_sampling_problem_at_other_pade_orders(qc, oracle, order):
    """UnitarySamplingProblem with `pade_order` != 4: the any-order kernels write into the shared per-interval blocks of a
    composed problem (rows_per_interval / offsets); F, dF, mu_d2F against the composed oracle."""
    from oracle_bridge import composed_oracle
    base = qc.multi_qubit_system(2)
    systems = [qc.QuantumSystem(base.H_drift * f, base.H_drives) for f in (0.9, 1.0, 1.15)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["CNOT"], 9, pade_order=order)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert isinstance(dyn, qc.ComposedQuantumDynamics) and len(dyn._parts) == 3
    for part in dyn._parts:
        assert qc._lib.lib.qc_kernel_name(part[2], 0) == b"mfma16-padeP" and qc._lib.lib.qc_kernel_name(part[2], 1) == b"mfma16-padeP-hess"
    ref = composed_oracle(inp)
    Z = inp.traj.datavec
    F, J = dyn.F_dF(Z)
    assert_close(F, ref.F(Z), "composed F")
    assert_close(J, ref.dF(Z), "composed dF")
    mu = np.random.default_rng(order).standard_normal(dyn.dims.n_rows)
    assert_close_h(dyn.mu_d2F(Z, mu), ref.mu_d2F(Z, mu), "composed hessian")
    dyn.close()


@pytest.mark.parametrize("N,m,order", [(8, 5, 6), (8, 3, 12), (20, 3, 4), (32, 2, 4)])
def test_shards_of_the_newer_kernels_concatenate(qc, oracle, N, m, order):
    """Knot shards (t_begin > 0) through the any-order kernels (F+dF and mu_d2F) and the 4 x 4-tile kernels: the shards'
    values concatenate bit-identically to the full evaluation."""
    prob, Z = random_problem(oracle, N=N, m=m, T=9, order=order, seed=N + order)
    mu = np.random.default_rng(5).standard_normal(prob.n_rows)
    full = RawHandle(qc, prob)
    assert full.dims.kernel == qc._lib.QC_KERNEL_MFMA
    F, J = full.F_jac(Z)
    H = full.hess(Z, mu)
    full.close()
    Fs, Js, Hs = [], [], []
    for a, b in ((0, 3), (3, 4), (4, 8)):
        h = RawHandle(qc, prob, t_range=(a, b))
        f, j = h.F_jac(Z)
        Fs.append(f)
        Js.append(j)
        # the multiplier vector of a shard call is the FULL vector (rows of interval t_begin first are read at t_begin * ddim)
        Hs.append(h.hess(Z, mu))
        h.close()
    np.testing.assert_array_equal(np.concatenate(Fs), F)
    np.testing.assert_array_equal(np.concatenate(Js), J)
    np.testing.assert_array_equal(np.concatenate(Hs), H)


def test_host_results_held_by_the_caller_are_not_overwritten(qc):
    """The Python mirror recycles its result arrays (QuantumDynamics._out): three evaluations at different points, all
    results kept, each still equal to a fresh evaluation afterwards."""
    inp = qc.config_inputs(2, T=12)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    rng = np.random.default_rng(0)
    Zs = [inp.traj.datavec + 1e-2 * k * rng.standard_normal(inp.traj.datavec.size) for k in range(3)]
    mu = rng.standard_normal(dyn.dims.n_rows)
    kept = [(dyn.F_dF(Z), dyn.mu_d2F(Z, mu)) for Z in Zs]
    for Z, ((F, J), H) in zip(Zs, kept):
        F2 = dyn.F(Z).copy()
        J2 = dyn.dF(Z).copy()
        H2 = dyn.mu_d2F(Z, mu).copy()
        np.testing.assert_array_equal(F, F2)
        np.testing.assert_array_equal(J, J2)
        np.testing.assert_array_equal(H, H2)
    dyn.close()


def test_device_resident_entry_points(qc, oracle):
    inp = qc.config_inputs(2, T=30)
    Z = inp.traj.datavec
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    F, J = dyn.F_dF(Z)
    dZ = torch.from_numpy(Z).cuda()
    dF = torch.full((dyn.dims.F_len,), float("nan"), dtype=torch.float64, device="cuda")
    dJ = torch.full((dyn.dims.jac_nnz,), float("nan"), dtype=torch.float64, device="cuda")
    dyn.F_dF_device(dZ, dF, dJ)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dF.cpu().numpy(), F)
    np.testing.assert_array_equal(dJ.cpu().numpy(), J)
    # residual only / Jacobian only
    dF.fill_(float("nan"))
    dyn.F_dF_device(dZ, dF, None)
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        dJ2 = torch.empty_like(dJ)
        dyn.F_dF_device(dZ, None, dJ2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dF.cpu().numpy(), F)
    np.testing.assert_array_equal(dJ2.cpu().numpy(), J)
    with pytest.raises(ValueError):
        dyn.F_dF_device(dZ.float(), dF, dJ)
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  BASELINE.json full sizes through size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,T", [(3, 1000), (4, 8000), (5, 500)])
def test_full_size_properties(qc, oracle, cfg, T):
    inp = qc.config_inputs(cfg, T=T)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    F, J = dyn.F_dF(Z)
    assert np.isfinite(F).all() and np.isfinite(J).all()
    nnz, n2, N = dyn.dims.jac_nnz_interval, prob.n ** 2, prob.N
    Jb = J.reshape(T - 1, nnz)
    # (a) the N diagonal copies of -F and of B are identical
    Fc = Jb[:, :N * n2].reshape(T - 1, N, n2)
    Bc = Jb[:, N * n2:2 * N * n2].reshape(T - 1, N, n2)
    assert (Fc == Fc[:, :1]).all() and (Bc == Bc[:, :1]).all()
    # (b) B - F = -h G exactly in exact arithmetic (even powers cancel): B + (-F) = -h (G0 + sum a_j G_j)
    zd = prob.zdim
    Zk = Z[:zd * T].reshape(T, zd)[:-1]
    a, h = Zk[:, prob.off_a:prob.off_a + prob.m], Zk[:, prob.off_dt]
    G = prob.G_drift[None] + np.einsum("tj,jkl->tkl", a, prob.G_drives)
    lhs = (Bc[:, 0] + Fc[:, 0]).reshape(T - 1, prob.n, prob.n).transpose(0, 2, 1)      # col-major blocks
    np.testing.assert_allclose(lhs, -h[:, None, None] * G, rtol=1e-12, atol=1e-13)
    # (c) directional derivative: J v == (F(Z + e v) - F(Z - e v)) / 2e  (COO mat-vec, any size)
    rng = np.random.default_rng(5)
    v = rng.standard_normal(Z.size)
    eps = 1e-6
    fd = (dyn.F(Z + eps * v) - dyn.F(Z - eps * v)) / (2 * eps)
    jr, jc = dyn.dF_structure
    Jv = np.zeros(prob.n_rows)
    np.add.at(Jv, jr, J * v[jc])
    np.testing.assert_allclose(Jv, fd, rtol=1e-6, atol=1e-7)
    # (d) a window of intervals against the oracle, bit-for-bit structure
    t0 = T // 2
    F_ref = oracle.F(prob, Z, t0, t0 + 3)
    J_ref = oracle.dF(prob, Z, t0, t0 + 3)
    assert_close(F[t0 * prob.ddim:(t0 + 3) * prob.ddim], F_ref)
    assert_close(J[t0 * nnz:(t0 + 3) * nnz], J_ref)
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Hessian of the Lagrangian (mu_d2F): values, structure, symmetry bookkeeping
# ------------------------------------------------------------------------------------------------
HTOL = dict(rtol=1e-10)


def assert_close_h(got, ref, what=""):
    scale = max(1.0, float(np.max(np.abs(ref)))) if ref.size else 1.0
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-11 * scale, err_msg=what)


@pytest.mark.parametrize("cfg,T", [(1, 50), (2, 60), (3, 24), (5, 5)])
def test_config_hessian_parity(qc, oracle, cfg, T):
    inp = qc.config_inputs(cfg, T=T)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    rng = np.random.default_rng(3)
    for mu in (np.ones(prob.n_rows), rng.standard_normal(prob.n_rows)):     # reference script uses ones
        H_ref = oracle.mu_d2F(prob, Z, mu)
        for kernel in ("auto", "lds"):
            dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
            assert_close_h(dyn.mu_d2F(Z, mu), H_ref, f"cfg{cfg} {kernel}")
            hr, hc = dyn.mu_d2F_structure
            rr, rc = oracle.hess_structure(prob)
            np.testing.assert_array_equal(hr, rr)
            np.testing.assert_array_equal(hc, rc)
            assert getattr(dyn, "μ∂²F") == dyn.mu_d2F
            dyn.close()


@pytest.mark.parametrize("N,m", [(1, 1), (2, 3), (3, 2), (4, 4), (8, 5)])
@pytest.mark.parametrize("order", [2, 4, 6, 10])
def test_random_hessian_parity(qc, oracle, N, m, order):
    prob, Z = random_problem(oracle, N=N, m=m, T=4, order=order, seed=500 + N + order)
    mu = np.random.default_rng(7).standard_normal(prob.n_rows)
    H_ref = oracle.mu_d2F(prob, Z, mu)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        assert_close_h(h.hess(Z, mu), H_ref, f"{kernel}")
        h.close()


@pytest.mark.parametrize("layout", ["standard", "shuffled"])
@pytest.mark.parametrize("free_time", [True, False])
def test_hessian_layouts_fixed_time_nonhermitian(qc, oracle, layout, free_time):
    prob, Z = random_problem(oracle, N=2, m=3, T=5, order=6, free_time=free_time, layout=layout, seed=9, hermitian=False)
    mu = np.random.default_rng(8).standard_normal(prob.n_rows)
    h = RawHandle(qc, prob)
    assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu))
    h.close()


def test_hessian_fixture_golden_and_exponential_unsupported(qc, oracle):
    with open(os.path.join(GOLD, "named_trajectory_type_1.json")) as f:
        fx = json.load(f)
    gold = np.load(os.path.join(GOLD, "fixture_outputs.npz"))
    data = np.array(fx["data"])
    comps = {"Ũ⃗": data[0:8], "a": data[8:10], "da": data[10:12], "dda": data[12:14], "Δt": data[14:15]}
    traj = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    sys_ = qc.QuantumSystem(0.1 * qc.PAULIS["Z"], [qc.PAULIS["X"], qc.PAULIS["Y"]])
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", sys_, traj, order=4), qc.DerivativeIntegrator("a", "da", traj),
             qc.DerivativeIntegrator("da", "dda", traj)]
    dyn = qc.QuantumDynamics(integ, traj)   # the default layout is the reference's: exactly the structural entries (the golden vectors)
    assert dyn.dims.hess_nnz_interval == 58
    assert_close_h(dyn.mu_d2F(traj.datavec, np.ones(dyn.dims.n_rows)), gold["mu_d2F"])
    hr, hc = dyn.mu_d2F_structure
    np.testing.assert_array_equal(hr, gold["mu_d2F_rows"])
    np.testing.assert_array_equal(hc, gold["mu_d2F_cols"])
    dyn.close()
    # line-aligned layout of device-resident consumers (hess_align = 16): every interval's 58 values padded to 64 with explicit zeros
    # that duplicate the interval's first entry
    dyn = qc.QuantumDynamics(integ, traj, hess_align=16)
    own = gold["mu_d2F"].size // (traj.T - 1)
    Hp = dyn.mu_d2F(traj.datavec, np.ones(dyn.dims.n_rows)).reshape(traj.T - 1, -1)
    assert Hp.shape[1] == 64 and own == 58 and not Hp[:, own:].any()
    assert_close_h(Hp[:, :own].reshape(-1), gold["mu_d2F"])
    hr, hc = (x.reshape(traj.T - 1, -1) for x in dyn.mu_d2F_structure)
    np.testing.assert_array_equal(hr[:, :own].reshape(-1), gold["mu_d2F_rows"])
    np.testing.assert_array_equal(hc[:, own:], np.repeat(hc[:, :1], 64 - own, axis=1))
    dyn.close()


@pytest.mark.parametrize("cfg,T", [(3, 40), (5, 9), (2, 30)])
def test_antisymmetric_generator_path_matches_general_path(qc, monkeypatch, cfg, T):
    """Hermitian Hamiltonians give exactly antisymmetric generators; the Hessian kernels then never load the transposed
    generator images (QcParams.antisym).  Against the general path (QC_NO_ANTISYM=1)."""
    inp = qc.config_inputs(cfg, T=T)
    Z = inp.traj.datavec
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("QC_NO_ANTISYM", flag)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        mu = np.random.default_rng(9).standard_normal(dyn.dims.n_rows)
        out.append(dyn.mu_d2F(Z, mu))
        dyn.close()
    # 2N = 16: a kernel of its own for antisymmetric generators (sign-free formulation, two product stages); 2N = 32: the same
    # kernel, images by negation, but an instantiation of its own (no masks at 16 levels x 16 columns): the compiler contracts
    # multiply-adds differently, the last bit may differ
    scale = np.abs(out[1]).max()
    np.testing.assert_allclose(out[0], out[1], rtol=1e-11, atol=1e-12 * scale)


@pytest.mark.parametrize("cfg,T,align", [(3, 1000, 0), (3, 1000, 16), (5, 500, 0), (5, 500, 16), (3, 3000, 16), (3, 4300, 0), (5, 1100, 16)])
def test_full_size_hessian_properties(qc, oracle, cfg, T, align):
    """Configs 3 (T=1000) and 5 (T=500) at full size, and stretched past the grids' caps (2 999 and 4 299 intervals on 1 024 persistent
    workgroups at 2N = 16; five intervals per workgroup at 2N = 32): linearity in mu, directional second derivative against
    the Jacobian, an oracle window in the middle of the trajectory."""
    inp = qc.config_inputs(cfg, T=T)
    prob = problem_from_inputs(inp)
    prob.hess_align = align or 1
    Z = inp.traj.datavec
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=align)
    rng = np.random.default_rng(11)
    mu1, mu2 = rng.standard_normal(prob.n_rows), rng.standard_normal(prob.n_rows)
    H1, H2, H12 = dyn.mu_d2F(Z, mu1), dyn.mu_d2F(Z, mu2), dyn.mu_d2F(Z, 2.0 * mu1 - 3.0 * mu2)
    np.testing.assert_allclose(H12, 2.0 * H1 - 3.0 * H2, rtol=RTOL, atol=1e-12 * np.abs(H1).max())
    # H v == d/de [J(Z + e v)^T mu] : symmetric COO mat-vec against a central difference of the Jacobian
    v = rng.standard_normal(Z.size)
    eps = 1e-6
    jr, jc = dyn.dF_structure
    def JTmu(Zv):
        out = np.zeros(Z.size)
        np.add.at(out, jc, dyn.dF(Zv) * mu1[jr])
        return out
    fd = (JTmu(Z + eps * v) - JTmu(Z - eps * v)) / (2 * eps)
    hr, hc = dyn.mu_d2F_structure
    Hv = np.zeros(Z.size)
    np.add.at(Hv, hr, H1 * v[hc])
    off = hr != hc
    np.add.at(Hv, hc[off], H1[off] * v[hr[off]])
    np.testing.assert_allclose(Hv, fd, rtol=2e-6, atol=2e-6)
    t0 = T // 2
    nn = dyn.dims.hess_nnz_interval
    own = len(oracle.hess_structure_local(prob))
    assert nn == (own if align == 0 else -(-own // 16) * 16)   # exactly the structural entries by default; hess_align = 16: whole 128-byte lines
    ref = oracle.mu_d2F(prob, Z, mu1, t0, t0 + (2 if cfg == 3 else 1))
    np.testing.assert_allclose(H1[t0 * nn:t0 * nn + ref.size], ref, rtol=1e-10, atol=1e-11)
    dyn.close()


@pytest.mark.parametrize("cfg,align", [(3, 0), (3, 16), (4, 0)])
def test_full_size_every_value_against_the_c_oracle(qc, cfg, align):
    """BASELINE configs 3 (T = 1000) and 4 (T = 8000, on one GPU) at full size: EVERY residual, Jacobian value and Hessian value of the
    host-buffer entry points against the C restatement of the oracle (oracle/qc_oracle.c, OpenMP: milliseconds at these sizes; itself
    checked against the numpy oracle in tests/test_oracle_c.py), rtol 1e-10; the device-resident one-call form gives the same bits."""
    import torch
    import oracle.qc_oracle_c as oc
    inp = qc.config_inputs(cfg)
    assert inp.traj.T == (1000 if cfg == 3 else 8000)
    prob = problem_from_inputs(inp)
    prob.hess_align = align or 1
    co = oc.COracle(prob)
    rng = np.random.default_rng(cfg)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(prob.n_rows)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=align)
    assert dyn.dims.hess_nnz_interval == (1832 if align == 0 else 1840)    # SURVEY 8's table: 1 832 structural entries per interval
    F, J = dyn.F_dF(Z)
    H = dyn.mu_d2F(Z, mu)
    Fr, Jr = co.F_dF(Z)
    Hr = co.mu_d2F(Z, mu)
    assert F.shape == Fr.shape and J.shape == Jr.shape and H.shape == Hr.shape
    assert_close(F, Fr, f"config {cfg} F")
    assert_close(J, Jr, f"config {cfg} dF")
    assert_close_h(H, Hr, f"config {cfg} mu_d2F")
    assert_close(dyn.F(Z), Fr, f"config {cfg} F alone")
    dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
    dF, dJ, dH = (torch.empty(int(n), dtype=torch.float64, device="cuda") for n in (dyn.dims.F_len, dyn.dims.jac_nnz, dyn.dims.hess_nnz))
    dyn.F_dF_mu_d2F_device(dZ, dmu, dF, dJ, dH)
    torch.cuda.synchronize()
    assert np.array_equal(dF.cpu().numpy(), F) and np.array_equal(dJ.cpu().numpy(), J)
    assert_same_hessian_values(dH.cpu().numpy(), H, dyn, "one call against two launches")     # bit for bit but the (a, a) sums (round 6)
    dyn.close()


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("free_time", [True, False])
def test_row_gather_forms_every_value_against_the_c_oracle(qc, m, free_time):
    """Pauli drives on three qubits (one entry per generator row), 1 .. 6 of them, more than one device round (T = 1100): mu_d2F alone and
    the one-call launch take their row-gather forms (qc_mfma16_hess_gathers / qc_mfma16_fused_gathers: the 2-, 4- and 6-drive
    instantiations).  EVERY value against the C oracle, and the one call bit for bit against the two launches."""
    import torch
    import oracle.qc_oracle_c as oc
    full = qc.multi_qubit_system(3)
    inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(full.H_drift, list(full.H_drives)[:m]), qc.GATES["TOFFOLI"], 1100, free_time=free_time)
    prob = problem_from_inputs(inp)
    prob.hess_align = 1
    co = oc.COracle(prob)
    rng = np.random.default_rng(40 + m)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(prob.n_rows)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert dyn.fused_kernel_name == "mfma16-pade4-fused-gather" and dyn.kernel_names[1] == "mfma16-pade4-hess-gather"
    F, J = dyn.F_dF(Z)
    H = dyn.mu_d2F(Z, mu)
    Fr, Jr = co.F_dF(Z)
    Hr = co.mu_d2F(Z, mu)
    assert_close(F, Fr, f"{m} Pauli drives F")
    assert_close(J, Jr, f"{m} Pauli drives dF")
    assert_close_h(H, Hr, f"{m} Pauli drives mu_d2F")
    dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
    dF, dJ, dH = (torch.empty(int(n), dtype=torch.float64, device="cuda") for n in (dyn.dims.F_len, dyn.dims.jac_nnz, dyn.dims.hess_nnz))
    dyn.F_dF_mu_d2F_device(dZ, dmu, dF, dJ, dH)
    torch.cuda.synchronize()
    assert np.array_equal(dF.cpu().numpy(), F) and np.array_equal(dJ.cpu().numpy(), J)
    assert_same_hessian_values(dH.cpu().numpy(), H, dyn, "one call against two launches")     # bit for bit but the (a, a) sums (round 6)
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Exponential integrator: residual, Jacobian, structure; mu_d2F (round 6: the reference solves :exponential
#  problems with the Hessian left on, unitary_smooth_pulse_problem.jl:224-266)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,m", [(1, 1), (2, 2), (3, 3), (4, 4), (8, 6)])
@pytest.mark.parametrize("free_time", [True, False])
def test_exponential_integrator_parity(qc, oracle, N, m, free_time):
    prob, Z = random_problem(oracle, N=N, m=m, T=4, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=700 + N)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        F, J = h.F_jac(Z)
        assert_close(F, oracle.F(prob, Z), f"exp F {kernel}")
        np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=RTOL, atol=1e-11 * max(1.0, np.abs(J).max()), err_msg=kernel)
        if kernel != kernels_for(qc, prob)[-1]:
            h.close()
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    assert h.dims.hess_nnz_interval == oracle.hess_nnz_interval(prob) > 0
    h.close()


def hess_structure_of(h):
    hr = np.empty(h.dims.hess_nnz, dtype=np.int64)
    hc = np.empty(h.dims.hess_nnz, dtype=np.int64)
    h.L.check(h.L.lib.qc_hess_structure(h.h, h.L.iptr(hr), h.L.iptr(hc), 0), h.h)
    return hr, hc


@pytest.mark.parametrize("N,m", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 2), (8, 6), (8, 8), (8, 1)])
@pytest.mark.parametrize("free_time", [True, False])
def test_exponential_integrator_hessian_parity(qc, oracle, N, m, free_time):
    """mu_d2F of the exponential integrator: every kernel that serves the descriptor against the numpy oracle (second Frechet
    derivative through the 3n x 3n block-triangular exponential), structure with array_equal, mu = ones (the reference's script,
    integrator_test_1qubit.jl:50) and random."""
    prob, Z = random_problem(oracle, N=N, m=m, T=4, free_time=free_time, integrator=oracle.EXPONENTIAL, seed=740 + N + m)
    rng = np.random.default_rng(5)
    for mu in (np.ones(prob.n_rows), rng.standard_normal(prob.n_rows)):
        ref = oracle.mu_d2F(prob, Z, mu)
        for kernel in kernels_for(qc, prob):
            h = RawHandle(qc, prob, kernel=kernel)
            assert h.dims.hess_nnz == ref.size
            assert_close_h(h.hess(Z, mu), ref, f"exp hessian {kernel} N={N} m={m}")
            hr, hc = hess_structure_of(h)
            rr, rc = oracle.hess_structure(prob)
            np.testing.assert_array_equal(hr, rr)
            np.testing.assert_array_equal(hc, rc)
            assert (hc < (np.repeat(np.arange(prob.T - 1), h.dims.hess_nnz_interval) + 1) * prob.zdim).all()   # nothing touches knot t+1
            h.close()


@pytest.mark.parametrize("layout", ["shuffled", "script"])
def test_exponential_integrator_hessian_layouts_large_steps_non_hermitian(qc, oracle, layout):
    prob, Z = random_problem(oracle, N=3, m=2, T=4, integrator=oracle.EXPONENTIAL, seed=22, layout=layout)
    Z[prob.off_dt::prob.zdim] = [0.01, 0.6, 2.5, 0.2]          # 0 .. 6 squarings
    mu = np.random.default_rng(6).standard_normal(prob.n_rows)
    ref = oracle.mu_d2F(prob, Z, mu)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        np.testing.assert_allclose(h.hess(Z, mu), ref, rtol=RTOL, atol=1e-10 * np.abs(ref).max(), err_msg=kernel)
        h.close()
    # generators that are not antisymmetric (non-Hermitian effective Hamiltonians): no formula may assume G^T = -G
    prob, Z = random_problem(oracle, N=4, m=3, T=3, integrator=oracle.EXPONENTIAL, seed=23, layout=layout, hermitian=False)
    mu = np.random.default_rng(7).standard_normal(prob.n_rows)
    ref = oracle.mu_d2F(prob, Z, mu)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        np.testing.assert_allclose(h.hess(Z, mu), ref, rtol=RTOL, atol=1e-10 * np.abs(ref).max(), err_msg=kernel)
        h.close()


@pytest.mark.parametrize("nq,T,free_time", [(1, 7, True), (2, 6, True), (3, 5, True), (3, 4, False)])
def test_exponential_integrator_hessian_row_gather_form(qc, oracle, monkeypatch, nq, T, free_time):
    """Pauli-string drives (one entry per generator row: BASELINE's systems) with `integrator=:exponential`: the row-gather forms of
    qc_mfma_exp.hip and qc_mfma_exp_hess.hip (the products G_j R, G_j QV of every Horner step as gathers) against the numpy oracle and
    against the dense-image forms of the same kernels (QC_NO_ELL=1 at create time), steps from 0 to 5 squarings."""
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI"}[nq]
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(nq), qc.GATES[gate], T, integrator="exponential", free_time=free_time)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec.copy()
    if free_time:
        Z[prob.off_dt::prob.zdim] = np.resize([0.2, 0.01, 1.1, 0.45, 2.0], T)
    rng = np.random.default_rng(31 + nq)
    gather = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert gather.kernel_names == ("mfma16-exp-gather", "mfma16-exp-hess-gather")
    monkeypatch.setenv("QC_NO_ELL", "1")
    dense = qc.QuantumDynamics(inp.integrators, inp.traj)
    monkeypatch.delenv("QC_NO_ELL")
    assert dense.kernel_names == ("mfma16-exp", "mfma16-exp-hess")
    # F + dF: the Horner steps of qc_mfma_exp.hip in both forms
    Fo, Jo = oracle.F(prob, Z), oracle.dF(prob, Z)
    for dyn, what in ((gather, "row gathers"), (dense, "dense images")):
        F, J = dyn.F_dF(Z, fresh=True)
        np.testing.assert_allclose(F, Fo, rtol=1e-10, atol=1e-12, err_msg=what)
        np.testing.assert_allclose(J, Jo, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Jo).max()), err_msg=what)
        np.testing.assert_array_equal(dyn.F(Z, fresh=True), F)           # (the residual-only launch: no chains, the same E)
    for mu in (np.ones(prob.n_rows), rng.standard_normal(prob.n_rows)):
        ref = oracle.mu_d2F(prob, Z, mu)
        Hg, Hd = gather.mu_d2F(Z, mu, fresh=True), dense.mu_d2F(Z, mu, fresh=True)
        assert_close_h(Hg, ref, f"exp hessian, row gathers, {nq} qubits")
        assert_close_h(Hd, ref, f"exp hessian, dense images, {nq} qubits")
        np.testing.assert_allclose(Hg, Hd, rtol=1e-12, atol=1e-13 * np.abs(ref).max())
    gather.close()
    dense.close()


@pytest.mark.parametrize("nq", [1, 2, 3])
def test_exponential_integrator_row_gather_long_trajectory(qc, coracle, nq):
    """Launches of 768 intervals and more: the dense-image form of qc_mfma_exp.hip goes to one wave per interval there, the row-gather
    form keeps its two waves -- every value of F + dF at T = 800 against the C oracle."""
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI"}[nq]
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(nq), qc.GATES[gate], 800, integrator="exponential")
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert dyn.kernel_names[0] == "mfma16-exp-gather"
    Fo, Jo = coracle.COracle(prob).F_dF(Z)
    F, J = dyn.F_dF(Z, fresh=True)
    np.testing.assert_allclose(F, Fo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Jo).max()))
    dyn.close()


@pytest.mark.parametrize("T,free_time,gate", [(4, True, "QFT16"), (3, False, "QFT16")])
def test_exponential_integrator_row_gather_form_four_qubits(qc, coracle, monkeypatch, T, free_time, gate):
    """The same at 2N = 32 (qc_mfma32_exp.hip, qc_mfma32_exp_hess.hip: the owners of the shared chains publish row-major copies of R and
    QV, the drive waves gather from them): four qubits with eight Pauli drives, F + dF and mu_d2F against the C oracle and against the
    dense-image forms, steps from 0 to 5 squarings."""
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(4), qc.GATES[gate], T, integrator="exponential", free_time=free_time)
    prob = problem_from_inputs(inp)
    Z = inp.traj.datavec.copy()
    if free_time:
        Z[prob.off_dt::prob.zdim] = np.resize([0.2, 0.01, 1.3, 0.5], T)
    mu = np.random.default_rng(41).standard_normal(prob.n_rows)
    gather = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert gather.kernel_names == ("mfma32-exp-gather", "mfma32-exp-hess-gather")
    monkeypatch.setenv("QC_NO_ELL", "1")
    dense = qc.QuantumDynamics(inp.integrators, inp.traj)
    monkeypatch.delenv("QC_NO_ELL")
    assert dense.kernel_names == ("mfma32-exp", "mfma32-exp-hess")
    C = coracle.COracle(prob)
    Fo, Jo = C.F_dF(Z)
    Ho = C.mu_d2F(Z, mu)
    for dyn, what in ((gather, "row gathers"), (dense, "dense images")):
        F, J = dyn.F_dF(Z, fresh=True)
        np.testing.assert_allclose(F, Fo, rtol=1e-10, atol=1e-12, err_msg=what)
        np.testing.assert_allclose(J, Jo, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Jo).max()), err_msg=what)
        assert_close_h(dyn.mu_d2F(Z, mu, fresh=True), Ho, f"exp hessian, four qubits, {what}")
    gather.close()
    dense.close()


@pytest.mark.parametrize("N,m,free_time", [(12, 5, True), (9, 8, False), (16, 3, True), (6, 4, True), (3, 7, False)])
def test_exponential_integrator_row_gather_forms_any_size(qc, oracle, coracle, N, m, free_time):
    """Random sparse drive Hamiltonians (a matching of the levels with real or imaginary couplings, or a diagonal: one entry per generator
    row) at sizes that pad the tiles -- 2N = 6 .. 32 -- with the exponential integrator: F + dF and mu_d2F of the row-gather kernels
    against the C oracle."""
    prob, Z = sparse_drive_problem(oracle, m=m, T=4, R=1, N=N, free_time=free_time, seed=90 + N, integrator=oracle.EXPONENTIAL)
    mu = np.random.default_rng(N).standard_normal(prob.n_rows)
    C = coracle.COracle(prob)
    Fo, Jo = C.F_dF(Z)
    Ho = C.mu_d2F(Z, mu)
    h = RawHandle(qc, prob, kernel="mfma")
    big = 2 * N > 16
    assert qc._lib.lib.qc_kernel_name(h.h, 0) == (b"mfma32-exp-gather" if big else b"mfma16-exp-gather")
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == (b"mfma32-exp-hess-gather" if big else b"mfma16-exp-hess-gather")
    F, J = h.F_jac(Z)
    np.testing.assert_allclose(F, Fo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(Jo).max()))
    assert_close_h(h.hess(Z, mu), Ho, f"exp hessian, row gathers, N={N} m={m}")
    h.close()


def test_exponential_integrator_hessian_16_levels_and_beyond(qc, oracle, coracle):
    """N = 9 .. 16 (2 x 2 tiles: qc_mfma32_exp_hess.hip; 1 .. 8 drives: every wave role) and N = 20 (the generic kernel alone) against
    the C oracle's forward-mode chains; both kernels where both serve."""
    for N, m, T in ((16, 3, 3), (16, 8, 3), (16, 1, 2), (12, 7, 3), (9, 4, 2), (20, 2, 2)):
        prob, Z = random_problem(oracle, N=N, m=m, T=T, integrator=oracle.EXPONENTIAL, seed=24 + N)
        mu = np.random.default_rng(8).standard_normal(prob.n_rows)
        ref = coracle.COracle(prob).mu_d2F(Z, mu)
        for kernel in kernels_for(qc, prob):
            h = RawHandle(qc, prob, kernel=kernel)
            assert_close_h(h.hess(Z, mu), ref, f"exp hessian N={N} {kernel}")
            h.close()


def test_exponential_integrator_large_step_and_host_mirror(qc, oracle):
    # large ||hG|| exercises the scaling-and-squaring branch (sq up to ~7)
    prob, Z = random_problem(oracle, N=3, m=2, T=3, integrator=oracle.EXPONENTIAL, seed=21, layout="shuffled")
    Z[prob.off_dt::prob.zdim] = [0.01, 2.5, 9.0]
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-10 * np.abs(Fr).max())
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-10 * np.abs(Jr).max())
    h.close()
    # through the host mirror: UnitaryExponentialIntegrator(state, control, system, traj)
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CNOT"], 12, integrator="exponential")
    probm = problem_from_inputs(inp)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Zv = inp.traj.datavec
    F, J = dyn.F_dF(Zv)
    assert_close(F, oracle.F(probm, Zv))
    np.testing.assert_allclose(J, oracle.dF(probm, Zv), rtol=RTOL, atol=1e-11)
    assert_close(dyn.F(Zv), oracle.F(probm, Zv))
    # orthogonality of the step: the -I (x) E block is (minus) an orthogonal matrix for antisymmetric G
    n = probm.n
    E = -J[:n * n].reshape(n, n, order="F")
    np.testing.assert_allclose(E @ E.T, np.eye(n), atol=1e-12)
    # ... and its Hessian of the Lagrangian, as Ipopt asks for it (`dynamics.mu_d2F(Z.datavec, mu)`, integrator_test_1qubit.jl:50-52)
    mu = np.random.default_rng(3).standard_normal(probm.n_rows)
    assert_close_h(dyn.mu_d2F(Zv, mu), oracle.mu_d2F(probm, Zv, mu), "host mirror exp hessian")
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Ket problems: K QuantumStatePadeIntegrators = an n x K iso state
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K,m", [(2, 1, 2), (2, 3, 2), (3, 2, 3), (8, 2, 6), (8, 1, 4), (16, 3, 2)])
@pytest.mark.parametrize("integrator", ["pade4", "pade8", "exp"])
def test_ket_problems(qc, oracle, N, K, m, integrator):
    integ = oracle.EXPONENTIAL if integrator == "exp" else oracle.PADE
    order = 8 if integrator == "pade8" else 4
    prob, Z = random_problem(oracle, N=N, m=m, T=5, order=order, integrator=integ, seed=900 + N + K, ncol=K)
    assert prob.s == 2 * N * K
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    assert_close(F, oracle.F(prob, Z), "ket F")
    np.testing.assert_allclose(J, oracle.dF(prob, Z), rtol=RTOL, atol=1e-11 * max(1.0, np.abs(J).max()))
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    mu = np.random.default_rng(2).standard_normal(prob.n_rows)
    for kernel in kernels_for(qc, prob):
        hk = RawHandle(qc, prob, kernel=kernel)
        assert_close_h(hk.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), f"ket hessian {integrator} {kernel}")
        hk.close()
    h.close()


def test_ket_problem_through_the_host_mirror(qc, oracle):
    sys_ = qc.multi_qubit_system(2)
    e = np.eye(4)
    inp = qc.quantum_state_smooth_pulse_inputs(sys_, [e[:, 0], e[:, 2], e[:, 3]], [e[:, 1], e[:, 3], e[:, 2]], 20)
    prob = problem_from_inputs(inp)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    F, J = dyn.F_dF(Z)
    assert_close(F, oracle.F(prob, Z))
    assert_close(J, oracle.dF(prob, Z))
    mu = np.random.default_rng(4).standard_normal(prob.n_rows)
    assert_close_h(dyn.mu_d2F(Z, mu), oracle.mu_d2F(prob, Z, mu))
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Several unitary integrators over a merged trajectory (UnitarySamplingProblem)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nq,K,T", [(1, 3, 11), (2, 2, 9), (3, 2, 6)])
def test_sampling_problem_composed_dynamics(qc, oracle, nq, K, T):
    from oracle_bridge import composed_oracle
    rng = np.random.default_rng(5)
    base = qc.multi_qubit_system(nq)
    systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.1 * rng.standard_normal()), base.H_drives) for _ in range(K)]
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI"}[nq]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES[gate], T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert isinstance(dyn, qc.ComposedQuantumDynamics) and len(dyn._parts) == K
    ref = composed_oracle(inp)
    Z = inp.traj.datavec
    assert dyn.dim == ref.rows == inp.traj.dims.states
    F, J = dyn.F_dF(Z)
    assert_close(F, ref.F(Z), "composed F")
    assert_close(J, ref.dF(Z), "composed dF")
    jr, jc = dyn.dF_structure
    rr, rc = ref.structure()
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    mu = rng.standard_normal(dyn.dims.n_rows)
    assert_close_h(dyn.mu_d2F(Z, mu), ref.mu_d2F(Z, mu), "composed hessian")
    hr, hc = dyn.mu_d2F_structure
    orr, oc = ref.hess_structure()
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    # the dense Jacobian assembled from COO equals the finite-difference Jacobian action
    v = rng.standard_normal(Z.size)
    eps = 1e-6
    fd = (dyn.F(Z + eps * v) - dyn.F(Z - eps * v)) / (2 * eps)
    Jv = np.zeros(dyn.dims.n_rows)
    np.add.at(Jv, jr, J * v[jc])
    np.testing.assert_allclose(Jv, fd, rtol=1e-6, atol=1e-7)
    dyn.close()


# ------------------------------------------------------------------------------------------------
#  Randomised sweep over descriptors, and the large-T index path
# ------------------------------------------------------------------------------------------------
def test_randomised_descriptor_sweep(qc, oracle):
    """40 random problems: dimension, drives (including none), order, free/fixed time, ket columns, derivative
    integrators of odd sizes at odd places.  F, dF, mu_d2F and both structures against the oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(40):
        N = int(rng.integers(1, 7))
        m = int(rng.integers(0, 5))
        order = int(rng.choice([2, 4, 4, 6, 8]))
        free_time = bool(rng.integers(0, 2))
        ncol = int(rng.choice([0, 0, 1, 2]))
        integ = oracle.EXPONENTIAL if rng.random() < 0.2 else oracle.PADE
        prob, Z = random_problem(oracle, N=N, m=max(m, 1), T=4, order=order, free_time=free_time, integrator=integ,
                                 seed=int(rng.integers(1 << 30)), ncol=ncol, layout=str(rng.choice(["standard", "shuffled"])))
        if m == 0:     # no drives at all: the generator is the drift alone
            prob.m = 0
            prob.G_drives = prob.G_drives[:0]
            prob.derivs = []
        elif rng.random() < 0.3:
            prob.derivs = prob.derivs[:1]
        h = RawHandle(qc, prob)
        F, J = h.F_jac(Z)
        Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
        tag = f"trial {trial}: N={N} m={m} order={order} ft={free_time} ncol={ncol} integ={integ}"
        assert_close(F, Fr, tag)
        np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()), err_msg=tag)
        jr, jc = h.structure()
        rr, rc = oracle.jac_structure(prob)
        np.testing.assert_array_equal(jr, rr, err_msg=tag)
        np.testing.assert_array_equal(jc, rc, err_msg=tag)
        if integ == oracle.PADE:
            mu = rng.standard_normal(prob.n_rows)
            assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), tag)
        h.close()


@pytest.mark.parametrize("N,m", [(8, 0), (8, 1), (8, 7), (8, 8), (8, 9), (16, 0), (16, 1), (16, 9)])
def test_mfma_drive_count_edges(qc, oracle, N, m):
    """MFMA kernels at the edges of their register/LDS drive blocks (0, odd, exactly 8, beyond 8)."""
    prob, Z = random_problem(oracle, N=N, m=max(m, 1), T=3, order=4, seed=300 + N + m)
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        F, J = h.F_jac(Z)
        assert_close(F, Fr, f"{kernel} F")
        assert_close(J, Jr, f"{kernel} dF")
        mu = np.random.default_rng(1).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), f"{kernel} hess")
        h.close()


@pytest.mark.parametrize("T", [2, 7])
def test_mfma32_fixed_time_without_drives(qc, oracle, T):
    """2N = 32 with neither a timestep nor an amplitude in the knots: the kernels' one vector load of (amplitudes | timestep)
    has nothing to fetch and must stay in bounds (first knot of the vector included)."""
    prob, Z = random_problem(oracle, N=16, m=1, T=T, order=4, free_time=False, seed=911 + T)
    prob.m = 0
    prob.G_drives = prob.G_drives[:0]
    prob.derivs = []
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    F, J = h.F_jac(Z)
    assert_close(F, Fr, "mfma32 F, m = 0, fixed dt")
    assert_close(J, Jr, "mfma32 dF, m = 0, fixed dt")
    mu = np.random.default_rng(T).standard_normal(prob.n_rows)
    assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "mfma32 hess, m = 0, fixed dt")
    h.close()


@pytest.mark.parametrize("m,free_time,layout,hermitian", [(8, True, "standard", True), (5, False, "shuffled", False), (2, True, "shuffled", False),
                                                          (8, False, "standard", False)])
def test_mfma32_hessian_variants(qc, oracle, m, free_time, layout, hermitian):
    """4-qubit MFMA Hessian kernel: odd interval counts, fixed time, shuffled knot layout, non-antisymmetric generators."""
    prob, Z = random_problem(oracle, N=16, m=m, T=4, order=4, free_time=free_time, layout=layout, seed=77 + m, hermitian=hermitian)
    for mu in (np.ones(prob.n_rows), np.random.default_rng(m).standard_normal(prob.n_rows)):
        H_ref = oracle.mu_d2F(prob, Z, mu)
        h = RawHandle(qc, prob, kernel="mfma")
        assert_close_h(h.hess(Z, mu), H_ref, "mfma32 hess")
        assert np.array_equal(h.hess(Z, mu), h.hess(Z, mu))      # fixed reduction order: bit-reproducible
        h.close()


@pytest.mark.parametrize("R,m,free_time,layout,T,dense_drift", [(1, 8, True, "standard", 5, True), (1, 3, False, "shuffled", 4, True), (1, 1, True, "shuffled", 2, False),
                                                                 (1, 6, True, "standard", 300, True), (2, 4, True, "standard", 6, True),
                                                                 (2, 8, True, "shuffled", 3, True), (2, 7, False, "standard", 5, False), (1, 4, True, "standard", 7, True)])
def test_sparse_drive_hessian_kernel(qc, oracle, monkeypatch, R, m, free_time, layout, T, dense_drift):
    """Drive generators with at most two entries per row (Pauli strings, ladder pairs) at 2N = 32 take the row-gather kernel
    (qc_mfma32_ell.hip: `mfma32-pade4-hess-ell`): against the oracle, against the dense-image kernel of the same handle shape
    (QC_NO_ELL=1), and bit-reproducible."""
    # (the last case: four diagonal drives touch the same entries of G: the deepest assembly plan)
    prob, Z = sparse_drive_problem(oracle, m=m, T=T, R=R, free_time=free_time, layout=layout, seed=31 * m + R, dense_drift=dense_drift,
                                   kinds=("diag",) if (R, m, T) == (1, 4, 7) else ("real", "imag", "diag"))
    h = RawHandle(qc, prob, kernel="mfma")
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == b"mfma32-pade4-hess-ell"
    monkeypatch.setenv("QC_NO_ELL", "1")
    hd = RawHandle(qc, prob, kernel="mfma")
    monkeypatch.delenv("QC_NO_ELL")
    assert qc._lib.lib.qc_kernel_name(hd.h, 1) == b"mfma32-pade4-hess"
    for mu in (np.ones(prob.n_rows), np.random.default_rng(m).standard_normal(prob.n_rows)):
        H = h.hess(Z, mu)
        if T <= 8:
            assert_close_h(H, oracle.mu_d2F(prob, Z, mu), "mfma32 ell hess vs oracle")
        assert_close_h(H, hd.hess(Z, mu), "mfma32 ell hess vs the dense-image kernel")
        assert np.array_equal(H, h.hess(Z, mu))
    # F + dF by the same kernel family (`mfma32-pade4-ell`): the compact form of the host path, then the full value vector
    assert qc._lib.lib.qc_kernel_name(h.h, 0) == b"mfma32-pade4-ell" and qc._lib.lib.qc_kernel_name(hd.h, 0) == b"mfma32-pade4"
    F, J = h.F_jac(Z)
    Fd, Jd = hd.F_jac(Z)
    if T <= 8:
        assert_close(F, oracle.F(prob, Z), "mfma32 ell F vs oracle")
        assert_close(J, oracle.dF(prob, Z), "mfma32 ell dF vs oracle")
    assert_close(F, Fd, "mfma32 ell F vs the dense-image kernel")
    assert_close(J, Jd, "mfma32 ell dF vs the dense-image kernel")
    assert np.array_equal(F, h.F(Z))                  # the residual-only launch: the same residuals to the bit
    monkeypatch.setenv("QC_HOST_COMPACT", "0")
    hf = RawHandle(qc, prob, kernel="mfma")
    monkeypatch.delenv("QC_HOST_COMPACT")
    F2, J2 = hf.F_jac(Z)
    assert np.array_equal(F2, F) and np.array_equal(J2, J)
    # ... and all three in one launch (`mfma32-pade4-fused-ell`), device-resident: the same bits as the two launches
    assert qc._lib.lib.qc_kernel_name(h.h, 2) == b"mfma32-pade4-fused-ell"
    import ctypes as C
    mu = np.random.default_rng(m).standard_normal(prob.n_rows)
    H = h.hess(Z, mu)
    dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
    new = lambda n: torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")
    dF, dJ, dH = new(h.dims.F_len), new(h.dims.jac_nnz), new(h.dims.hess_nnz)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    qc._lib.check(qc._lib.lib.qc_eval_F_jac_hess_dev(h.h, ptr(dZ), ptr(dmu), ptr(dF), ptr(dJ), ptr(dH), C.c_void_p(torch.cuda.current_stream().cuda_stream)), h.h)
    torch.cuda.synchronize()
    assert np.array_equal(dJ.cpu().numpy(), J) and np.array_equal(dH.cpu().numpy(), H)
    assert_close(dF.cpu().numpy(), F, "one-call residuals")
    h.close()
    hd.close()
    hf.close()


def test_dense_drives_keep_the_dense_image_kernel(qc, oracle):
    """Three entries in one generator row, or five drives on the same entries of G, and the handle stays with qc_mfma32_hess.hip."""
    prob, Z = sparse_drive_problem(oracle, m=5, T=3, R=1, seed=9, kinds=("diag",))
    h = RawHandle(qc, prob, kernel="mfma")
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == b"mfma32-pade4-hess" and qc._lib.lib.qc_kernel_name(h.h, 0) == b"mfma32-pade4"
    mu = np.random.default_rng(2).standard_normal(prob.n_rows)
    assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "dense fallback, five diagonal drives")
    h.close()
    prob, Z = sparse_drive_problem(oracle, m=3, T=3, R=2, seed=5)
    prob.G_drives[1][3, :] = 0.0
    prob.G_drives[1][:, 3] = 0.0
    prob.G_drives[1][3, [5, 9, 11]] = (0.3, -0.2, 0.7)
    prob.G_drives[1][[5, 9, 11], 3] = (-0.3, 0.2, -0.7)          # still antisymmetric, three entries in row 3
    h = RawHandle(qc, prob, kernel="mfma")
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == b"mfma32-pade4-hess"
    mu = np.random.default_rng(2).standard_normal(prob.n_rows)
    assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "dense fallback")
    h.close()


def test_mfma32_hessian_large_T_matches_lds_kernel(qc):
    """Thousands of workgroups (one per interval), XCD remap included, against the LDS kernel."""
    inp = qc.config_inputs(5, T=2300)
    Z = inp.traj.datavec
    out = {}
    for kernel in ("mfma", "lds"):
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        mu = np.random.default_rng(4).standard_normal(dyn.dims.n_rows)
        out[kernel] = dyn.mu_d2F(Z, mu)
        dyn.close()
    assert_close_h(out["mfma"], out["lds"], "T=2300")


def test_large_T_indexing(qc, oracle):
    """T = 60 000 knots (2.4 GB of Jacobian values, device-resident): 64-bit offsets, persistent grid; spot-check intervals."""
    T = 60000
    inp = qc.config_inputs(3, T=8)
    rng = np.random.default_rng(3)
    base = inp.traj.data
    reps = -(-T // base.shape[1])
    data = np.tile(base, (1, reps))[:, :T] + 1e-3 * rng.standard_normal((base.shape[0], T))
    comps = {nm: data[r.start:r.stop] for nm, r in inp.traj.components.items()}
    traj = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", inp.system, traj, order=4), qc.DerivativeIntegrator("a", "da", traj),
             qc.DerivativeIntegrator("da", "dda", traj)]
    from types import SimpleNamespace
    prob = problem_from_inputs(SimpleNamespace(integrators=integ, traj=traj))
    dyn = qc.QuantumDynamics(integ, traj)
    Zh = traj.datavec
    dZ = torch.from_numpy(Zh).cuda()
    dF = torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda")
    dJ = torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda")
    dyn.F_dF_device(dZ, dF, dJ)
    torch.cuda.synchronize()
    nnz, dd = int(dyn.dims.jac_nnz_interval), prob.ddim
    for t0 in (0, 1, 12345, 32767, 32768, T - 3, T - 2):
        Fr, Jr = oracle.F(prob, Zh, t0, t0 + 1), oracle.dF(prob, Zh, t0, t0 + 1)
        assert_close(dF[t0 * dd:(t0 + 1) * dd].cpu().numpy(), Fr, f"F t={t0}")
        assert_close(dJ[t0 * nnz:(t0 + 1) * nnz].cpu().numpy(), Jr, f"dF t={t0}")
    assert torch.isfinite(dJ).all()
    dyn.close()


@pytest.mark.parametrize("case", ["cfg3", "cfg5", "exp", "kets", "odd"])
def test_compact_host_transfer_equals_full_transfer(qc, oracle, case, monkeypatch):
    """The host-buffer entry point ships one copy of the replicated -F / B blocks over PCIe and replicates on the host;
    the result must be bit-identical to the plain full-vector copy (QC_HOST_COMPACT=0), for every thread count."""
    if case == "cfg3":
        inp = qc.config_inputs(3, T=333)
    elif case == "cfg5":
        inp = qc.config_inputs(5, T=40)
    elif case == "exp":
        inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CX"], 70, integrator="exponential")
    elif case == "kets":
        s2 = qc.multi_qubit_system(2)
        inp = qc.quantum_state_smooth_pulse_inputs(s2, [np.eye(4)[:, 0], np.eye(4)[:, 1], np.eye(4)[:, 2]],
                                                   [np.eye(4)[:, 1], np.eye(4)[:, 0], np.eye(4)[:, 3]], 45)
    else:
        inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(np.diag([0.0, 1.0, 2.5]), [np.diag([1.0, 1.0], 1) + np.diag([1.0, 1.0], -1)]),
                                             np.eye(3, dtype=complex), 33, free_time=False)
    Z = inp.traj.datavec
    out = {}
    # QC_HOST_COMPACT: 0 full copy, 1 direct-to-host, 2 packed;  QC_HOST_LANDING: 1 one watched launch (default), 0 chunk launches
    for mode, threads, landing in (("0", "1", "1"), ("1", "1", "1"), ("1", "3", "1"), ("1", "8", "1"), ("1", "8", "0"), ("1", "2", "0"), ("2", "1", "1"),
                                   ("2", "5", "1")):
        monkeypatch.setenv("QC_HOST_COMPACT", mode)
        monkeypatch.setenv("QC_HOST_THREADS", threads)
        monkeypatch.setenv("QC_HOST_LANDING", landing)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        F, J = dyn.F_dF(Z)
        J2 = dyn.dF(Z)
        assert np.array_equal(J, J2)
        F2 = dyn.F(Z)
        assert np.array_equal(F, F2)
        H = None
        if dyn.dims.hess_nnz:
            mu = np.random.default_rng(3).standard_normal(int(dyn.dims.n_rows))
            H = dyn.mu_d2F(Z, mu)
            assert np.array_equal(H, dyn.mu_d2F(Z, mu))          # the pinned blocks are re-armed between calls
        F3, J3 = dyn.F_dF(Z)                                     # ... for the Jacobian as well
        assert np.array_equal(F3, F) and np.array_equal(J3, J)
        out[(mode, threads, landing)] = (F, J, H)
        dyn.close()
    ref = out[("0", "1", "1")]
    for k, (F, J, H) in out.items():
        assert np.array_equal(F, ref[0]) and np.array_equal(J, ref[1]), k
        assert (H is None and ref[2] is None) or np.array_equal(H, ref[2]), k


def test_host_path_with_non_finite_inputs_and_the_sentinel_pattern(qc, monkeypatch):
    """The one-copy host path watches its pinned block for a sentinel word (a signalling NaN).  Non-finite inputs are evaluated,
    not rejected (Ipopt probes wild points) -- including inputs that carry the sentinel's own bit pattern: the call must return,
    and return the bits of the plain full copy (QC_HOST_COMPACT=0)."""
    inp = qc.config_inputs(3, T=130)
    Z = inp.traj.datavec.copy()
    sentinel = np.array([0x7FF4C0DEC0DE5A5A], dtype=np.uint64).view(np.float64)[0]
    zdim = inp.traj.dim
    Z[5 * zdim + 3] = np.nan                       # a state entry
    Z[17 * zdim + inp.traj.offset("a") + 1] = np.inf
    Z[40 * zdim + inp.traj.offset("da")] = sentinel
    Z[41 * zdim + inp.traj.offset("dda") + 2] = -sentinel
    Z[90 * zdim + 7] = sentinel
    mu = np.random.default_rng(0).standard_normal(int(qc.QuantumDynamics(inp.integrators, inp.traj).dims.n_rows))
    mu[11] = sentinel
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("QC_HOST_COMPACT", mode)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        F, J = dyn.F_dF(Z)
        H = dyn.mu_d2F(Z, mu)
        F2, J2 = dyn.F_dF(Z)                       # the pinned blocks were re-armed: the second call sees the same
        assert np.array_equal(F.view(np.uint64), F2.view(np.uint64)) and np.array_equal(J.view(np.uint64), J2.view(np.uint64))
        out[mode] = (F, J, H)
        dyn.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.isnan(a).any()
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("case", ["cfg3", "cfg3_long", "cfg3_fixed_dt", "m1", "m2", "m3", "m4", "m5", "cfg5", "cfg1",
                                  "pauli2_long", "pauli3_long_fixed_dt", "pauli4_long", "pauli5_long", "dense3_long"])
def test_fused_launch_is_bit_identical(qc, case):
    """qc_eval_F_jac_hess_dev: dF and mu_d2F (and F) at one point in one call.  Where the fused kernel serves the handle (2N = 16,
    1 .. 6 drives, Hermitian Hamiltonians: BASELINE configs 3 / 4) it is ONE launch whose values equal the two launches' bit for bit;
    elsewhere the call is the two launches."""
    import torch
    fused_expected = True
    if case == "cfg3":
        inp = qc.config_inputs(3, T=257)
    elif case == "cfg3_long":
        inp = qc.config_inputs(3, T=1100)                  # more than one round of the device (1024 workgroups): the row-gather form
        fused_expected = "mfma16-pade4-fused-gather"
    elif case == "cfg3_fixed_dt":
        inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(3), qc.GATES["TOFFOLI"], 40, free_time=False)
    elif case in ("m1", "m2", "m3", "m4", "m5"):
        m = int(case[1])
        rng = np.random.default_rng(m)
        herm = lambda: (lambda A: (A + A.conj().T) / 2)(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))
        inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(herm(), [herm() for _ in range(m)]), qc.GATES["TOFFOLI"], 31)
    elif case.startswith("pauli"):
        # Pauli drives on three qubits (one entry per generator row), more than one device round: the one-call launch takes its
        # row-gather form (qc_mfma16_fused_gathers) -- the 2-, 4- and 6-drive instantiations, free and fixed time steps
        m = int(case[5])
        full = qc.multi_qubit_system(3)
        sysm = qc.QuantumSystem(full.H_drift, list(full.H_drives)[:m])
        inp = qc.unitary_smooth_pulse_inputs(sysm, qc.GATES["TOFFOLI"], 1030, free_time="fixed_dt" not in case)
        fused_expected = "mfma16-pade4-fused-gather"
    elif case == "dense3_long":          # dense drive generators keep the images at any length
        rng = np.random.default_rng(33)
        herm = lambda: (lambda A: (A + A.conj().T) / 2)(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))
        inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(herm(), [herm() for _ in range(3)]), qc.GATES["TOFFOLI"], 1030)
    elif case == "cfg5":
        inp, fused_expected = qc.config_inputs(5, T=20), "mfma32-pade4-fused-ell"      # Pauli drives: the row-gather kernels (qc_mfma32_ell.hip)
    else:
        inp, fused_expected = qc.config_inputs(1, T=50), False
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    want = {True: "mfma16-pade4-fused", False: "two-launches"}.get(fused_expected, fused_expected)
    assert dyn.fused_kernel_name == want, dyn.fused_kernel_name
    rng = np.random.default_rng(1)
    Z = torch.from_numpy(inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)).cuda()
    mu = torch.from_numpy(rng.standard_normal(int(dyn.dims.n_rows))).cuda()
    new = lambda n: torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")
    F1, J1, H1 = new(dyn.dims.F_len), new(dyn.dims.jac_nnz), new(dyn.dims.hess_nnz)
    F2, J2, H2 = new(dyn.dims.F_len), new(dyn.dims.jac_nnz), new(dyn.dims.hess_nnz)
    dyn.F_dF_device(Z, F1, J1)
    dyn.mu_d2F_device(Z, mu, H1)
    dyn.F_dF_mu_d2F_device(Z, mu, F2, J2, H2)
    torch.cuda.synchronize()
    for a, b, what in ((F1, F2, "F"), (J1, J2, "dF")):
        assert not torch.isnan(b).any(), what
        assert torch.equal(a, b), f"{what}: {(a != b).sum().item()} of {a.numel()} values differ, max {(a - b).abs().max().item():.3e}"
    assert not torch.isnan(H2).any()
    # mu_d2F: bit for bit, except that the stand-alone launch of a handle whose drives have one entry per row takes the (a, a) block
    # from the Gram matrix (qc_mfma_hess_g2.hip, round 6): those entries agree to rounding
    assert_same_hessian_values(H2.cpu().numpy(), H1.cpu().numpy(), dyn, f"{case}: one call against two launches")
    if inp.traj.T <= 64:      # ... and the one-call values against the ORACLE directly (every fused instantiation, not only through the two launches)
        import __graft_entry__ as g
        o = g.load_oracle()
        prob = problem_from_inputs(inp)
        Zh, muh = Z.cpu().numpy(), mu.cpu().numpy()
        assert_close(F2.cpu().numpy(), o.F(prob, Zh), f"{case}: one-call F vs oracle")
        assert_close(J2.cpu().numpy(), o.dF(prob, Zh), f"{case}: one-call dF vs oracle")
        assert_close_h(H2.cpu().numpy(), o.mu_d2F(prob, Zh, muh), f"{case}: one-call mu_d2F vs oracle")
    # without the residuals
    J3, H3 = new(dyn.dims.jac_nnz), new(dyn.dims.hess_nnz)
    dyn.F_dF_mu_d2F_device(Z, mu, None, J3, H3)
    torch.cuda.synchronize()
    assert torch.equal(J3, J1) and torch.equal(H3, H2)
    dyn.close()


def test_device_entry_points_can_be_captured_in_a_hip_graph(qc):
    """The _dev entry points only enqueue kernels on the caller's stream (after the first call has sized the handle's buffers): a HIP
    graph captured around them replays to the same values.  (profiles/graph_probe.py: a replay of one iteration's two launches costs
    22.5 us against 17.9 us for the stream launches -- a graph launch is dearer than the two launch gaps it saves.)"""
    import torch
    inp = qc.config_inputs(3, T=130)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    d = dyn.dims
    rng = np.random.default_rng(5)
    Z = torch.from_numpy(inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)).cuda()
    mu = torch.from_numpy(rng.standard_normal(int(d.n_rows))).cuda()
    new = lambda n: torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")
    F, J, H, F2 = new(d.F_len), new(d.jac_nnz), new(d.hess_nnz), new(d.F_len)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        dyn.F_dF_mu_d2F_device(Z, mu, F, J, H, s)
        dyn.F_dF_device(Z, F2, None, s)
        s.synchronize()
        ref = (F.clone(), J.clone(), H.clone(), F2.clone())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            dyn.F_dF_mu_d2F_device(Z, mu, F, J, H, torch.cuda.current_stream())
            dyn.F_dF_device(Z, F2, None, torch.cuda.current_stream())
    for t in (F, J, H, F2):
        t.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip((F, J, H, F2), ref):
        assert torch.equal(a, b)
    del graph
    dyn.close()


@pytest.mark.parametrize("m,free_time", [(6, True), (5, True), (3, False), (2, True), (1, True)])
def test_two_wave_hessian_kernel_equals_the_one_wave_kernel(qc, oracle, m, free_time):
    """mu_d2F at 2N = 16: launches of up to 1024 intervals take the two-wave kernel (qc_mfma_hess2.hip), longer ones the one-wave
    kernel (qc_mfma_hess.hip).  A trajectory of 1030 knots in one handle (one-wave) and in two shards on device 0 (two-wave): the
    same bits; a window of both against the oracle."""
    from oracle_bridge import problem_from_inputs
    rng = np.random.default_rng(10 + m)
    herm = lambda: (lambda A: (A + A.conj().T) / 2)(rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)))
    system = qc.QuantumSystem(herm(), [herm() for _ in range(m)])
    inp = qc.unitary_smooth_pulse_inputs(system, qc.GATES["TOFFOLI"], 1030, free_time=free_time)
    inp_short = qc.unitary_smooth_pulse_inputs(system, qc.GATES["TOFFOLI"], 40, free_time=free_time)
    one = qc.QuantumDynamics(inp.integrators, inp.traj)
    two = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0, 0])
    short = qc.QuantumDynamics(inp_short.integrators, inp_short.traj)
    assert one.kernel_names[1] == "mfma16-pade4-hess" and short.kernel_names[1] == "mfma16-pade4-hess2", (one.kernel_names, short.kernel_names)
    # the two-wave kernel against the oracle directly, every value (not only through the one-wave kernel)
    Zs = inp_short.traj.datavec + 1e-2 * rng.standard_normal(inp_short.traj.datavec.size)
    mus = rng.standard_normal(int(short.dims.n_rows))
    assert_close_h(short.mu_d2F(Zs, mus), oracle.mu_d2F(problem_from_inputs(inp_short), Zs, mus), "two-wave kernel vs oracle")
    short.close()
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(int(one.dims.n_rows))
    H1, H2 = one.mu_d2F(Z, mu), two.mu_d2F(Z, mu)
    assert np.array_equal(H1.view(np.uint64), H2.view(np.uint64)), f"{(H1 != H2).sum()} of {H1.size} values differ, max {np.abs(H1 - H2).max():.3e}"
    Ho = oracle.mu_d2F(problem_from_inputs(inp), Z, mu, 0, 3)
    assert np.abs(H2[:Ho.size] - Ho).max() <= 1e-10 * max(1.0, np.abs(Ho).max())
    one.close()
    two.close()


@pytest.mark.parametrize("cfg,T", [(3, 257), (5, 33), (1, 50)])
def test_new_x_elision_and_unaligned_buffers(qc, cfg, T):
    """qc_set_new_x(h, 0): the knots on the device are used and Z is not read at all (handing in garbage proves it); caller
    arrays at odd addresses give the same bits."""
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    rng = np.random.default_rng(cfg)
    Z1 = inp.traj.datavec.copy()
    Z2 = Z1 + 1e-2 * rng.standard_normal(Z1.size)
    mu = rng.standard_normal(int(dyn.dims.n_rows))
    ref1 = (dyn.F(Z1, fresh=True), dyn.dF(Z1, fresh=True), dyn.mu_d2F(Z1, mu, fresh=True))     # (held across many calls of the same closures:
    ref2 = (dyn.F(Z2, fresh=True), dyn.dF(Z2, fresh=True), dyn.mu_d2F(Z2, mu, fresh=True))     #  not the ring's vectors)
    assert not np.array_equal(ref1[1], ref2[1])
    # Ipopt's order: residuals at a new x, then Jacobian and Hessian with new_x = false -- Z is not read
    garbage = np.full(Z1.size, np.nan)
    dyn.set_new_x(True)
    assert np.array_equal(dyn.F(Z1), ref1[0])
    dyn.set_new_x(False)
    assert np.array_equal(dyn.dF(garbage), ref1[1])
    assert np.array_equal(dyn.mu_d2F(garbage, mu), ref1[2])
    F, J = dyn.F_dF(garbage)
    assert np.array_equal(F, ref1[0]) and np.array_equal(J, ref1[1])
    dyn.set_new_x(True)
    assert np.array_equal(dyn.dF(Z2), ref2[1])
    dyn.set_new_x(False)
    assert np.array_equal(dyn.mu_d2F(garbage, mu), ref2[2])
    dyn.set_new_x(True)
    # a fresh handle has nothing on the device: new_x = 0 is ignored until a Z has been seen
    d2 = qc.QuantumDynamics(inp.integrators, inp.traj)
    d2.set_new_x(False)
    assert np.array_equal(d2.F(Z1), ref1[0])
    assert np.array_equal(d2.dF(garbage), ref1[1])
    d2.close()
    # caller arrays that are not page-aligned (slices of larger arrays): the copy engine reads / writes them in place
    Fh = np.empty(int(dyn.dims.F_len) + 5)[3:3 + int(dyn.dims.F_len)]
    Hh = np.empty(int(dyn.dims.hess_nnz) + 9)[7:7 + int(dyn.dims.hess_nnz)]
    Zr = np.empty(Z1.size + 3)[1:1 + Z1.size]
    mur = np.empty(mu.size + 3)[2:2 + mu.size]
    Jh = np.empty(int(dyn.dims.jac_nnz) + 1)[1:]
    Zr[:] = Z2
    mur[:] = mu
    for _ in range(2):
        Fh[:] = -1.0
        Hh[:] = -1.0
        Jh[:] = -1.0
        assert dyn.F(Zr, out=Fh) is Fh and np.array_equal(Fh, ref2[0])
        Fh[:] = -1.0
        dyn.F_dF(Zr, out=(Fh, Jh))
        assert np.array_equal(Fh, ref2[0]) and np.array_equal(Jh, ref2[1])
        dyn.mu_d2F(Zr, mur, out=Hh)
        assert np.array_equal(Hh, ref2[2])
    dyn.close()


@pytest.mark.parametrize("m,free_time,layout,hermitian", [(6, True, "standard", True), (8, False, "shuffled", False), (1, True, "shuffled", False),
                                                          (5, False, "standard", True), (0, True, "standard", True)])
def test_mfma_exponential_variants(qc, oracle, m, free_time, layout, hermitian):
    """3-qubit MFMA exponential kernel: drive-count edges, fixed time, shuffled layout, non-antisymmetric generators,
    timesteps from tiny to large (0 .. 6 squarings), residual-only entry point."""
    prob, Z = random_problem(oracle, N=8, m=max(m, 1), T=6, free_time=free_time, integrator=oracle.EXPONENTIAL, layout=layout,
                             seed=900 + m, hermitian=hermitian)
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    if free_time:
        Z[prob.off_dt::prob.zdim] = [1e-3, 0.05, 0.2, 0.9, 3.0, 0.2]
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    F, J = h.F_jac(Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)            # residual-only instantiation, same arithmetic
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    h.close()


def test_mfma_exponential_config3_matches_lds_kernel_and_rollout(qc, oracle):
    """Full-size exponential problem: MFMA against the LDS kernel, and the residual vanishes on a rolled-out trajectory."""
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(3), qc.GATES["TOFFOLI"], 1000, integrator="exponential")
    Z = inp.traj.datavec.copy()
    out = {}
    for kernel in ("mfma", "lds"):
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        out[kernel] = dyn.F_dF(Z)
        if kernel == "mfma":
            R = dyn.rollout(Z, qc.operator_to_iso_vec(np.eye(8, dtype=complex)))
            Zr = Z.reshape(1000, -1).copy()
            off = inp.traj.offset("Ũ⃗")
            Zr[:, off:off + 128] = R.T
            Fz = dyn.F(Zr.ravel()).reshape(999, -1)
            assert np.abs(Fz[:, :128]).max() < 1e-11
        dyn.close()
    np.testing.assert_allclose(out["mfma"][0], out["lds"][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["mfma"][1], out["lds"][1], rtol=RTOL, atol=1e-11)


def test_mfma_exponential_nonfinite_inputs_do_not_hang(qc, oracle):
    prob, Z = random_problem(oracle, N=8, m=3, T=4, integrator=oracle.EXPONENTIAL, seed=5)
    Z[prob.zdim + prob.off_a] = np.nan
    Z[2 * prob.zdim + prob.off_dt] = np.inf
    h = RawHandle(qc, prob, kernel="mfma")
    F, J = h.F_jac(Z)
    ref = oracle.F(prob, np.nan_to_num(Z, nan=0.1, posinf=0.2))
    Fm = F.reshape(3, -1)
    np.testing.assert_allclose(Fm[0][:prob.s], ref.reshape(3, -1)[0][:prob.s], rtol=RTOL, atol=1e-11)      # the state rows of interval 0 are clean
    assert not np.isfinite(Fm[1][:prob.s]).all() and not np.isfinite(Fm[2][:prob.s]).all()
    h.close()


@pytest.mark.parametrize("ncol", [1, 3, 5, 7])
@pytest.mark.parametrize("integrator", ["pade", "exponential"])
def test_mfma16_kernels_with_ket_states(qc, oracle, ncol, integrator):
    """K < 8 kets on 3 qubits through the 2N = 16 MFMA kernels (state tile partly unused): F, dF, mu_d2F against the oracle,
    with the state component placed LAST in the knot (the masked column loads must not run past the trajectory)."""
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=8, m=3 + (ncol % 2), T=5, order=4, integrator=integ, seed=40 + ncol, ncol=ncol, layout="shuffled",
                             hermitian=(ncol != 3))
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    F, J = h.F_jac(Z)
    assert_close(F, Fr, "ket F")
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    if integrator == "pade":
        mu = np.random.default_rng(ncol).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "ket hess")
    h.close()


def test_batched_launch_equals_one_launch_per_handle(qc, oracle):
    """qc_eval_*_dev_multi: the systems of a sampling problem in one launch (gridDim.y = systems) give bit-identical
    vectors to one launch per handle, and a set that cannot share a launch (a handle forced onto the LDS kernel) falls back."""
    import ctypes as C
    import torch
    base = qc.multi_qubit_system(3)
    systems = [qc.QuantumSystem(base.H_drift * f, base.H_drives) for f in (0.9, 1.0, 1.1, 1.2)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["TOFFOLI"], 41)
    L = qc._lib
    for kernel in ("auto", "lds"):
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        assert isinstance(dyn, qc.ComposedQuantumDynamics) and len(dyn._parts) == 4
        Z = torch.from_numpy(inp.traj.datavec).cuda()
        mu = torch.randn(int(dyn.dims.n_rows), dtype=torch.float64, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = []
        for multi in (True, False):
            F = torch.full((int(dyn.dims.F_len),), float("nan"), dtype=torch.float64, device="cuda")
            J = torch.full((int(dyn.dims.jac_nnz),), float("nan"), dtype=torch.float64, device="cuda")
            H = torch.full((int(dyn.dims.hess_nnz),), float("nan"), dtype=torch.float64, device="cuda")
            if multi:
                dyn.F_dF_device(Z, F, J)
                dyn.mu_d2F_device(Z, mu, H)
            else:
                for part in dyn._parts:
                    L.check(L.lib.qc_eval_F_jac_dev(part[2], Z.data_ptr(), F.data_ptr(), J.data_ptr(), s), part[2])
                    L.check(L.lib.qc_eval_hess_dev(part[2], Z.data_ptr(), mu.data_ptr(), H.data_ptr(), s), part[2])
            torch.cuda.synchronize()
            outs.append((F.cpu().numpy(), J.cpu().numpy(), H.cpu().numpy()))
        for k, (a, b) in enumerate(zip(*outs)):
            assert np.isfinite(a).all()
            if k < 2:
                assert np.array_equal(a, b)
            else:       # (the batched launch keeps the stage-A form of the (a, a) sums; one launch per handle takes the Gram form)
                assert_same_hessian_values(a, b, dyn, "batched launch against one launch per handle")
        dyn.close()
    assert L.lib.qc_eval_F_jac_dev_multi(None, 0, None, None, None, None) == L.QC_ERR_INVALID


@pytest.mark.parametrize("N,integrator", [(8, "pade"), (8, "exponential"), (16, "pade")])
@pytest.mark.parametrize("T", [2, 3])
def test_mfma_kernels_on_the_shortest_trajectories(qc, oracle, N, integrator, T):
    """One or two intervals: half-empty workgroups of the pair-of-intervals kernels, single-interval runs of the
    persistent Hessian kernel, shards of one interval."""
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=N, m=4, T=T, order=4, integrator=integ, seed=10 * N + T)
    h = RawHandle(qc, prob, kernel="mfma")
    F, J = h.F_jac(Z)
    assert_close(F, oracle.F(prob, Z), "F")
    Jr = oracle.dF(prob, Z)
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    if integrator == "pade":
        mu = np.random.default_rng(T).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "hess")
    h.close()
    if T == 3:      # a shard holding only the last interval
        hs = RawHandle(qc, prob, kernel="mfma", t_range=(1, 2))
        Fs, Js = hs.F_jac(Z)
        np.testing.assert_array_equal(Fs, F[prob.ddim:])
        np.testing.assert_array_equal(Js, J[J.size // 2:])
        hs.close()


def test_8f_rows_against_the_golden_fixture(qc):
    """The §8f device paths on the reference's data fixture against tests/golden/fixture_outputs_8f.npz (build-oracle
    outputs, generated by tests/golden/make_golden.py): exponential integrator, rollout, fidelity, cost terms.  No
    oracle code runs in this test: the kernels are compared with committed numbers."""
    import json
    fx = json.load(open(os.path.join(GOLD, "named_trajectory_type_1.json")))
    gold = np.load(os.path.join(GOLD, "fixture_outputs_8f.npz"))
    data = np.array(fx["data"])
    comps = {"Ũ⃗": data[0:8], "a": data[8:10], "da": data[10:12], "dda": data[12:14], "Δt": data[14:15]}
    traj = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt", goal={"Ũ⃗": gold["goal"]}, initial={"Ũ⃗": gold["init"]})
    Zp = np.array([[1, 0], [0, -1]], dtype=complex)
    sys_ = qc.QuantumSystem(0.1 * Zp, [qc.PAULIS["X"], qc.PAULIS["Y"]])
    integrators = [qc.UnitaryExponentialIntegrator("Ũ⃗", "a", sys_, traj), qc.DerivativeIntegrator("a", "da", traj),
                   qc.DerivativeIntegrator("da", "dda", traj)]
    dyn = qc.QuantumDynamics(integrators, traj)
    Zv = traj.datavec
    F, J = dyn.F_dF(Zv)
    np.testing.assert_allclose(F, gold["exp_F"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(J, gold["exp_dF"], rtol=RTOL, atol=1e-11)
    roll = dyn.rollout(Zv, gold["init"])
    np.testing.assert_allclose(roll, gold["rollout"], rtol=1e-10, atol=1e-12)
    assert abs(qc.unitary_rollout_fidelity(traj, sys_) - float(gold["fidelity"])) < 1e-12
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=1.0)
    Zr = Zv.copy()
    Zr[obj.state_indices] = roll[:, -1]
    F1 = float(gold["fidelity"])
    assert abs(obj.L(Zr) - abs(1.0 - F1)) < 1e-12
    np.testing.assert_allclose(obj.grad_L(Zr), -np.sign(1.0 - F1) * gold["fidelity_grad"], rtol=RTOL, atol=1e-12)
    spec = (qc.QuadraticRegularizer("a", traj, gold["terms_R"][:2]) + qc.QuadraticRegularizer("da", traj, gold["terms_R"][2:])
            + qc.MinimumTimeObjective(traj, 1.5))
    terms = qc.TrajectoryObjective(spec, traj, dt_scaled=True)      # the golden vectors were made with the dt-scaled weighting
    Jt, gt, Ht = terms.L_grad_hess(Zv)
    assert abs(Jt - float(gold["terms_J"])) < 1e-13
    np.testing.assert_allclose(gt, gold["terms_grad"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(Ht, gold["terms_hess"], rtol=1e-13, atol=1e-15)
    np.testing.assert_array_equal(terms.hess_structure[0], gold["terms_hess_rows"])
    np.testing.assert_array_equal(terms.hess_structure[1], gold["terms_hess_cols"])
    for o in (dyn, obj, terms):
        o.close()


@pytest.mark.parametrize("m,free_time,layout,hermitian", [(8, True, "standard", True), (3, False, "shuffled", False), (1, True, "shuffled", True),
                                                          (0, True, "standard", True)])
def test_mfma32_exponential_variants(qc, oracle, m, free_time, layout, hermitian):
    """4-qubit MFMA exponential kernel (one workgroup per interval, one wave per drive) against the oracle: drive-count
    edges, fixed time, shuffled layout, non-antisymmetric generators, timesteps from tiny to large, residual-only entry."""
    prob, Z = random_problem(oracle, N=16, m=max(m, 1), T=4, free_time=free_time, integrator=oracle.EXPONENTIAL, layout=layout,
                             seed=1300 + m, hermitian=hermitian)
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    if free_time:
        Z[prob.off_dt::prob.zdim] = [1e-3, 0.2, 1.1, 0.2]
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    F, J = h.F_jac(Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    h.close()


def test_mfma32_exponential_config5_matches_lds_kernel(qc):
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(4), qc.GATES["QFT16"], 120, integrator="exponential")
    Z = inp.traj.datavec
    out = {}
    for kernel in ("mfma", "lds"):
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        out[kernel] = dyn.F_dF(Z)
        dyn.close()
    np.testing.assert_allclose(out["mfma"][0], out["lds"][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["mfma"][1], out["lds"][1], rtol=RTOL, atol=1e-11)


@pytest.mark.parametrize("ncol", [1, 5, 9, 15])
def test_mfma32_exponential_with_ket_states(qc, oracle, ncol):
    prob, Z = random_problem(oracle, N=16, m=2 + ncol % 3, T=3, integrator=oracle.EXPONENTIAL, seed=60 + ncol, ncol=ncol, layout="shuffled")
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    F, J = h.F_jac(Z)
    assert_close(F, Fr, "F")
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    h.close()


@pytest.mark.parametrize("ncol", [1, 6, 11, 15])
def test_mfma32_pade_kernels_with_ket_states(qc, oracle, ncol):
    """K < 16 kets on 4 qubits through the 2N = 32 MFMA kernels: F, dF, mu_d2F against the oracle."""
    prob, Z = random_problem(oracle, N=16, m=2 + ncol % 4, T=4, order=4, seed=80 + ncol, ncol=ncol, layout="shuffled", hermitian=(ncol != 6))
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    F, J = h.F_jac(Z)
    assert_close(F, Fr, "ket F")
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    mu = np.random.default_rng(ncol).standard_normal(prob.n_rows)
    assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "ket hess")
    h.close()


@pytest.mark.parametrize("N,m", [(9, 3), (11, 8), (13, 1), (15, 5)])
@pytest.mark.parametrize("integrator", ["pade", "exponential"])
def test_mfma32_kernels_on_padded_systems(qc, oracle, N, m, integrator):
    """Systems with 9 .. 15 levels (e.g. two 3-level transmons: N = 9) on the 2N = 32 MFMA kernels, zero-padded to the
    2 x 2 tiles: F, dF, mu_d2F and the exponential integrator against the oracle."""
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=N, m=m, T=4, order=4, integrator=integ, seed=7 * N + m, layout="shuffled" if N % 2 else "standard",
                             hermitian=(N != 11))
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    h = RawHandle(qc, prob, kernel="mfma")
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    F, J = h.F_jac(Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    jr, jc = h.structure()
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    if integrator == "pade":
        mu = np.random.default_rng(N).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "hess")
    h.close()


@pytest.mark.parametrize("N,m,integrator", [(20, 3, "pade"), (24, 2, "exponential"), (32, 2, "pade"), (32, 2, "exponential"),
                                            (40, 2, "pade"), (40, 1, "exponential"), (48, 1, "pade")])
def test_systems_beyond_the_lds_budget_use_the_global_workspace(qc, oracle, N, m, integrator):
    """More than ~18 levels (5 qubits: N = 32) do not fit 160 KB of LDS per interval: the same kernels run with their
    scratch in a global-memory workspace -- beyond 32 levels too (two 7-level transmons: N = 49; the library refuses no size).
    F, dF, mu_d2F against the oracle."""
    integ = oracle.PADE if integrator == "pade" else oracle.EXPONENTIAL
    prob, Z = random_problem(oracle, N=N, m=m, T=3 if N <= 32 else 2, order=4, integrator=integ, seed=N + m)
    h = RawHandle(qc, prob, kernel="lds")      # (order-4 Pade would otherwise take the 4 x 4-tile MFMA kernel)
    assert h.dims.kernel == qc._lib.QC_KERNEL_LDS
    F, J = h.F_jac(Z)
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    if integrator == "pade" and (N <= 20 or N == 40):
        mu = np.random.default_rng(N).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "hess")
    h.close()


def test_six_qubits_are_not_refused(qc, oracle):
    """64 levels (2N = 128, 8192 state entries per knot, 2.1 M Jacobian values per interval): the generic kernel with its scratch in the
    global workspace; residuals against the oracle (whose dense Jacobian blocks would need 0.5 GB per interval), the Jacobian through its
    action on a direction against finite differences of the residuals."""
    prob, Z = random_problem(oracle, N=64, m=1, T=2, order=4, integrator=oracle.PADE, seed=64)
    h = RawHandle(qc, prob)
    F, J = h.F_jac(Z)
    Fr = oracle.F(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Fr).max()))
    jr, jc = h.structure()
    rng = np.random.default_rng(0)
    v = rng.standard_normal(Z.size)
    eps = 1e-6
    Jv = np.zeros(F.size)
    np.add.at(Jv, jr, J * v[jc])
    fd = (h.F(Z + eps * v) - h.F(Z - eps * v)) / (2 * eps)
    np.testing.assert_allclose(Jv, fd, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(fd).max()))
    h.close()


@pytest.mark.parametrize("N,m,T,free_time,layout,ncol", [
    (32, 2, 3, True, "standard", 0), (32, 10, 2, True, "standard", 0), (32, 3, 4, False, "shuffled", 0), (32, 0, 3, True, "standard", 0),
    (17, 3, 4, True, "standard", 0), (20, 1, 3, True, "script", 0), (27, 4, 3, True, "shuffled", 0), (31, 2, 2, False, "standard", 0),
    (32, 3, 3, True, "standard", 5), (24, 2, 3, True, "standard", 32), (32, 2, 3, True, "standard", 1)])
def test_mfma64_kernel_matches_oracle(qc, oracle, N, m, T, free_time, layout, ncol):
    """Order-4 Pade at 17 .. 32 levels (5 qubits: N = 32): the 4 x 4-tile MFMA kernel (qc_mfma64_kernels.hip), zero-padded and
    masked below 32 levels / 32 state columns; F, dF and the residual-only launch against the oracle, mu_d2F (which falls
    back to the global-workspace kernel) where the oracle is quick enough."""
    prob, Z = random_problem(oracle, N=N, m=m, T=T, order=4, free_time=free_time, layout=layout, ncol=ncol, seed=3 * N + m)
    if m == 0:
        prob.derivs = []          # no drives: no derivative integrators either
    h = RawHandle(qc, prob)
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    assert qc._lib.lib.qc_kernel_name(h.h, 0) == b"mfma64-pade4"
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == (b"mfma64-pade4-hess" if prob.m <= 14 else b"lds-gws-hess")
    F, J = h.F_jac(Z)
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_array_equal(h.F(Z), F)
    hl = RawHandle(qc, prob, kernel="lds")
    Fl, Jl = hl.F_jac(Z)
    np.testing.assert_allclose(J, Jl, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Jl).max()))
    np.testing.assert_allclose(F, Fl, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Fl).max()))
    if int(h.dims.hess_nnz):
        # mu_d2F: the 4 x 4-tile Hessian kernel (scalar blocks = Gram matrix of the scratch slots) against the
        # global-workspace kernel, and against the oracle where the oracle is quick enough
        mu = np.random.default_rng(N).standard_normal(prob.n_rows)
        Hm, Hl = h.hess(Z, mu), hl.hess(Z, mu)
        assert_close_h(Hm, Hl, "hess vs lds")
        np.testing.assert_array_equal(Hm, h.hess(Z, mu))          # bit-reproducible
        if N <= 20:
            assert_close_h(Hm, oracle.mu_d2F(prob, Z, mu), "hess")
    hl.close()
    h.close()


def test_mfma64_kernel_long_trajectory_and_poisoned_outputs(qc, oracle):
    """More intervals than CUs (several rounds of workgroups) and NaN-poisoned output buffers: every value is written,
    and the 4 x 4-tile kernel agrees with the global-workspace kernel at T = 600."""
    import torch
    prob, Z = random_problem(oracle, N=32, m=3, T=600, order=4, seed=77)
    h = RawHandle(qc, prob)
    hl = RawHandle(qc, prob, kernel="lds")
    import ctypes as C
    L = qc._lib
    Zd = torch.from_numpy(Z).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for hh in (h, hl):
        Fd = torch.full((int(hh.dims.F_len),), float("nan"), dtype=torch.float64, device="cuda")
        Jd = torch.full((int(hh.dims.jac_nnz),), float("nan"), dtype=torch.float64, device="cuda")
        L.check(L.lib.qc_eval_F_jac_dev(hh.h, Zd.data_ptr(), Fd.data_ptr(), Jd.data_ptr(), st), hh.h)
        torch.cuda.synchronize()
        assert not torch.isnan(Fd).any() and not torch.isnan(Jd).any()
        outs.append((Fd, Jd))
    for a, b_ in zip(outs[0], outs[1]):
        scale = max(1.0, float(b_.abs().max()))
        assert float((a - b_).abs().max()) <= 1e-11 * scale
    # the Hessian kernel walks the intervals with 256 workgroups (several intervals each, scratch reused)
    mu = torch.randn(int(h.dims.n_rows), dtype=torch.float64, device="cuda")
    hs = []
    for hh in (h, hl):
        Hd = torch.full((int(hh.dims.hess_nnz),), float("nan"), dtype=torch.float64, device="cuda")
        L.check(L.lib.qc_eval_hess_dev(hh.h, Zd.data_ptr(), mu.data_ptr(), Hd.data_ptr(), st), hh.h)
        torch.cuda.synchronize()
        assert not torch.isnan(Hd).any()
        hs.append(Hd)
    scale = max(1.0, float(hs[1].abs().max()))
    assert float((hs[0] - hs[1]).abs().max()) <= 1e-11 * scale
    h.close()
    hl.close()


@pytest.mark.parametrize("T,m", [(100, 3), (129, 1), (66, 0), (258, 2)])
def test_mfma64_interval_split_thresholds(qc, T, m):
    """5 qubits at the trajectory lengths where several workgroups share an interval (parts = 2 for 65 .. 128 intervals,
    4 below, 1 above): F, dF and mu_d2F of the 4 x 4-tile kernels against the global-workspace LDS kernels, outputs
    poisoned with NaN first; m = 0 / 1 leave workgroups without a drive."""
    import ctypes as C
    import __graft_entry__ as g
    o = g.load_oracle()
    prob, Z = random_problem(o, N=32, m=max(m, 1), T=T, seed=64 + T)
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    L = qc._lib
    h, hl = RawHandle(qc, prob, kernel="mfma"), RawHandle(qc, prob, kernel="lds")
    assert L.lib.qc_kernel_name(h.h, 0) == b"mfma64-pade4" and L.lib.qc_kernel_name(hl.h, 0) == b"lds-gws"
    Zd = torch.from_numpy(Z).cuda()
    mu = torch.randn(int(h.dims.n_rows), dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for hh in (h, hl):
        Fd = torch.full((int(hh.dims.F_len),), float("nan"), dtype=torch.float64, device="cuda")
        Jd = torch.full((int(hh.dims.jac_nnz),), float("nan"), dtype=torch.float64, device="cuda")
        L.check(L.lib.qc_eval_F_jac_dev(hh.h, Zd.data_ptr(), Fd.data_ptr(), Jd.data_ptr(), st), hh.h)
        Hd = torch.full((max(1, int(hh.dims.hess_nnz)),), float("nan"), dtype=torch.float64, device="cuda")
        if hh.dims.hess_nnz:
            L.check(L.lib.qc_eval_hess_dev(hh.h, Zd.data_ptr(), mu.data_ptr(), Hd.data_ptr(), st), hh.h)
        else:
            Hd.zero_()
        torch.cuda.synchronize()
        assert not torch.isnan(Fd).any() and not torch.isnan(Jd).any() and not torch.isnan(Hd).any()
        outs.append((Fd, Jd, Hd))
    for a, b_ in zip(outs[0], outs[1]):
        scale = max(1.0, float(b_.abs().max()))
        assert float((a - b_).abs().max()) <= 1e-11 * scale
    h.close()
    hl.close()


@pytest.mark.parametrize("T", [257, 258])
def test_mfma32_single_interval_threshold(qc, oracle, T):
    """Config-5-sized system at 256 / 257 intervals: the one-interval-per-workgroup instantiation (<= 256 intervals) and
    the pair-per-workgroup one agree with the LDS kernel, and with the oracle on the last intervals (the odd tail)."""
    prob, Z = random_problem(oracle, N=16, m=3, T=T, seed=T)
    h, hl = RawHandle(qc, prob, kernel="mfma"), RawHandle(qc, prob, kernel="lds")
    F, J = h.F_jac(Z)
    Fl, Jl = hl.F_jac(Z)
    assert_close(F, Fl, "mfma32 vs lds F")
    assert_close(J, Jl, "mfma32 vs lds dF")
    nnz, dd = int(h.dims.jac_nnz_interval), int(h.dims.ddim)
    assert_close(J[-2 * nnz:], oracle.dF(prob, Z, T - 3, T - 1), "tail dF")
    assert_close(F[-2 * dd:], oracle.F(prob, Z, T - 3, T - 1), "tail F")
    h.close()
    hl.close()


@pytest.mark.parametrize("order", [2, 6, 8, 10, 12, 20])
@pytest.mark.parametrize("N,m,ncol,free_time", [(8, 6, 0, True), (8, 5, 0, False), (8, 9, 3, True), (5, 2, 0, True), (3, 1, 2, True), (8, 0, 0, True)])
def test_any_order_mfma_kernel_matches_oracle(qc, oracle, order, N, m, ncol, free_time):
    """Pade orders other than 4 at up to 8 levels: the register-resident any-order MFMA kernel (qc_mfma_padeP.hip) -- F, dF
    and the residual-only launch against the oracle; mu_d2F of such a handle is served by the LDS kernel."""
    prob, Z = random_problem(oracle, N=N, m=max(m, 1), T=4, order=order, free_time=free_time, ncol=ncol, seed=order * 100 + N + m)
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    h = RawHandle(qc, prob)
    assert h.dims.kernel == qc._lib.QC_KERNEL_MFMA
    assert qc._lib.lib.qc_kernel_name(h.h, 0) == b"mfma16-padeP"
    assert qc._lib.lib.qc_kernel_name(h.h, 1) == (b"mfma16-padeP-hess" if prob.m <= 8 else b"lds-hess")
    F, J = h.F_jac(Z)
    Fr, Jr = oracle.F(prob, Z), oracle.dF(prob, Z)
    np.testing.assert_allclose(F, Fr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Fr).max()))
    np.testing.assert_allclose(J, Jr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Jr).max()))
    np.testing.assert_allclose(h.F(Z), F, rtol=0, atol=1e-13 * max(1.0, np.abs(Fr).max()))
    if int(h.dims.hess_nnz):
        mu = np.random.default_rng(order).standard_normal(prob.n_rows)
        assert_close_h(h.hess(Z, mu), oracle.mu_d2F(prob, Z, mu), "hess")
    h.close()


def test_kernel_names_of_the_baseline_configurations(qc):
    """Which device kernels serve BASELINE.json's configurations (qc_kernel_name): the tuned MFMA kernels, not a generic path."""
    expect = {1: ("mfma16-pade4", "mfma16-pade4-hess"), 2: ("mfma16-pade4", "mfma16-pade4-hess"),
              3: ("mfma16-pade4", "mfma16-pade4-hess-gather"), 5: ("mfma32-pade4-ell", "mfma32-pade4-hess-ell")}     # (hess-gather: Pauli drives are row gathers in the one-wave kernel; hess2 = two waves per interval, <= 1024 intervals, dense drives; ell: Pauli drives are row gathers)
    for cfg, names in expect.items():
        inp = qc.config_inputs(cfg, T=5)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        assert dyn.kernel_names == names, (cfg, dyn.kernel_names)
        dyn.close()
    s3 = qc.multi_qubit_system(3)
    for kw, names in [(dict(integrator="exponential"), ("mfma16-exp-gather", "mfma16-exp-hess-gather")), (dict(pade_order=12), ("mfma16-padeP", "mfma16-padeP-hess"))]:
        inp = qc.unitary_smooth_pulse_inputs(s3, qc.GATES["TOFFOLI"], 5, **kw)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        assert dyn.kernel_names == names, (kw, dyn.kernel_names)
        dyn.close()
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(5), np.eye(32, dtype=complex), 4)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert dyn.kernel_names == ("mfma64-pade4", "mfma64-pade4-hess")
    dyn.close()
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel="lds")
    assert dyn.kernel_names == ("lds-gws", "lds-gws-hess")
    inp4 = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(4), np.eye(16, dtype=complex), 4, integrator="exponential")
    d4 = qc.QuantumDynamics(inp4.integrators, inp4.traj)
    assert d4.kernel_names == ("mfma32-exp-gather", "mfma32-exp-hess-gather") and d4.fused_kernel_name == "two-launches"
    d4.close()
    dyn.close()


def test_stress_script_short_run():
    """tests/stress_gpu.py (random systems up to 32 levels, all integrators and kernels) with 60 trials; the full runs
    (about 5000 trials, worst relative error 6.3e-13) are quoted in DESIGN.md."""
    import subprocess
    import sys as _sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([_sys.executable, os.path.join(here, "stress_gpu.py"), "60", "99"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "60 trials ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("nq,integrator", [(2, "pade"), (3, "pade"), (3, "exponential")])
def test_quantum_state_sampling_problem(qc, oracle, nq, integrator):
    """QuantumStateSamplingProblem: K systems x several kets over a merged trajectory = one handle per system (a run of
    ket integrators each), values interleaved per interval in integrator order; against the composed oracle."""
    base = qc.multi_qubit_system(nq)
    systems = [qc.QuantumSystem(base.H_drift * f, base.H_drives) for f in (0.9, 1.1)]
    N = base.levels
    psi0 = [np.eye(N)[:, 0], np.eye(N)[:, 1], (np.eye(N)[:, 0] + 1j * np.eye(N)[:, 2]) / np.sqrt(2)]
    psi1 = [np.eye(N)[:, 1], np.eye(N)[:, 0], np.eye(N)[:, 3]]
    inp = qc.quantum_state_sampling_inputs(systems, psi0, psi1, 7, integrator=integrator)
    from oracle_bridge import composed_oracle
    groups = qc.split_groups(inp.integrators)
    assert [len(g) for g in groups] == [3, 5]
    ref = composed_oracle(inp)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert isinstance(dyn, qc.ComposedQuantumDynamics)
    Z = inp.traj.datavec
    F, J = dyn.F_dF(Z)
    assert_close(F, ref.F(Z), "state sampling F")
    Jr = ref.dF(Z)
    np.testing.assert_allclose(J, Jr, rtol=RTOL, atol=1e-11 * max(1.0, np.abs(Jr).max()))
    jr, jc = dyn.dF_structure
    rr, rc = ref.structure()
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    if integrator == "pade":
        mu = np.random.default_rng(nq).standard_normal(dyn.dims.n_rows)
        assert_close_h(dyn.mu_d2F(Z, mu), ref.mu_d2F(Z, mu), "state sampling hessian")
    dyn.close()


def test_shard_of_a_layout_with_odd_block_sizes(qc, oracle):
    """Fixed time, five drives, one derivative integrator: jac_nnz and hess_nnz per interval are odd, so a shard starting at an
    odd interval writes at an address that is 8- but not 16-byte aligned."""
    import torch
    prob, Z = random_problem(oracle, N=8, m=5, T=6, order=4, free_time=False, seed=77)
    prob.derivs = prob.derivs[:1]
    assert oracle.jac_nnz_interval(prob) % 2 == 1
    Jr = oracle.dF(prob, Z)
    Fr = oracle.F(prob, Z)
    for kernel in ("mfma", "lds"):
        h = RawHandle(qc, prob, kernel=kernel, t_range=(1, 4))
        dZ = torch.from_numpy(Z).cuda()
        dF = torch.zeros(Fr.size, dtype=torch.float64, device="cuda")
        dJ = torch.zeros(Jr.size, dtype=torch.float64, device="cuda")
        nnz, dd = oracle.jac_nnz_interval(prob), prob.ddim
        s = torch.cuda.current_stream().cuda_stream
        qc._lib.check(qc._lib.lib.qc_eval_F_jac_dev(h.h, dZ.data_ptr(), dF.data_ptr() + 8 * dd, dJ.data_ptr() + 8 * nnz, s), h.h)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dJ.cpu().numpy()[nnz:4 * nnz], Jr[nnz:4 * nnz], rtol=RTOL, atol=1e-11)
        np.testing.assert_allclose(dF.cpu().numpy()[dd:4 * dd], Fr[dd:4 * dd], rtol=1e-10, atol=1e-12)
        assert not dJ.cpu().numpy()[:nnz].any() and not dJ.cpu().numpy()[4 * nnz:].any()
        h.close()


def test_config5_callback_set_on_one_stream(qc, oracle):
    """BASELINE config 5 as the reference's evaluator sees it (UnitaryMinimumTimeProblem, unitary_minimum_time_problem.jl:67-111):
    dynamics F / dF / mu_d2F, FinalUnitaryFreePhaseFidelityConstraint (value, gradient, Hessian), MinimumTimeObjective and
    the regularisers -- every device-side callback of one Ipopt iteration enqueued on ONE stream at the full T = 500, then
    compared with the oracles (C restatement for the full-size dynamics, numpy for a window and for the terms)."""
    import ctypes as C
    import oracle.qc_oracle_c as oc
    L = qc._lib
    inp = qc.config_inputs(5)
    T, N = inp.traj.T, 16
    rng = np.random.default_rng(55)
    phase_ops = [np.diag([1.0, -1.0]).astype(complex)] * 4
    phases = rng.standard_normal(4)
    comps = {name: inp.traj[name] for name in inp.traj.names}
    traj = qc.NamedTrajectory(comps, controls=inp.traj.controls, timestep=inp.traj.timestep, goal=inp.traj.goal,
                              global_data={"ϕ": phases})
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", inp.system, traj, order=4), qc.DerivativeIntegrator("a", "da", traj),
             qc.DerivativeIntegrator("da", "dda", traj)]
    dyn = qc.QuantumDynamics(integ, traj)
    assert dyn.kernel_names == ("mfma32-pade4-ell", "mfma32-pade4-hess-ell") and dyn.dims.n_cols == traj.dim * T + 4
    con = qc.FinalUnitaryFreePhaseFidelityConstraint("Ũ⃗", "ϕ", phase_ops, 0.99, traj)
    terms = qc.TrajectoryObjective(qc.QuadraticRegularizer("a", traj, 1e-2) + qc.QuadraticRegularizer("da", traj, 1e-2)
                                   + qc.QuadraticRegularizer("dda", traj, 1e-2) + qc.MinimumTimeObjective(traj, 1.0), traj)
    Zh = traj.datavec
    mu_h = rng.standard_normal(int(dyn.dims.n_rows))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        Z = torch.from_numpy(Zh).cuda()
        mu = torch.from_numpy(mu_h).cuda()
        dF = torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device="cuda")
        dJ = torch.empty(int(dyn.dims.jac_nnz), dtype=torch.float64, device="cuda")
        dH = torch.empty(int(dyn.dims.hess_nnz), dtype=torch.float64, device="cuda")
        P = con._f.P
        x = torch.cat([Z[torch.from_numpy(con.state_indices).cuda()]])          # final state and the phases, gathered on the device
        fval = torch.empty(2, dtype=torch.float64, device="cuda")
        fgrad = torch.empty(P, dtype=torch.float64, device="cuda")
        fhess = torch.empty(P * (P + 1) // 2, dtype=torch.float64, device="cuda")
        tJ = torch.empty(1, dtype=torch.float64, device="cuda")
        tg = torch.empty(Zh.size, dtype=torch.float64, device="cuda")
        nnz_t = C.c_int64()
        L.check(L.lib.qc_terms_hess_nnz(terms._h, C.byref(nnz_t)))
        tH = torch.empty(nnz_t.value, dtype=torch.float64, device="cuda")
        sp = C.c_void_p(st.cuda_stream)
        dyn.F_dF_device(Z, dF, dJ, st)
        dyn.mu_d2F_device(Z, mu, dH, st)
        L.check(L.lib.qc_fidelity_eval_dev(con._f._h, C.c_void_p(x.data_ptr()), C.c_void_p(fval.data_ptr()), C.c_void_p(fgrad.data_ptr()),
                                           C.c_void_p(fhess.data_ptr()), sp))
        L.check(L.lib.qc_terms_eval_dev(terms._h, C.c_void_p(Z.data_ptr()), C.c_void_p(tJ.data_ptr()), C.c_void_p(tg.data_ptr()),
                                        C.c_void_p(tH.data_ptr()), sp))
    st.synchronize()
    # dynamics against the C restatement at full size, and against the numpy oracle on a window
    prob = problem_from_inputs(type("I", (), {"integrators": integ, "traj": traj})())
    co = oc.COracle(prob)
    Fr, Jr = co.F_dF(Zh)
    assert_close(dF.cpu().numpy(), Fr, "config 5 F")
    assert_close(dJ.cpu().numpy(), Jr, "config 5 dF")
    assert_close_h(dH.cpu().numpy(), co.mu_d2F(Zh, mu_h), "config 5 mu_d2F")
    nnz = int(dyn.dims.jac_nnz_interval)
    assert_close(dJ.cpu().numpy()[250 * nnz:251 * nnz], oracle.dF(prob, Zh, 250, 251), "config 5 dF window")
    # fidelity constraint with free phases
    Fv, gv, Hv = oracle.free_phase_fidelity_value_grad_hess(Zh[con.state_indices], traj.goal["Ũ⃗"], phase_ops)
    assert abs(float(fval[0]) - Fv) < 1e-12
    np.testing.assert_allclose(fgrad.cpu().numpy(), gv, rtol=1e-10, atol=1e-12)
    r, c = np.triu_indices(P)
    order = np.lexsort((r, c))
    np.testing.assert_allclose(fhess.cpu().numpy(), Hv[r[order], c[order]], rtol=1e-10, atol=1e-11)
    assert abs(con.g(Zh)[0] - (Fv - 0.99)) < 1e-12
    # objective terms
    idx = np.concatenate([np.asarray(traj.components[n]) for n in ("a", "da", "dda")])
    tm = oracle.Terms(T=T, zdim=traj.dim, off_dt=traj.offset("Δt"), reg_index=np.sort(idx), reg_R=np.full(idx.size, 1e-2), D=1.0, n_mt=T - 1,
                      global_dim=4)
    assert abs(float(tJ[0]) - oracle.terms_value(tm, Zh)) < 1e-10 * max(1.0, abs(oracle.terms_value(tm, Zh)))
    np.testing.assert_allclose(tg.cpu().numpy(), oracle.terms_grad(tm, Zh), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(tH.cpu().numpy(), oracle.terms_hess(tm, Zh), rtol=1e-12, atol=1e-14)
    for o_ in (dyn, con, terms):
        o_.close()


@pytest.mark.parametrize("N,m", [(8, 6), (4, 3), (16, 2)])
@pytest.mark.parametrize("dims", [(), (None,), (None, None, None), (None, None, None, 5), (None, 70), (3, None, 65, None, 1)])
def test_derivative_integrator_counts_and_sizes(qc, oracle, N, m, dims):
    """0, 1, 3, 4 and 5 derivative integrators, dimensions up to 70: the kernels serve up to two (Hessian) / four (F + dF)
    integrators of at most 64 components from registers requested with the interval's other loads, anything else through the
    generic tail -- both must give the oracle's rows, Jacobian entries and d2/d(dx_i) dh = -mu_i Hessian entries."""
    rng = np.random.default_rng(41 + N + len(dims))
    n, s = 2 * N, 2 * N * N
    dims = [m if d is None else d for d in dims]
    A = lambda: (lambda X: (X + X.conj().T) / 2)(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)))   # noqa: E731
    G0 = oracle.generator(A())
    Gd = np.array([oracle.generator(A()) for _ in range(m)]).reshape(m, n, n)
    # knot: [U (s), a (m), then per derivative integrator a pair (x, dx) chained where the dimension allows, dt]
    off, derivs = s + m, []
    prev_off, prev_dim = s, m                       # the first chain starts at the amplitudes
    for d in dims:
        if d == prev_dim:                           # chained: x = the previous dx (a -> da -> dda ...)
            x_off = prev_off
        else:                                       # an independent pair
            x_off = off
            off += d
        derivs.append(oracle.DerivSpec(x_off, off, d))
        prev_off, prev_dim = off, d
        off += d
    zdim = off + 1
    T = 4
    prob = oracle.Problem(N=N, m=m, T=T, zdim=zdim, off_U=0, off_a=s, off_dt=zdim - 1, G_drift=G0, G_drives=Gd, dt_fixed=0.17,
                          integrator=oracle.PADE, order=4, derivs=derivs, ncol=0)
    Z = rng.standard_normal(zdim * T) * 0.5
    Z[zdim - 1::zdim] = rng.uniform(0.1, 0.3, size=T)
    mu = rng.standard_normal(prob.n_rows)
    F_ref, J_ref, H_ref = oracle.F(prob, Z), oracle.dF(prob, Z), oracle.mu_d2F(prob, Z, mu)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        F, J = h.F_jac(Z)
        assert_close(F, F_ref, f"F {kernel} {dims}")
        assert_close(J, J_ref, f"dF {kernel} {dims}")
        assert_close_h(h.hess(Z, mu), H_ref, f"hess {kernel} {dims}")
        jr, jc = h.structure()
        rr, rc = oracle.jac_structure(prob)
        assert np.array_equal(jr, rr) and np.array_equal(jc, rc)
        h.close()


@pytest.mark.parametrize("case", ["cfg3", "cfg3_long", "cfg2_fixed_dt", "cfg5", "exp3", "order6", "qutrit_lds"])
def test_shuffled_value_blocks_give_the_same_values_blockwise(qc, oracle, case):
    """qc_desc.jac_block_order / hess_block_order: every kernel family writes its blocks at the offsets the descriptor's order implies
    -- F, dF, mu_d2F and the one-call form of a handle with shuffled blocks equal the default handle's entry for entry (matched through
    the structures), bit for bit; the host-buffer Jacobian path leaves its compact form where the replicated blocks do not lead."""
    import torch
    from oracle_bridge import assert_same_hessian_values
    kw = {}
    if case == "cfg3":
        inp = qc.config_inputs(3, T=40)
    elif case == "cfg3_long":
        inp = qc.config_inputs(3, T=1100)
    elif case == "cfg2_fixed_dt":
        inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CNOT"], 30, free_time=False)
    elif case == "cfg5":
        inp = qc.config_inputs(5, T=12)
    elif case == "exp3":
        inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(3), qc.GATES["TOFFOLI"], 20, integrator="exponential")
    elif case == "order6":
        inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CNOT"], 20, pade_order=6)
    else:
        rng0 = np.random.default_rng(3)
        herm = lambda: (lambda A: (A + A.conj().T) / 2)(rng0.standard_normal((3, 3)) + 1j * rng0.standard_normal((3, 3)))
        inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(herm(), [herm(), herm()]), np.eye(3, dtype=complex), 12)
        kw = dict(kernel="lds")
    rng = np.random.default_rng(17)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    ref = qc.QuantumDynamics(inp.integrators, inp.traj, **kw)
    mu = rng.standard_normal(int(ref.dims.n_rows))
    F0, J0 = ref.F_dF(Z, fresh=True)
    H0 = ref.mu_d2F(Z, mu, fresh=True)
    (jr0, jc0), (hr0, hc0) = ref.dF_structure, ref.mu_d2F_structure
    key = lambda r, c: r.astype(np.int64) * (int(ref.dims.n_cols) + 1) + c
    for jo, ho in (([4, 2, 0, 3, 1], [7, 4, 0, 5, 2, 6, 1, 3]), ([1, 0, 3, 2, 4], [1, 0, 3, 2, 5, 4, 6, 7])):
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, jac_block_order=jo, hess_block_order=ho, **kw)
        assert dyn.kernel_names == ref.kernel_names
        F1, J1 = dyn.F_dF(Z, fresh=True)
        H1 = dyn.mu_d2F(Z, mu, fresh=True)
        (jr1, jc1), (hr1, hc1) = dyn.dF_structure, dyn.mu_d2F_structure
        np.testing.assert_array_equal(F1, F0)
        # entries are unique: sort both by (row, col) and compare values
        p0, p1 = np.argsort(key(jr0, jc0), kind="stable"), np.argsort(key(jr1, jc1), kind="stable")
        np.testing.assert_array_equal(key(jr0, jc0)[p0], key(jr1, jc1)[p1])
        np.testing.assert_array_equal(J1[p1], J0[p0])
        q0, q1 = np.argsort(key(hr0, hc0), kind="stable"), np.argsort(key(hr1, hc1), kind="stable")
        np.testing.assert_array_equal(key(hr0, hc0)[q0], key(hr1, hc1)[q1])
        np.testing.assert_array_equal(H1[q1], H0[q0])
        np.testing.assert_array_equal(dyn.dF(Z, fresh=True), J1)
        # device-resident: the two launches and the one call
        dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
        dF, dJ, dH = (torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda") for n in (dyn.dims.F_len, dyn.dims.jac_nnz, dyn.dims.hess_nnz))
        dyn.F_dF_mu_d2F_device(dZ, dmu, dF, dJ, dH)
        torch.cuda.synchronize()
        assert np.array_equal(dF.cpu().numpy(), F1) and np.array_equal(dJ.cpu().numpy(), J1)
        Hd = dH.cpu().numpy()
        assert np.isfinite(Hd).all()
        # (one call against two launches: bit for bit but the (a, a) sums; compared through the default handle's column mask, permuted)
        np.testing.assert_allclose(Hd[q1], H0[q0], rtol=1e-11, atol=1e-12 * max(1.0, np.abs(H0).max()))
        dyn.close()
    ref.close()
