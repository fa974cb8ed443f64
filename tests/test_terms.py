"""Whole-trajectory cost terms: quadratic regularisers, minimum-time term, equal-timestep rows.
CPU: the oracle's analytic gradient / Hessian against complex-step and finite differences of its own value, the
host-only structure entry points against the oracle.  GPU: `qc_terms_*` against the oracle."""
import ctypes as C

import numpy as np
import pytest


def make_terms(oracle, qc, cfg=1, T=9, free_time=True, baseline=False, dt_scaled=True, D=0.0, seed=0, global_dim=0):
    """A trajectory of a smooth-pulse problem, regularisers on a / da / dda, and the oracle's view of the same terms."""
    rng = np.random.default_rng(seed)
    inp = qc.config_inputs(cfg, T=T) if free_time else qc.unitary_smooth_pulse_inputs(
        qc.multi_qubit_system(1), qc.GATES["H"], T, free_time=False)
    traj = inp.traj
    m = len(traj.components["a"])
    Rs = {"a": 1e-2, "da": rng.uniform(0.5, 2.0, m), "dda": 0.3}
    bl = {"a": rng.standard_normal((m, T)) if baseline else None, "da": None, "dda": None}
    spec = None
    for name in ("a", "da", "dda"):
        term = qc.QuadraticRegularizer(name, traj, Rs[name], baseline=bl[name])
        spec = term if spec is None else spec + term
    if D:
        spec = spec + qc.MinimumTimeObjective(traj, D)
    idx = np.concatenate([traj.components[n] for n in ("a", "da", "dda")])
    R = np.concatenate([np.full(m, Rs["a"]), Rs["da"], np.full(m, Rs["dda"])])
    order = np.argsort(idx)
    base = None
    if baseline:
        base = np.zeros((T, 3 * m))
        base[:, :m] = bl["a"].T
        base = base[:, order]
    free = isinstance(traj.timestep, str)
    tm = oracle.Terms(T=T, zdim=traj.dim, off_dt=traj.offset(traj.timestep) if free else -1, reg_index=idx[order], reg_R=R[order],
                      baseline=base, dt_scaled=dt_scaled, dt_fixed=0.0 if free else float(traj.timestep), D=D,
                      n_mt=T - 1 if D else 0, global_dim=global_dim)
    Z = traj.datavec + 0.05 * rng.standard_normal(traj.datavec.size)
    return traj, spec, tm, Z


@pytest.mark.parametrize("free_time,baseline,dt_scaled,D", [(True, False, True, 0.0), (True, True, True, 2.5), (True, False, False, 1.0),
                                                            (False, True, True, 0.0), (False, False, False, 0.0)])
def test_oracle_terms_derivatives(qc, oracle, free_time, baseline, dt_scaled, D):
    D = D if free_time else 0.0
    traj, spec, tm, Z = make_terms(oracle, qc, T=6, free_time=free_time, baseline=baseline, dt_scaled=dt_scaled, D=D)
    g = oracle.terms_grad(tm, Z)
    eps = 1e-30
    gcs = np.array([np.imag(oracle.terms_value(tm, Z.astype(complex) + 1j * eps * e)) / eps for e in np.eye(Z.size)])
    np.testing.assert_allclose(g, gcs, rtol=1e-13, atol=1e-15)
    r, c = oracle.terms_hess_structure(tm)
    assert np.all(r <= c) and len(set(zip(r.tolist(), c.tolist()))) == r.size
    H = oracle.dense_from_coo(oracle.terms_hess(tm, Z), r, c, (Z.size, Z.size), symmetric=True)
    h = 1e-6
    Hfd = np.array([(oracle.terms_grad(tm, Z + h * e) - oracle.terms_grad(tm, Z - h * e)) / (2 * h) for e in np.eye(Z.size)])
    np.testing.assert_allclose(H, Hfd, rtol=1e-7, atol=1e-8)


def test_terms_known_answers(qc, oracle):
    traj, spec, tm, Z = make_terms(oracle, qc, T=5, D=3.0)
    # all regularised entries zero: only the minimum-time term remains, = D * sum of the first T-1 timesteps
    Z0 = Z.copy().reshape(tm.T, tm.zdim)
    Z0[:, tm.reg_index] = 0.0
    assert abs(oracle.terms_value(tm, Z0.ravel()) - 3.0 * Z0[:-1, tm.off_dt].sum()) < 1e-14
    # doubling every timestep quadruples the regulariser part
    tm0 = oracle.Terms(**{**tm.__dict__, "D": 0.0})
    Z2 = Z.copy().reshape(tm.T, tm.zdim)
    Z2[:, tm.off_dt] *= 2
    assert abs(oracle.terms_value(tm0, Z2.ravel()) / oracle.terms_value(tm0, Z) - 4.0) < 1e-13


def test_terms_structure_host_entry_points(qc, oracle):
    """No device needed: nnz and structure from the descriptor, bit-exact against the oracle."""
    traj, spec, tm, Z = make_terms(oracle, qc, cfg=2, T=7, D=1.0)
    idx = np.ascontiguousarray(tm.reg_index, dtype=np.int32)
    R = np.ascontiguousarray(tm.reg_R)
    d = qc._lib.qc_terms_desc()
    d.T, d.zdim, d.off_dt, d.global_dim, d.n_reg = tm.T, tm.zdim, tm.off_dt, 0, idx.size
    d.weighting = qc._lib.QC_REG_DT_SCALED
    d.reg_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    d.reg_R = qc._lib.dptr(R)
    d.min_time_D, d.min_time_knots = 1.0, tm.T - 1
    nnz = C.c_int64()
    assert qc._lib.lib.qc_terms_desc_hess_nnz(C.byref(d), C.byref(nnz)) == 0
    r0, c0 = oracle.terms_hess_structure(tm)
    assert nnz.value == r0.size == tm.T * (2 * idx.size + 1)
    for one_based in (0, 1):
        r = np.empty(nnz.value, dtype=np.int64)
        c = np.empty(nnz.value, dtype=np.int64)
        assert qc._lib.lib.qc_terms_desc_hess_structure(C.byref(d), qc._lib.iptr(r), qc._lib.iptr(c), one_based) == 0
        np.testing.assert_array_equal(r, r0 + one_based)
        np.testing.assert_array_equal(c, c0 + one_based)
    # plain weighting: diagonal only
    d.weighting = qc._lib.QC_REG_PLAIN
    qc._lib.lib.qc_terms_desc_hess_nnz(C.byref(d), C.byref(nnz))
    assert nnz.value == tm.T * idx.size
    # invalid descriptors
    for mut in (lambda: setattr(d, "weighting", 5), lambda: setattr(d, "off_dt", tm.zdim), lambda: setattr(d, "n_reg", tm.zdim + 1),
                lambda: setattr(d, "min_time_knots", tm.T + 1), lambda: setattr(d, "T", 0)):
        d.weighting, d.off_dt, d.n_reg, d.min_time_knots, d.T = qc._lib.QC_REG_DT_SCALED, tm.off_dt, idx.size, tm.T - 1, tm.T
        mut()
        assert qc._lib.lib.qc_terms_desc_hess_nnz(C.byref(d), C.byref(nnz)) == qc._lib.QC_ERR_INVALID
        assert qc._lib.lib.qc_terms_last_error(None)
    d.weighting, d.off_dt, d.n_reg, d.min_time_knots, d.T = qc._lib.QC_REG_DT_SCALED, tm.off_dt, idx.size, tm.T - 1, tm.T
    bad = idx.copy()
    bad[1] = bad[0]
    d.reg_index = bad.ctypes.data_as(C.POINTER(C.c_int32))
    assert qc._lib.lib.qc_terms_desc_hess_nnz(C.byref(d), C.byref(nnz)) == qc._lib.QC_ERR_INVALID


def test_timesteps_all_equal_constraint(qc):
    inp = qc.config_inputs(1, T=6)
    con = qc.TimeStepsAllEqualConstraint("Δt", inp.traj)
    Z = inp.traj.datavec.copy()
    assert con.dim == 5 and np.all(con.g(Z) == 0.0)
    Z[con.indices[2]] += 0.25
    g = con.g(Z)
    assert g[2] == 0.25 and np.count_nonzero(g) == 1
    rows, cols = con.jac_structure
    J = np.zeros((con.dim, Z.size))
    np.add.at(J, (rows, cols), con.dg())
    np.testing.assert_allclose(J @ Z, g, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T,free_time,baseline,dt_scaled,D", [(1, 9, True, False, True, 0.0), (1, 50, True, True, True, 2.5),
                                                                  (2, 33, True, False, False, 1.0), (1, 7, False, True, True, 0.0),
                                                                  (1, 7, False, False, False, 0.0), (3, 1000, True, True, True, 0.7),
                                                                  (5, 64, True, False, True, 1.0)])
def test_terms_kernel_matches_oracle(qc, oracle, cfg, T, free_time, baseline, dt_scaled, D):
    traj, spec, tm, Z = make_terms(oracle, qc, cfg=cfg, T=T, free_time=free_time, baseline=baseline, dt_scaled=dt_scaled, D=D, seed=cfg)
    obj = qc.TrajectoryObjective(spec, traj, dt_scaled=dt_scaled)
    J, g, H = obj.L_grad_hess(Z)
    Jo = oracle.terms_value(tm, Z)
    assert abs(J - Jo) <= 1e-12 * max(1.0, abs(Jo))
    np.testing.assert_allclose(g, oracle.terms_grad(tm, Z), rtol=1e-13, atol=1e-15)
    r, c = oracle.terms_hess_structure(tm)
    np.testing.assert_array_equal(obj.hess_structure[0], r)
    np.testing.assert_array_equal(obj.hess_structure[1], c)
    np.testing.assert_allclose(H, oracle.terms_hess(tm, Z), rtol=1e-13, atol=1e-15)
    assert obj.L(Z) == J and getattr(obj, "∇L")(Z).tobytes() == g.tobytes()      # deterministic reduction
    obj.close()


@pytest.mark.gpu
def test_terms_device_entry_and_minimum_time_only(qc, oracle):
    import torch
    inp = qc.config_inputs(1, T=20)
    traj = inp.traj
    obj = qc.TrajectoryObjective(qc.MinimumTimeObjective(traj, 4.0), traj)
    Z = traj.datavec.copy()
    off = traj.offset("Δt")
    assert obj.hess_nnz == 0
    assert abs(obj.L(Z) - 4.0 * Z.reshape(20, traj.dim)[:-1, off].sum()) < 1e-13
    g = obj.grad_L(Z).reshape(20, traj.dim)
    assert np.all(g[:-1, off] == 4.0) and g[-1, off] == 0.0 and np.count_nonzero(g) == 19
    dZ = torch.from_numpy(Z).cuda()
    dJ = torch.zeros(1, dtype=torch.float64, device="cuda")
    dg = torch.full((Z.size,), float("nan"), dtype=torch.float64, device="cuda")
    obj.eval_device(dZ, dJ, dg)
    torch.cuda.synchronize()
    assert dJ.item() == obj.L(Z) and np.array_equal(dg.cpu().numpy().reshape(20, -1), g)
    # NaN in a state entry does not leak into the (zero) gradient of unregularised entries
    Z[3] = np.nan
    assert np.isfinite(obj.grad_L(Z)).all()
    obj.close()
