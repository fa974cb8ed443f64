"""One handle over several GPUs (qc_create_multi) and the descriptor options of round 2.

CPU part: shard arithmetic, dims and structures of descriptors with rows placed by state component (needs no GPU).
GPU part: a multi-device handle with several shards ON ONE DEVICE (device_ids = [0, 0, 0]: the only multi-shard layout a
1-GPU box can run) must return bit-identical F / dF / mu_d2F to the single-device handle, through the host-buffer entry
points (fan-out threads, per-shard pinned staging, direct-to-host compact transfer) and through the device-resident ones;
ShardedDynamics (the process-per-GPU flavour) with real HIP handles; the calling thread's current device is left alone."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle_bridge import problem_from_inputs, random_problem


# ------------------------------------------------------------------------------------------------
#  host-only
# ------------------------------------------------------------------------------------------------
def _script_layout(qc, T=6, nq=1, seed=0):
    """The trajectory of reference test/scripts/integrator_test_1qubit.jl:22-34: (U, a, g, da, dt), controls = (da,),
    integrators [Pade(U, a), Derivative(a, da)] -- the state component g has NO integrator."""
    rng = np.random.default_rng(seed)
    sys_ = qc.multi_qubit_system(nq) if nq > 1 else qc.QuantumSystem(qc.PAULIS["Z"], [qc.PAULIS["X"], qc.PAULIS["Y"]])
    N, m = sys_.levels, sys_.n_drives
    goal = qc.GATES["X"] if nq == 1 else np.eye(N, dtype=complex)[::-1]
    Z = qc.NamedTrajectory(
        {"Ũ⃗": qc.unitary_geodesic(np.eye(N, dtype=complex), goal, T) + 0.01 * rng.standard_normal((2 * N * N, T)),
         "a": rng.standard_normal((m, T)), "g": rng.standard_normal((m, T)), "da": rng.standard_normal((m, T)),
         "Δt": np.full((1, T), 0.1) + 0.01 * rng.random((1, T))},
        controls=("da",), timestep="Δt")
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", sys_, Z), qc.DerivativeIntegrator("a", "da", Z)]
    return sys_, Z, integ


def _oracle_by_component(o, inp_like, traj, qc):
    prob = problem_from_inputs(inp_like)
    prob.rows_per_interval = int(traj.dims.states)
    prob.row_offset = qc.state_row_offset(traj, "Ũ⃗")
    prob.deriv_rows = [qc.state_row_offset(traj, "a")]
    return prob


def test_rows_by_component_dims_and_structure(qc, oracle):
    """n_rows == Z.dims.states * (T - 1) when a state component has no integrator (integrator_test_script.jl:23-44): the rows
    of `g` exist and are structurally empty; structures equal the oracle's."""
    from types import SimpleNamespace
    sys_, Z, integ = _script_layout(qc)
    d, keep = qc.make_desc(integ, Z, rows="by_component")
    dims = qc.desc_dims(d)
    assert dims.n_rows == Z.dims.states * (Z.T - 1) == 12 * 5 and dims.n_cols == Z.dim * Z.T + Z.global_dim
    assert dims.ddim == 10 and dims.F_len == 12 * 5
    prob = _oracle_by_component(oracle, SimpleNamespace(integrators=integ, traj=Z), Z, qc)
    jr, jc, hr, hc = qc.desc_structures(d)
    orr, oc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, orr)
    np.testing.assert_array_equal(jc, oc)
    ohr, ohc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, ohr)
    np.testing.assert_array_equal(hc, ohc)
    used = np.unique(jr % 12)
    assert list(used) == list(range(10))          # rows 10, 11 (the component g) carry no entry
    # stacked rows (the default): 10 rows per interval, the same columns
    d2, _ = qc.make_desc(integ, Z)
    assert qc.desc_dims(d2).n_rows == 10 * 5
    # invalid placements are rejected
    bad, _ = qc.make_desc(integ, Z, rows="by_component")
    bad.deriv_row_off[0] = 4                       # overlaps the unitary rows 0..7
    out = qc._lib.qc_dims_t()
    assert qc._lib.lib.qc_desc_dims(C.byref(bad), C.byref(out)) == qc._lib.QC_ERR_INVALID
    bad2, _ = qc.make_desc(integ, Z, rows="by_component")
    bad2.rows_per_interval = 9
    assert qc._lib.lib.qc_desc_dims(C.byref(bad2), C.byref(out)) == qc._lib.QC_ERR_INVALID


def test_struct_sizes_are_exported(qc):
    L = qc._lib
    assert L.lib.qc_sizeof_desc() == C.sizeof(L.qc_desc) and L.lib.qc_sizeof_dims() == C.sizeof(L.qc_dims_t)
    assert L.lib.qc_sizeof_terms_desc() == C.sizeof(L.qc_terms_desc)


@pytest.mark.skipif(torch.cuda.is_available(), reason="this box has a GPU")
def test_multi_create_fails_loudly_without_a_device(qc):
    inp = qc.config_inputs(1, T=5)
    with pytest.raises(qc.QCollocError) as e:
        qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0, 0])
    assert e.value.code == qc._lib.QC_ERR_NO_DEVICE
    d, keep = qc.make_desc(inp.integrators, inp.traj)
    h = C.c_void_p()
    assert qc._lib.lib.qc_create_multi(C.byref(d), 0, (C.c_int32 * 1)(0), C.byref(h)) == qc._lib.QC_ERR_INVALID
    assert qc._lib.lib.qc_multi_count(None) == 0


# ------------------------------------------------------------------------------------------------
#  GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T,shards,full_from,integrator", [(3, 100, 3, None, "pade"), (3, 5, 8, None, "pade"), (1, 50, 2, None, "pade"),
                                                               (5, 21, 3, None, "pade"), (2, 37, 4, None, "pade"), (3, 150, 4, "1", "pade"),
                                                               (3, 40, 3, None, "exponential"), (5, 9, 2, None, "exponential")])
def test_multi_handle_equals_single_handle(qc, oracle, cfg, T, shards, full_from, integrator, monkeypatch):
    """device_ids = [0] * shards: F, dF, mu_d2F, structures and dims bit-identical to the single-device handle (T = 5 with
    8 shards leaves empty trailing shards).  full_from: QC_HOST_MULTI_FULL -- the shards copy the Jacobian values in full (what a
    handle over four or more distinct devices does by default: N links beat one host's replication) instead of in the compact form."""
    if full_from is not None:
        monkeypatch.setenv("QC_HOST_MULTI_FULL", full_from)
    inp = qc.config_inputs(cfg, T=T, integrator=integrator)     # (exponential: its mu_d2F too, round 6)
    Z = inp.traj.datavec
    one = qc.QuantumDynamics(inp.integrators, inp.traj)
    many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0] * shards)
    assert many.n_shards == shards and one.n_shards == 0
    for f in ("n_rows", "n_cols", "ddim", "jac_nnz_interval", "hess_nnz_interval", "n_intervals", "F_len", "jac_nnz", "hess_nnz", "Z_len"):
        assert getattr(one.dims, f) == getattr(many.dims, f), f
    chunk = -(-(T - 1) // shards)
    for i in range(shards):
        dev, t0, t1 = many.shard_info(i)
        assert dev == 0 and t0 == min(i * chunk, T - 1) and t1 == min((i + 1) * chunk, T - 1)
    for a, b in zip(one._structure(), many._structure()):
        np.testing.assert_array_equal(a, b)
    F1, J1 = one.F_dF(Z)
    F2, J2 = many.F_dF(Z)
    np.testing.assert_array_equal(F1, F2)
    np.testing.assert_array_equal(J1, J2)
    np.testing.assert_array_equal(one.F(Z), many.F(Z))
    np.testing.assert_array_equal(one.dF(Z), many.dF(Z))
    mu = np.random.default_rng(cfg).standard_normal(one.dims.n_rows)
    np.testing.assert_array_equal(one.mu_d2F(Z, mu), many.mu_d2F(Z, mu))
    # a second point, results of the first still held by the caller
    Zb = Z + 1e-3 * np.random.default_rng(1).standard_normal(Z.size)
    F3, J3 = many.F_dF(Zb)
    np.testing.assert_array_equal(F3, one.F_dF(Zb)[0])
    assert not np.array_equal(J3, J2) and np.array_equal(J2, J1)
    # and against the oracle
    prob = problem_from_inputs(inp)
    np.testing.assert_allclose(F2, oracle.F(prob, Z), rtol=1e-10, atol=1e-12)
    if integrator == "exponential" and T <= 12:
        H = oracle.mu_d2F(prob, Z, mu)
        np.testing.assert_allclose(many.mu_d2F(Z, mu), H, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(H).max()))
    one.close()
    many.close()


@pytest.mark.gpu
def test_multi_handle_device_resident_and_all_gather(qc, oracle):
    """qc_multi_eval_*_dev writes every shard's slice into full-length vectors; the in-library RCCL all-gather runs with one
    rank on this 1-GPU box (communicator creation, grouped in-place call) and refuses repeated devices."""
    L = qc._lib
    inp = qc.config_inputs(3, T=40)
    Z = inp.traj.datavec
    one = qc.QuantumDynamics(inp.integrators, inp.traj)
    F1, J1 = one.F_dF(Z)
    mu = np.random.default_rng(3).standard_normal(one.dims.n_rows)
    H1 = one.mu_d2F(Z, mu)
    for shards in (1, 3):
        many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0] * shards)
        nj, nf, nh = int(one.dims.jac_nnz_interval), int(one.dims.ddim), int(one.dims.hess_nnz_interval)
        lenJ = int(L.lib.qc_multi_padded_len(many._h, nj))
        assert lenJ == -(-39 // shards) * shards * nj
        dZ = torch.from_numpy(Z).cuda()
        dmu = torch.from_numpy(mu).cuda()
        dJ = [torch.full((lenJ,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
        dF = [torch.full((int(L.lib.qc_multi_padded_len(many._h, nf)),), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
        dH = [torch.full((int(L.lib.qc_multi_padded_len(many._h, nh)),), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
        torch.cuda.synchronize()
        arr = lambda ts: (C.c_void_p * shards)(*[t.data_ptr() for t in ts])
        L.check(L.lib.qc_multi_eval_F_jac_dev(many._h, arr([dZ] * shards), arr(dF), arr(dJ)), many._h)
        L.check(L.lib.qc_multi_eval_hess_dev(many._h, arr([dZ] * shards), arr([dmu] * shards), arr(dH)), many._h)
        L.check(L.lib.qc_multi_sync(many._h), many._h)
        chunk = -(-39 // shards)
        for i in range(shards):      # every shard wrote exactly its own slice of its vector
            lo, hi = min(i * chunk, 39), min((i + 1) * chunk, 39)
            np.testing.assert_array_equal(dJ[i].cpu().numpy()[lo * nj:hi * nj], J1[lo * nj:hi * nj])
            np.testing.assert_array_equal(dF[i].cpu().numpy()[lo * nf:hi * nf], F1[lo * nf:hi * nf])
            np.testing.assert_array_equal(dH[i].cpu().numpy()[lo * nh:hi * nh], H1[lo * nh:hi * nh])
            rest = np.concatenate([dJ[i].cpu().numpy()[:lo * nj], dJ[i].cpu().numpy()[hi * nj:]])
            assert np.isnan(rest).all()
        rc = L.lib.qc_multi_all_gather_dev(many._h, arr(dJ), nj)
        if shards == 1:
            L.check(rc, many._h)      # one rank: RCCL loaded, communicator built, in-place gather = identity
            L.check(L.lib.qc_multi_sync(many._h), many._h)
            np.testing.assert_array_equal(dJ[0].cpu().numpy()[:J1.size], J1)
        else:
            assert rc == L.QC_ERR_UNSUPPORTED and b"distinct devices" in L.lib.qc_last_error(many._h)
        # the single-device entry points refuse a multi handle instead of guessing a device
        assert L.lib.qc_eval_F_jac_dev(many._h, C.c_void_p(dZ.data_ptr()), None, C.c_void_p(dJ[0].data_ptr()), None) == L.QC_ERR_INVALID
        many.close()
    one.close()


@pytest.mark.gpu
def test_config4_partition_at_full_size_every_value_against_the_c_oracle(qc, coracle):
    """BASELINE config 4 -- 3-qubit Toffoli, T = 8000, knot-sharded 8 ways -- with the eight shards on this box's one device
    (`devices = [0] * 8`: the partition, the halo knots, the per-shard uploads / launches / slices are those of the 8-GPU handle;
    only the links are shared).  Host path: F_dF, mu_d2F, F -- EVERY value against oracle/qc_oracle.c and bit-identical to the
    single handle.  Device path: qc_multi_eval_F_jac_dev / _hess_dev write each shard's slice of full-length vectors.
    (Intervals are independent given Z: unitary_smooth_pulse_problem.jl:14-16.)"""
    from oracle_bridge import problem_from_inputs
    L = qc._lib
    shards = 8
    inp = qc.config_inputs(4)
    T = inp.traj.T
    assert T == 8000
    prob = problem_from_inputs(inp)
    prob.hess_align = 1
    co = coracle.COracle(prob)
    rng = np.random.default_rng(4)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(prob.n_rows)
    Fr, Jr = co.F_dF(Z)
    Hr = co.mu_d2F(Z, mu)

    def close(got, ref, what, atol=1e-12):
        scale = max(1.0, float(np.max(np.abs(ref))))
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=atol * scale, err_msg=what)

    many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0] * shards)
    assert many.n_shards == shards
    chunk = -(-(T - 1) // shards)
    assert [many.shard_info(i)[1:] for i in range(shards)] == [(min(i * chunk, T - 1), min((i + 1) * chunk, T - 1)) for i in range(shards)]
    Fm, Jm = many.F_dF(Z)
    Hm = many.mu_d2F(Z, mu)
    F_only = many.F(Z)
    close(Fm, Fr, "config 4, 8 shards: F")
    close(Jm, Jr, "config 4, 8 shards: dF")
    close(Hm, Hr, "config 4, 8 shards: mu_d2F", atol=1e-11)
    np.testing.assert_array_equal(F_only, Fm)
    del Fr, Jr, Hr
    one = qc.QuantumDynamics(inp.integrators, inp.traj)
    F1, J1 = one.F_dF(Z)
    H1 = one.mu_d2F(Z, mu)
    assert np.array_equal(F1, Fm) and np.array_equal(J1, Jm) and np.array_equal(H1, Hm)
    assert np.array_equal(one.F(Z), F_only)
    one.close()
    # device-resident: every shard writes its own slice of its own full-length vector (here: eight vectors on one device)
    nj, nf, nh = int(many.dims.jac_nnz_interval), int(many.dims.ddim), int(many.dims.hess_nnz_interval)
    dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
    lens = {k: int(L.lib.qc_multi_padded_len(many._h, k)) for k in (nj, nf, nh)}
    assert lens[nj] == chunk * shards * nj
    dJ = [torch.full((lens[nj],), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
    dF = [torch.full((lens[nf],), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
    dH = [torch.full((lens[nh],), float("nan"), dtype=torch.float64, device="cuda") for _ in range(shards)]
    torch.cuda.synchronize()
    arr = lambda ts: (C.c_void_p * shards)(*[t.data_ptr() for t in ts])
    L.check(L.lib.qc_multi_eval_F_jac_dev(many._h, arr([dZ] * shards), arr(dF), arr(dJ)), many._h)
    L.check(L.lib.qc_multi_eval_hess_dev(many._h, arr([dZ] * shards), arr([dmu] * shards), arr(dH)), many._h)
    L.check(L.lib.qc_multi_sync(many._h), many._h)
    for i in range(shards):
        lo, hi = min(i * chunk, T - 1), min((i + 1) * chunk, T - 1)
        assert torch.equal(dJ[i][lo * nj:hi * nj].cpu(), torch.from_numpy(Jm[lo * nj:hi * nj])), f"shard {i}: dF slice"
        assert torch.equal(dF[i][lo * nf:hi * nf].cpu(), torch.from_numpy(Fm[lo * nf:hi * nf])), f"shard {i}: F slice"
        assert torch.equal(dH[i][lo * nh:hi * nh].cpu(), torch.from_numpy(Hm[lo * nh:hi * nh])), f"shard {i}: mu_d2F slice"
        assert bool(torch.isnan(dJ[i][:lo * nj]).all()) and bool(torch.isnan(dJ[i][hi * nj:]).all()), f"shard {i} wrote outside its slice"
    many.close()


@pytest.mark.gpu
@pytest.mark.parametrize("full_size", [False, True])
def test_multi_handle_on_distinct_devices_with_rccl_all_gather(qc, oracle, full_size):
    """Runs wherever at least two GPUs are visible (the driver's 8-GPU box; skipped on the 1-GPU boxes): one shard per device,
    qc_multi_eval_*_dev on each device's own vectors, then the in-library RCCL all-gather with one rank per device -- afterwards
    EVERY device holds the full value vector, bit-identical to a single-device evaluation; the host-buffer entry points of the
    same handle (N PCIe links) return the same arrays."""
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least two GPUs")
    L = qc._lib
    # The ONLY environment excuse besides "< 2 devices": librccl cannot be opened at all.  torch's own copy is in this process
    # already when torch was built with RCCL (the library prefers that copy); else the system one must load.
    rccl_loadable = any("librccl.so" in line for line in open("/proc/self/maps"))
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        if rccl_loadable:
            break
        try:
            C.CDLL(name, mode=C.RTLD_GLOBAL)
            rccl_loadable = True
        except OSError:
            pass
    shards = min(ndev, 8)
    # full_size: BASELINE config 4's trajectory (T = 8000 over eight devices; 1000 knots per device on a smaller box) -- the first
    # box with several GPUs finds the all-gather of 42.6 MB per device waiting at the size the metric is quoted on
    T = (8000 if shards == 8 else 1000 * shards) if full_size else 64 * shards + 1 + 5          # (small: a short last shard)
    inp = qc.config_inputs(3, T=T)
    Z = inp.traj.datavec
    one = qc.QuantumDynamics(inp.integrators, inp.traj, device=0)
    F1, J1 = one.F_dF(Z)
    mu = np.random.default_rng(9).standard_normal(one.dims.n_rows)
    H1 = one.mu_d2F(Z, mu)
    many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=list(range(shards)))
    assert many.n_shards == shards and [many.shard_info(i)[0] for i in range(shards)] == list(range(shards))
    # host buffers through N devices
    Fm, Jm = many.F_dF(Z)
    np.testing.assert_array_equal(Fm, F1)
    np.testing.assert_array_equal(Jm, J1)
    np.testing.assert_array_equal(many.mu_d2F(Z, mu), H1)
    np.testing.assert_array_equal(many.F(Z), F1)
    # device-resident + all-gather
    nj, nf, nh = int(one.dims.jac_nnz_interval), int(one.dims.ddim), int(one.dims.hess_nnz_interval)
    lens = {k: int(L.lib.qc_multi_padded_len(many._h, k)) for k in (nj, nf, nh)}
    dZ, dmu, dJ, dF, dH = [], [], [], [], []
    for i in range(shards):
        dev = torch.device("cuda", i)
        dZ.append(torch.from_numpy(Z).to(dev))
        dmu.append(torch.from_numpy(mu).to(dev))
        dJ.append(torch.full((lens[nj],), float("nan"), dtype=torch.float64, device=dev))
        dF.append(torch.full((lens[nf],), float("nan"), dtype=torch.float64, device=dev))
        dH.append(torch.full((lens[nh],), float("nan"), dtype=torch.float64, device=dev))
    for i in range(shards):
        torch.cuda.synchronize(i)      # qcolloc.h: inputs complete before qc_multi_eval_*_dev (internal streams)
    arr = lambda ts: (C.c_void_p * shards)(*[t.data_ptr() for t in ts])
    L.check(L.lib.qc_multi_eval_F_jac_dev(many._h, arr(dZ), arr(dF), arr(dJ)), many._h)
    L.check(L.lib.qc_multi_eval_hess_dev(many._h, arr(dZ), arr(dmu), arr(dH)), many._h)
    L.check(L.lib.qc_multi_sync(many._h), many._h)
    chunk = -(-(T - 1) // shards)
    for i in range(shards):          # every shard wrote its own slice on its own device
        lo, hi = min(i * chunk, T - 1), min((i + 1) * chunk, T - 1)
        np.testing.assert_array_equal(dJ[i].cpu().numpy()[lo * nj:hi * nj], J1[lo * nj:hi * nj])
        np.testing.assert_array_equal(dF[i].cpu().numpy()[lo * nf:hi * nf], F1[lo * nf:hi * nf])
        np.testing.assert_array_equal(dH[i].cpu().numpy()[lo * nh:hi * nh], H1[lo * nh:hi * nh])
    for bufs, per in ((dJ, nj), (dF, nf), (dH, nh)):
        rc = L.lib.qc_multi_all_gather_dev(many._h, arr(bufs), per)
        if rc != L.QC_OK and not rccl_loadable:
            pytest.skip("librccl cannot be opened on this box: " + L.lib.qc_last_error(many._h).decode())
        # with >= 2 devices visible and librccl loadable, anything but QC_OK is a broken collective, not an environment matter
        assert rc == L.QC_OK, "qc_multi_all_gather_dev failed with librccl loadable and %d devices visible: %s" % (ndev, L.lib.qc_last_error(many._h).decode())
    L.check(L.lib.qc_multi_sync(many._h), many._h)
    for i in range(shards):
        np.testing.assert_array_equal(dJ[i].cpu().numpy()[:J1.size], J1)
        np.testing.assert_array_equal(dF[i].cpu().numpy()[:F1.size], F1)
        np.testing.assert_array_equal(dH[i].cpu().numpy()[:H1.size], H1)
    assert torch.cuda.current_device() == 0      # the library left the caller's device alone
    many.close()
    one.close()


@pytest.mark.gpu
def test_sharded_dynamics_with_real_handles(qc, oracle):
    """The process-per-GPU flavour (sharding.ShardedDynamics) with its default rank-local evaluator, the HIP handle: two
    and three ranks' shards on device 0, concatenated = the full evaluation; an empty tail shard is a valid no-op handle."""
    from qcolloc_amd.sharding import ShardedDynamics
    inp = qc.config_inputs(3, T=30)
    Z = inp.traj.datavec
    full = qc.QuantumDynamics(inp.integrators, inp.traj)
    F, J = full.F_dF(Z)
    mu = np.random.default_rng(2).standard_normal(full.dims.n_rows)
    H = full.mu_d2F(Z, mu)
    for world in (2, 3):
        parts = [ShardedDynamics(inp.integrators, inp.traj, r, world, device=0) for r in range(world)]
        assert [p.n_local for p in parts] == [t1 - t0 for t0, t1 in parts[0].shards]
        np.testing.assert_array_equal(np.concatenate([p.local.F_dF(Z)[0] for p in parts]), F)
        np.testing.assert_array_equal(np.concatenate([p.local.F_dF(Z)[1] for p in parts]), J)
        np.testing.assert_array_equal(np.concatenate([p.local.mu_d2F(Z, mu) for p in parts]), H)
        for p in parts:
            p.local.close()
    inp2 = qc.config_inputs(1, T=3)          # 2 intervals on 4 ranks: ranks 2 and 3 are empty
    parts = [ShardedDynamics(inp2.integrators, inp2.traj, r, 4, device=0) for r in range(4)]
    assert [p.empty for p in parts] == [False, False, True, True]
    assert parts[3].local.dims.n_intervals == 0 and parts[3].local.F_dF(inp2.traj.datavec)[1].size == 0
    dz = torch.from_numpy(inp2.traj.datavec).cuda()
    parts[3].local.F_dF_device(dz, torch.empty(1, dtype=torch.float64, device="cuda"), torch.empty(1, dtype=torch.float64, device="cuda"))
    for p in parts:
        p.local.close()
    full.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [1, 3, 4])      # (4 qubits: the sparse-drive kernels of qc_mfma32_ell.hip, one derivative integrator)
def test_rows_by_component_on_the_device(qc, oracle, nq):
    """The reference harness layout (state component g without an integrator): F has Z.dims.states rows per interval with
    zeros in the empty rows, mu of that length is accepted, values equal the oracle's; also through a multi-device handle."""
    from types import SimpleNamespace
    sys_, Z, integ = _script_layout(qc, T=9, nq=nq, seed=nq)
    prob = _oracle_by_component(oracle, SimpleNamespace(integrators=integ, traj=Z), Z, qc)
    zv = Z.datavec
    mu = np.random.default_rng(7).standard_normal(Z.dims.states * (Z.T - 1))
    for devices in (None, [0, 0, 0]):
        for kernel in ("auto", "lds"):
            dyn = qc.QuantumDynamics(integ, Z, rows="by_component", devices=devices, kernel=kernel)
            assert dyn.dims.n_rows == Z.dims.states * (Z.T - 1) == mu.size
            F, J = dyn.F_dF(zv)
            Fr = oracle.F(prob, zv)
            np.testing.assert_allclose(F, Fr, rtol=1e-10, atol=1e-12)
            s, m = 2 * sys_.levels ** 2, sys_.n_drives
            assert not F.reshape(Z.T - 1, -1)[:, s + m:].any()          # rows of g: structurally empty, delivered as 0
            np.testing.assert_allclose(J, oracle.dF(prob, zv), rtol=1e-10, atol=1e-12)
            np.testing.assert_array_equal(dyn.F(zv), F)
            H = dyn.mu_d2F(zv, mu)
            Hr = oracle.mu_d2F(prob, zv, mu)
            np.testing.assert_allclose(H, Hr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Hr).max()))
            jr, jc = dyn.dF_structure
            orr, oc = oracle.jac_structure(prob)
            np.testing.assert_array_equal(jr, orr)
            np.testing.assert_array_equal(jc, oc)
            dyn.close()


@pytest.mark.gpu
def test_entry_points_leave_the_current_device_alone(qc):
    """Every entry point selects its handle's device itself and restores the caller's: with one GPU the only
    observable part is that nothing moves; the 2-GPU half runs when a second device is visible."""
    inp = qc.config_inputs(1, T=8)
    Z = inp.traj.datavec
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(0)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, device=ndev - 1)
    assert torch.cuda.current_device() == 0
    dyn.F_dF(Z)
    dyn.mu_d2F(Z, np.ones(dyn.dims.n_rows))
    assert torch.cuda.current_device() == 0
    if ndev > 1:
        dz = torch.from_numpy(Z).to(f"cuda:{ndev - 1}")
        dF = torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device=dz.device)
        dJ = torch.empty(int(dyn.dims.jac_nnz), dtype=torch.float64, device=dz.device)
        dyn.F_dF_device(dz, dF, dJ, stream=torch.cuda.current_stream(ndev - 1))     # handle on device 1 while device 0 is current
        torch.cuda.synchronize(ndev - 1)
        assert torch.cuda.current_device() == 0
        np.testing.assert_array_equal(dJ.cpu().numpy(), dyn.dF(Z))
        many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=list(range(ndev)))
        np.testing.assert_array_equal(many.dF(Z), dyn.dF(Z))
        assert torch.cuda.current_device() == 0
        many.close()
    dyn.close()
