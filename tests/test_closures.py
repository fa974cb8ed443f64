"""The vector-returning closures `dynamics.F(Z)`, `dynamics.dF(Z)`, `dynamics.mu_d2F(Z, mu)` -- the only call shapes the
reference's evaluator uses (reference test/scripts/integrator_test_1qubit.jl:45-52) -- and the upload elision (`set_new_x`) when the
handle behind them is shared: result-ring lifetime, `fresh=True`, and the handle's upload count (`qc_knot_generation`) as the guard."""
import os
import sys

import numpy as np
import pytest

from oracle_bridge import problem_from_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_retired_regulariser_ids_are_refused(qc):
    """ABI 0.4: QC_REG_* are 2 / 3; the values 0 and 1, which ABI 0.1 - 0.3 gave both meanings in turn, fail loudly (no GPU needed:
    the descriptor check runs before anything touches a device)."""
    import ctypes as C
    L = qc._lib
    assert (L.QC_REG_DT_SCALED, L.QC_REG_PLAIN) == (2, 3) and L.lib.qc_abi_version() == 6
    idx = np.array([8, 9], dtype=np.int32)
    R = np.ones(2)
    d = L.qc_terms_desc()
    d.T, d.zdim, d.off_dt, d.n_reg, d.min_time_knots = 5, 15, 14, 2, 4
    d.reg_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    d.reg_R = L.dptr(R)
    nnz = C.c_int64()
    for w, ok in ((L.QC_REG_DT_SCALED, True), (L.QC_REG_PLAIN, True), (0, False), (1, False), (4, False)):
        d.weighting = w
        rc = L.lib.qc_terms_desc_hess_nnz(C.byref(d), C.byref(nnz))
        assert (rc == 0) == ok, (w, rc)
    assert L.lib.qc_knot_generation(None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T", [(1, 20), (3, 12), (3, 300)])
def test_closure_results_are_the_callers_while_held(qc, oracle, cfg, T):
    """dynamics.F / dF / mu_d2F return vectors that stay intact for as long as the caller holds them -- a list of trial-point
    residuals, a finite-difference loop -- while the evaluator's use-and-drop pattern recycles a few pinned ones."""
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    prob = problem_from_inputs(inp)
    rng = np.random.default_rng(cfg)
    Zs = [inp.traj.datavec + 1e-2 * k * rng.standard_normal(inp.traj.datavec.size) for k in range(6)]
    mu = rng.standard_normal(int(dyn.dims.n_rows))
    want = {"F": [oracle.F(prob, Z) for Z in Zs], "dF": [oracle.dF(prob, Z) for Z in Zs], "H": [oracle.mu_d2F(prob, Z, mu) for Z in Zs]}
    for name, call in (("F", dyn.F), ("dF", dyn.dF), ("H", lambda Z: dyn.mu_d2F(Z, mu))):
        got = [call(Z) for Z in Zs]                           # six results held at once: more than the ring has
        assert len({g.ctypes.data for g in got}) == 6
        for g, w in zip(got, want[name]):                     # every one still holds its own values
            np.testing.assert_allclose(g, w, rtol=1e-10, atol=1e-13)
        del got, g
        seen = {call(Zs[k % 6]).ctypes.data for k in range(8)}   # use and drop: the ring's vectors come round
        assert len(seen) <= 3
    F, J = dyn.F_dF(Zs[2])
    np.testing.assert_allclose(F, want["F"][2], rtol=1e-10, atol=1e-13)
    # fresh=True: never one of the ring's
    keep = dyn.dF(Zs[0], fresh=True)
    np.testing.assert_allclose(keep, want["dF"][0], rtol=1e-10, atol=1e-13)
    # the ring's values are those of the caller-owned-buffer call, bit for bit
    mine = np.empty(int(dyn.dims.jac_nnz))
    dyn.dF(Zs[1], out=mine)
    np.testing.assert_array_equal(mine, dyn.dF(Zs[1]))
    dyn.close()
    assert dyn._rings == {}
    d0 = qc.QuantumDynamics(inp.integrators, inp.traj, result_ring=0)
    a, b = d0.dF(Zs[0]), d0.dF(Zs[1])
    assert a.ctypes.data != b.ctypes.data
    np.testing.assert_allclose(a, want["dF"][0], rtol=1e-10, atol=1e-13)
    d0.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,T,devices", [(1, 20, None), (3, 300, None), (5, 9, None), (3, 300, [0, 0, 0])])
def test_pinned_arrays_take_the_residuals_in_place(qc, oracle, cfg, T, devices):
    """qc_host_alloc: memory the library has pinned is written by the residual kernel directly (no device-to-host copy); same bits as
    the copy into an ordinary array; a slice of a block counts; a layout with rows no kernel writes keeps the copy (its zeros must
    be delivered); the closures' own vectors come from there."""
    import ctypes as C
    import gc
    L = qc._lib
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, devices=devices, result_ring=0)
    rng = np.random.default_rng(cfg)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    n = int(dyn.dims.F_len)
    plain = dyn.F(Z, out=np.full(n, np.nan))
    p = C.c_void_p()
    assert L.lib.qc_host_alloc((n + 64) * 8, C.byref(p)) == 0 and p.value
    big = np.ctypeslib.as_array((C.c_double * (n + 64)).from_address(p.value))
    big[:] = np.nan
    view = big[7:7 + n]
    dyn.F(Z, out=view)
    assert np.array_equal(view, plain) and np.isnan(big[:7]).all() and np.isnan(big[7 + n:]).all()
    F2, J2 = dyn.F_dF(Z, out=(view, np.empty(int(dyn.dims.jac_nnz))))               # the other calls accept pinned arrays as well
    assert np.array_equal(F2, plain)
    if dyn.dims.hess_nnz:
        mu = rng.standard_normal(int(dyn.dims.n_rows))
        Hp = qc.pinned_zeros(int(dyn.dims.hess_nnz))
        assert np.array_equal(dyn.mu_d2F(Z, mu, out=Hp), dyn.mu_d2F(Z, mu, out=np.empty(Hp.size)))
    assert L.lib.qc_host_free(C.c_void_p(p.value + 8)) != 0                          # not the start of a block
    del view, big, F2
    assert L.lib.qc_host_free(p) == 0 and L.lib.qc_host_free(p) != 0                 # (a second free of the same block is refused)
    dyn.close()
    # the closures' own vectors are pinned blocks (for sizes where it matters), released when the last reference goes
    d2 = qc.QuantumDynamics(inp.integrators, inp.traj, devices=devices)
    f = d2.F(Z)
    assert np.array_equal(f, plain)
    # (round 6: a returned vector sits on a lease object -- the ring's ownership token --, which holds the pinned block)
    assert type(f.base).__name__ == "_Lease" and (type(f.base.owner.base).__name__ == "_PinnedBlock") == (f.nbytes >= (64 << 10))
    d2.close()
    assert np.array_equal(f, plain)                   # a result outlives its evaluator
    del f, d2
    gc.collect()
    if cfg == 1 and devices is None:      # rows placed by component: a state component without an integrator leaves rows no kernel writes
        tr = inp.traj
        Tl = 600                          # (long enough for the residual vector to be worth pinning)
        inl = qc.config_inputs(1, T=Tl)
        tr = inl.traj
        comps = {nm: tr.data[r.start:r.stop] for nm, r in tr.components.items()}
        comps["g"] = rng.standard_normal((3, tr.T))
        names = list(comps)
        names.insert(1, names.pop(names.index("g")))
        traj = qc.NamedTrajectory({k: comps[k] for k in names}, controls=tr.controls, timestep=tr.timestep)
        integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", inl.system, traj, order=4), qc.DerivativeIntegrator("a", "da", traj),
                 qc.DerivativeIntegrator("da", "dda", traj)]
        d3 = qc.QuantumDynamics(integ, traj, rows="by_component", result_ring=0)
        ref = d3.F(traj.datavec, out=np.full(int(d3.dims.F_len), np.nan))
        reg = qc.pinned_zeros(int(d3.dims.F_len))
        assert type(reg.base).__name__ == "_PinnedBlock"
        reg[:] = np.nan
        d3.F(traj.datavec, out=reg)
        assert np.array_equal(reg, ref) and (ref.reshape(traj.T - 1, -1)[:, 8:11] == 0.0).all()
        d3.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_upload_elision_is_guarded_by_the_handles_upload_count(qc, oracle, devices):
    """between `eval_constraint(x)` and the Jacobian / Hessian at the same x, (a) a rollout through the same handle must
    not replace the knots on the device -- on a multi-device handle it used to replace shard 0's only --, and (b) any other
    host-buffer call through the shared handle (dyn.F at another point, a bound call) must make the evaluator upload again."""
    inp = qc.config_inputs(1, T=40)
    traj = inp.traj
    dyn = qc.QuantumDynamics(inp.integrators, traj, devices=devices)
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=100.0)
    ev = qc.QuantumControlEvaluator(dyn, [obj])
    prob = problem_from_inputs(inp)
    rng = np.random.default_rng(11)
    x = traj.datavec + 1e-2 * rng.standard_normal(traj.datavec.size)
    other = traj.datavec + 0.3 * rng.standard_normal(traj.datavec.size)
    lam = rng.standard_normal(ev.n_constraints)
    c, J, H = np.empty(ev.n_constraints), np.empty(ev.jac_nnz), np.empty(ev.hess_nnz)
    off, cnt = ev._hess_dyn
    J_ref, H_ref = oracle.dF(prob, x), oracle.mu_d2F(prob, x, lam[:ev.n_dynamics_rows])

    # (a) a rollout of ANOTHER trajectory vector between the residuals and the derivatives
    g0 = dyn.knot_generation()
    ev.eval_constraint(c, x)
    assert dyn.knot_generation() == g0 + 1
    init = other[:traj.dim][traj.offset("Ũ⃗"):traj.offset("Ũ⃗") + 8].copy()
    dyn.rollout(other, init)
    assert dyn.knot_generation() == g0 + 1                    # the rollout has a device buffer of its own
    ev.eval_constraint_jacobian(J, x)
    ev.eval_hessian_lagrangian(H, x, 1.0, lam)
    assert ev.stats["uploads_elided"] == 2                    # still elided -- and still right:
    np.testing.assert_allclose(J, J_ref, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(H[off:off + cnt], H_ref, rtol=1e-10, atol=1e-13)

    # (b) somebody else evaluates another point through the same handle
    ev.eval_constraint(c, x)                                  # (served from the cache: same x)
    dyn.F(other)
    elided = ev.stats["uploads_elided"]
    ev.eval_constraint_jacobian(J, x)
    assert ev.stats["uploads_elided"] == elided and ev.stats["F_dF"] == 1      # not elided: the fused call sent x up again
    np.testing.assert_allclose(J, J_ref, rtol=1e-10, atol=1e-13)
    bound_F = np.empty(int(dyn.dims.F_len))
    call = dyn.bind_host("F", other, F=bound_F)
    ev.eval_constraint(c, x)                                  # cached residuals of x
    assert call() == 0
    ev.eval_hessian_lagrangian(H, x, 1.0, lam)
    assert ev.stats["uploads_elided"] == elided
    np.testing.assert_allclose(H[off:off + cnt], H_ref, rtol=1e-10, atol=1e-13)
    # and the library itself is back at new_x = 1 after every evaluator call: a direct call at another point sees its own Z
    np.testing.assert_allclose(dyn.dF(other), oracle.dF(prob, other), rtol=1e-10, atol=1e-13)
    dyn.close()
    obj.close()


@pytest.mark.gpu
def test_deadline_turns_a_lost_copy_into_an_error(qc):
    """QC_HOST_TIMEOUT_MS: every host-side wait on the device has a deadline.  With a deadline of half a microsecond the watched copy of
    a config-3 evaluation (0.3 ms) cannot make it: the call must come back with QC_ERR_HIP and a message -- not hang, not crash --, and
    the handle must still close.  (The lock-free team itself is exercised with a stalling stand-in engine under the sanitizers on the
    CPU: tests/test_host_team.py.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import __graft_entry__ as g\n"
        "qc = g.load_package()\n"
        "inp = qc.config_inputs(3, T=1000)\n"
        "dyn = qc.QuantumDynamics(inp.integrators, inp.traj)\n"
        "for k in range(10):\n"                     # (the first calls spend milliseconds allocating pinned blocks, and a call whose thread the host's
                                                    #  CPU quota puts to sleep behind the launch finds its copy landed: no wait, no deadline -- seen once in ten runs)
        "    try:\n"
        "        dyn.F_dF(inp.traj.datavec)\n"
        "        print('NO ERROR')\n"
        "    except qc._lib.QCollocError as e:\n"
        "        print('ERROR:', e)\n"
        "dyn.close()\n"
        "print('closed')\n")
    env = dict(os.environ, QC_HOST_TIMEOUT_MS="0.0005")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.count("ERROR:") >= 2 and "timed out" in r.stdout and "QC_HOST_TIMEOUT_MS" in r.stdout, r.stdout
    assert "closed" in r.stdout


@pytest.mark.gpu
def test_handles_are_independent_across_threads(qc):
    """INTEGRATION.md "Ownership, errors, threading": a handle may be used from any thread (one evaluation in flight at a time), and
    DIFFERENT handles may be used at the same time -- Julia may run Ipopt's callbacks on another OS thread than the one that built the
    problem, and two solves in one process share the library's worker pool.  Three threads, each with a handle of its own (created on
    the main thread), hammer the host-buffer entry points concurrently; every result equals the single-threaded one bit for bit."""
    import threading
    cases = []
    for cfg, T in ((3, 400), (2, 150), (5, 40)):
        inp = qc.config_inputs(cfg, T=T)
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
        Z = inp.traj.datavec
        mu = np.random.default_rng(cfg).standard_normal(int(dyn.dims.n_rows))
        F, J = dyn.F_dF(Z, fresh=True)
        H = dyn.mu_d2F(Z, mu, fresh=True)
        cases.append((dyn, Z, mu, F, J, H))
    errors = []

    def work(case, reps):
        dyn, Z, mu, F, J, H = case
        try:
            Fo, Jo, Ho = np.empty_like(F), np.empty_like(J), np.empty_like(H)
            for r in range(reps):
                Fo[:] = 0.0
                dyn.F_dF(Z, out=(Fo, Jo))
                dyn.mu_d2F(Z, mu, out=Ho)
                if not (np.array_equal(Fo, F) and np.array_equal(Jo, J) and np.array_equal(Ho, H)):
                    errors.append(("values differ", int(dyn.dims.ddim), r))
                    return
                dyn.F(Z, out=Fo)
                if not np.array_equal(Fo, F):
                    errors.append(("F differs", int(dyn.dims.ddim), r))
                    return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(c, 25)) for c in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors
    for c in cases:
        c[0].close()


KNOBS = ["QC_HOST_COMPACT", "QC_FUSED_VARIANT", "QC_STORE_MODE", "QC_NO_FUSED", "QC_NO_ELL", "QC_NO_ANTISYM", "QC_HOST_WAIT", "QC_HOST_THREADS",
         "QC_HOST_TAPER", "QC_HOST_TAIL_SPLIT", "QC_HOST_STREAMS", "QC_HOST_REARM_JOBS", "QC_HOST_REARM", "QC_HOST_PIECE_KB", "QC_HOST_PIECES",
         "QC_HOST_NT", "QC_HOST_NOWATCH", "QC_HOST_NBUF", "QC_HOST_MULTI_FULL", "QC_HOST_LANDING", "QC_HOST_HESS_CHUNKS", "QC_HOST_F_DIRECT",
         "QC_HOST_F_CHUNKS", "QC_HOST_FILL_NT", "QC_HOST_CHUNKS", "QC_HOST_AFFINITY", "QC_HESS_TWO_WAVES", "QC_HESS_GRID", "QC_ELL_JAC",
         "QC_DEBUG_SKIP", "QC_HOST_TIMEOUT_MS", "QC_HOST_TRACE", "QC_EXP_ELL", "QC_HESS_G2", "QC_HESS_ELL", "QC_FUSED_ELL", "QC_LIST_BATCH"]
_KNOB_SCRIPT = r"""
import sys, json
sys.path[:0] = [%(root)r, %(tests)r]
import numpy as np
import __graft_entry__ as g
qc = g.load_package()
out = {}
for cfg, T in ((3, 70), (5, 6), (1, 9)):
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    mu = np.cos(np.arange(int(dyn.dims.n_rows)))
    for rep in range(3):                     # (the ring of pinned blocks turns)
        F, J = dyn.F_dF(Z, fresh=True)
        H = dyn.mu_d2F(Z, mu, fresh=True)
        F1 = dyn.F(Z, fresh=True)
    dyn.set_new_x(False)
    J2 = dyn.dF(Z, fresh=True)
    dyn.set_new_x(True)
    many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0, 0, 0])
    Fm, Jm = many.F_dF(Z, fresh=True)
    w = lambda a, k: float(np.sum(a * (1 + np.arange(a.size) %% k)))
    out[str(cfg)] = [w(F, 7), w(J, 11), w(H, 13), w(F1, 7), w(J2, 11), w(Fm, 7), w(Jm, 11)]
    many.close(); dyn.close()
base = qc.multi_qubit_system(1)
lst = qc.unitary_sampling_inputs([base, qc.QuantumSystem(base.H_drift * 1.3, base.H_drives)], qc.GATES["H"], 40)
dyn = qc.QuantumDynamics(lst.integrators, lst.traj)
F, J = dyn.F_dF(lst.traj.datavec, fresh=True)
out["list"] = [float(np.sum(F * (1 + np.arange(F.size) %% 7))), float(np.sum(J * (1 + np.arange(J.size) %% 11)))]
dyn.close()
for nq in (2, 4):          # the exponential integrator (round 6: row-gather / dense-image forms, 2N = 8 and 32)
    inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(nq), qc.GATES["CNOT" if nq == 2 else "QFT16"], 5, integrator="exponential")
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    F, J = dyn.F_dF(Z, fresh=True)
    H = dyn.mu_d2F(Z, np.cos(np.arange(dyn.dims.n_rows)), fresh=True)
    w = lambda a, k: float(np.sum(a * (1 + np.arange(a.size) %% k)))
    out["exp%%d" %% nq] = [w(F, 7), w(J, 11), w(H, 13)]
    dyn.close()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_environment_knobs_with_nonsense_values_never_break_a_result(qc):
    """the product reads ~30 QC_* environment knobs and validated none.  Every knob set to the same nonsense value at once
    (0, a negative number, text, a huge number): the evaluations either run to the default run's values (rtol 1e-10 -- a knob may pick
    another kernel or transfer path, never another answer) or fail with a library error; no crash, no hang."""
    import json
    import subprocess
    script = _KNOB_SCRIPT % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}

    def run(value):
        env = {k: v for k, v in os.environ.items() if not k.startswith("QC_")}
        if value is not None:
            env.update({k: value for k in KNOBS})
            if value in ("1", "2"):              # (a deadline of 1 - 2 ms is a VALID setting, and too short for a first call)
                del env["QC_HOST_TIMEOUT_MS"]
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (value, r.stderr[-1500:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert line, (value, r.stdout[-500:], r.stderr[-500:])
        return json.loads(line[-1][7:])

    ref = run(None)
    for value in ("0", "-7", "abc", "1000000000", "1", "2"):
        got = run(value)
        for key in ref:
            np.testing.assert_allclose(got[key], ref[key], rtol=1e-10, err_msg=f"knobs = {value!r}, {key}")
