"""The integrator lists of the reference's other templates on the path: `UnitaryBangBangProblem` ([U, D] over a trajectory
with one control derivative and the L1 slack components, unitary_bang_bang_problem.jl:102-121,149-152,163-175; its own test
runs Pade order 12 with `control_name=:u`, :205-215) and `UnitaryDirectSumProblem` ([U_1, D, D, U_2, D, D, ...] over the
members' merged, suffixed trajectories, every member with its own controls, unitary_direct_sum_problem.jl:104,127-130).

CPU part: the oracle on these lists against the members evaluated on their own trajectories and against finite differences.
GPU part (`-m gpu`): the HIP path through the C ABI against the oracle, rtol 1e-10, structures array_equal."""
import numpy as np
import pytest

from oracle_bridge import composed_oracle, problem_from_inputs

RTOL = 1e-10


def close(got, ref, what="", atol=1e-12):
    scale = max(1.0, float(np.max(np.abs(ref)))) if ref.size else 1.0
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=atol * scale, err_msg=what)


def direct_sum_members(qc, free_time, T=7, orders=(4, 4), exp_second=False):
    p1 = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(1), qc.GATES["X"], T, free_time=free_time, pade_order=orders[0])
    kw = dict(integrator="exponential") if exp_second else dict(pade_order=orders[1])
    p2 = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CNOT"], T, free_time=free_time, seed=3, **kw)
    if free_time:   # the direct sum shares one timestep row: give the members the same one
        p2.traj.data[p2.traj.components["Δt"].start, :] = p1.traj["Δt"][0]
    return p1, p2


# ------------------------------------------------------------------------------------------------
#  CPU: the lists, and the oracle over them
# ------------------------------------------------------------------------------------------------
def test_integrator_lists_of_every_template_split_into_groups(qc):
    s1 = qc.multi_qubit_system(1)
    smooth = qc.unitary_smooth_pulse_inputs(s1, qc.GATES["H"], 5)
    assert [len(g) for g in qc.split_groups(smooth.integrators)] == [3]
    samp = qc.unitary_sampling_inputs([s1, s1, s1], qc.GATES["H"], 5)
    assert [len(g) for g in qc.split_groups(samp.integrators)] == [1, 1, 3]
    bang = qc.unitary_bang_bang_inputs(s1, qc.GATES["H"], 5, control_name="u")
    assert [len(g) for g in qc.split_groups(bang.integrators)] == [2]
    assert bang.traj.names == ("Ũ⃗", "u", "du", "Δt", "s1_du", "s2_du")
    assert bang.traj.dims.states == 8 + 2                     # the slacks are controls: no dynamics rows
    ds = qc.unitary_direct_sum_inputs(direct_sum_members(qc, False))
    assert [len(g) for g in qc.split_groups(ds.integrators)] == [3, 3]
    assert ds.traj.names == ("Ũ⃗1", "a1", "da1", "dda1", "Ũ⃗2", "a2", "da2", "dda2")
    assert ds.traj.dims.states == (8 + 4) + (32 + 8)
    with pytest.raises(NotImplementedError):
        qc.split_groups(smooth.integrators[1:])               # a list starting with a derivative integrator
    with pytest.raises(ValueError):
        qc.unitary_direct_sum_inputs([smooth])                # "At least two problems are required" (:69)
    bb = qc.unitary_bang_bang_inputs(s1, qc.GATES["H"], 5)
    with pytest.raises(ValueError):
        qc.unitary_direct_sum_inputs([bb, bb])                # "Only smooth pulse problems are supported." (:72)


@pytest.mark.parametrize("free_time", [False, True])
def test_direct_sum_oracle_equals_members_on_their_own(qc, oracle, free_time):
    """A direct sum is its members side by side: values equal the members' own evaluations, and the structure is the
    members' structure with rows / columns moved to the merged trajectory's positions."""
    p1, p2 = direct_sum_members(qc, free_time)
    ds = qc.unitary_direct_sum_inputs([p1, p2], labels=["a", "b"])
    ref = composed_oracle(ds)
    Z = ds.traj.datavec
    T = ds.traj.T
    members = [(problem_from_inputs(p), p) for p in (p1, p2)]
    Fm = [oracle.F(pr, p.traj.datavec).reshape(T - 1, -1) for pr, p in members]
    np.testing.assert_array_equal(ref.F(Z), np.concatenate(Fm, axis=1).reshape(-1))
    Jm = [oracle.dF(pr, p.traj.datavec).reshape(T - 1, -1) for pr, p in members]
    np.testing.assert_array_equal(ref.dF(Z), np.concatenate(Jm, axis=1).reshape(-1))
    # structure: dense Jacobian of the sum = block placement of the members' dense Jacobians
    rr, rc = ref.structure()
    J = np.zeros((ref.rows * (T - 1), Z.size))
    np.add.at(J, (rr, rc), ref.dF(Z))
    ro = 0
    for (pr, p), lab in zip(members, "ab"):
        r, c = oracle.jac_structure(pr)
        Jd = np.zeros((pr.n_rows, p.traj.datavec.size))
        np.add.at(Jd, (r, c), oracle.dF(pr, p.traj.datavec))
        # member column (knot t, row i of component nm) -> merged column
        col = np.empty(p.traj.dim, dtype=np.int64)
        for nm in p.traj.names:
            tgt = nm if nm == p.traj.timestep else nm + lab
            col[list(p.traj.components[nm])] = list(ds.traj.components[tgt])
        rows = (np.arange(T - 1)[:, None] * ref.rows + ro + np.arange(pr.ddim)[None, :]).reshape(-1)
        cols = (np.arange(T)[:, None] * ds.traj.dim + col[None, :]).reshape(-1)
        np.testing.assert_array_equal(J[np.ix_(rows, cols)], Jd)
        J[np.ix_(rows, cols)] = 0.0
        ro += pr.ddim
    assert not J.any()                                         # nothing outside the members' blocks
    # Lagrangian Hessian: sum of the members' (the shared timestep row collects both)
    rng = np.random.default_rng(0)
    mu = rng.standard_normal(ref.rows * (T - 1))
    hr, hc = ref.hess_structure()
    H = np.zeros((Z.size, Z.size))
    np.add.at(H, (hr, hc), ref.mu_d2F(Z, mu))
    eps = 1e-6
    v = rng.standard_normal(Z.size)

    def grad(z):
        g = np.zeros(Z.size)
        np.add.at(g, rc, ref.dF(z) * mu[rr])
        return g
    Hs = H + H.T - np.diag(np.diag(H))
    np.testing.assert_allclose(Hs @ v, (grad(Z + eps * v) - grad(Z - eps * v)) / (2 * eps), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("order,free_time", [(12, True), (4, False)])
def test_bang_bang_oracle_never_touches_the_slacks(qc, oracle, order, free_time):
    inp = qc.unitary_bang_bang_inputs(qc.multi_qubit_system(1), qc.GATES["H"], 9, pade_order=order, free_time=free_time, control_name="u")
    prob = problem_from_inputs(inp)
    assert prob.ddim == inp.traj.dims.states == 10 and prob.order == order and len(prob.derivs) == 1
    Z = inp.traj.datavec
    r, c = oracle.jac_structure(prob)
    slack = np.concatenate([np.array(inp.traj.components[n]) for n in ("s1_du", "s2_du")])
    assert not np.isin(c % inp.traj.dim, slack).any()
    hr, hc = oracle.hess_structure(prob)
    assert not np.isin(hr % inp.traj.dim, slack).any() and not np.isin(hc % inp.traj.dim, slack).any()
    rng = np.random.default_rng(1)
    Z2 = Z.copy().reshape(inp.traj.T, inp.traj.dim)
    Z2[:, slack] = rng.standard_normal((inp.traj.T, slack.size))
    np.testing.assert_array_equal(oracle.F(prob, Z2.reshape(-1)), oracle.F(prob, Z))
    v = rng.standard_normal(Z.size)
    eps = 1e-6
    Jv = np.zeros(prob.n_rows)
    np.add.at(Jv, r, oracle.dF(prob, Z) * v[c])
    np.testing.assert_allclose(Jv, (oracle.F(prob, Z + eps * v) - oracle.F(prob, Z - eps * v)) / (2 * eps), rtol=1e-6, atol=1e-8)


# ------------------------------------------------------------------------------------------------
#  GPU: the HIP path on the same lists
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nq,order,free_time,integrator", [(1, 12, True, "pade"), (1, 12, False, "pade"), (2, 4, True, "pade"),
                                                            (3, 4, True, "pade"), (3, 12, True, "pade"), (2, 4, True, "exponential"),
                                                            (4, 4, True, "pade")])
def test_bang_bang_problem_parity(qc, oracle, nq, order, free_time, integrator):
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI", 4: "QFT16"}[nq]
    T = 51 if nq == 1 else 9           # the reference's test: T = 51 (unitary_bang_bang_problem.jl:203)
    inp = qc.unitary_bang_bang_inputs(qc.multi_qubit_system(nq), qc.GATES[gate], T, pade_order=order, free_time=free_time,
                                      integrator=integrator, control_name="u")
    prob = problem_from_inputs(inp)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    assert dyn.dim == inp.traj.dims.states == prob.ddim
    F, J = dyn.F_dF(Z, fresh=True)
    close(F, oracle.F(prob, Z), "bang-bang F")
    close(J, oracle.dF(prob, Z), "bang-bang dF")
    close(dyn.F(Z, fresh=True), oracle.F(prob, Z), "bang-bang F alone")
    jr, jc = dyn.dF_structure
    r, c = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, r)
    np.testing.assert_array_equal(jc, c)
    # (the exponential integrator too, round 6: the templates solve it with the Hessian on, unitary_bang_bang_problem.jl:164-174)
    rng = np.random.default_rng(2)
    for mu in (np.ones(prob.n_rows), rng.standard_normal(prob.n_rows)):
        close(dyn.mu_d2F(Z, mu, fresh=True), oracle.mu_d2F(prob, Z, mu), "bang-bang hessian", atol=1e-11)
    hr, hc = dyn.mu_d2F_structure
    orr, oc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    dyn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("free_time,orders,exp_second", [(False, (4, 4), False), (True, (4, 4), False), (False, (12, 6), False),
                                                         (False, (4, 4), True)])
def test_direct_sum_problem_parity(qc, oracle, free_time, orders, exp_second):
    p1, p2 = direct_sum_members(qc, free_time, T=11, orders=orders, exp_second=exp_second)
    ds = qc.unitary_direct_sum_inputs([p1, p2])
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    assert isinstance(dyn, qc.ComposedQuantumDynamics) and len(dyn._parts) == 2
    ref = composed_oracle(ds)
    Z = ds.traj.datavec
    assert dyn.dim == ref.rows == ds.traj.dims.states
    F, J = dyn.F_dF(Z, fresh=True)
    close(F, ref.F(Z), "direct sum F")
    close(J, ref.dF(Z), "direct sum dF")
    jr, jc = dyn.dF_structure
    rr, rc = ref.structure()
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    # the members through their own handles, on their own trajectories: the same numbers, bit for bit
    T = ds.traj.T
    own = []
    for p in (p1, p2):
        d1 = qc.QuantumDynamics(p.integrators, p.traj)
        own.append(d1.F(p.traj.datavec, fresh=True).reshape(T - 1, -1))
        d1.close()
    np.testing.assert_array_equal(F, np.concatenate(own, axis=1).reshape(-1))
    rng = np.random.default_rng(4)            # (a Pade and an exponential member: both have their Hessian, round 6)
    mu = rng.standard_normal(dyn.dims.n_rows)
    close(dyn.mu_d2F(Z, mu, fresh=True), ref.mu_d2F(Z, mu), "direct sum hessian", atol=1e-11)
    hr, hc = dyn.mu_d2F_structure
    orr, oc = ref.hess_structure()
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    dyn.close()


@pytest.mark.gpu
def test_direct_sum_of_three_through_the_evaluator(qc, oracle):
    """`UnitaryDirectSumProblem([prob1, prob2, prob1], ...)` (unitary_direct_sum_problem.jl:254) behind the MOI-shaped evaluator."""
    p1, p2 = direct_sum_members(qc, False, T=8)
    ds = qc.unitary_direct_sum_inputs([p1, p2, p1])
    assert ds.traj.names[-4:] == ("Ũ⃗3", "a3", "da3", "dda3")
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    ref = composed_oracle(ds)
    ev = qc.QuantumControlEvaluator(dyn, [])
    Z = ds.traj.datavec
    g = np.zeros(ev.n_constraints)
    ev.eval_constraint(g, Z)
    close(g, ref.F(Z), "evaluator constraint")
    vals = np.zeros(ev.jac_nnz)
    ev.eval_constraint_jacobian(vals, Z)
    close(vals, ref.dF(Z), "evaluator jacobian")
    dyn.close()


@pytest.mark.gpu
def test_list_entry_points_error_behaviour_and_new_x(qc, oracle):
    """`qc_eval_*_list` through ctypes: refusals with messages (never a crash), Ipopt's new_x = false on the list's first handle,
    and the upload count a binding compares before it elides."""
    import ctypes as C
    L = qc._lib
    p1, p2 = direct_sum_members(qc, False, T=6)
    ds = qc.unitary_direct_sum_inputs([p1, p2])
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    ref = composed_oracle(ds)
    hs, n = dyn._handles, len(dyn._parts)
    Z = ds.traj.datavec
    F = np.zeros(int(dyn.dims.F_len))
    J = np.zeros(int(dyn.dims.jac_nnz))
    assert L.lib.qc_eval_F_jac_list(hs, n, L.dptr(Z), L.dptr(F), L.dptr(J)) == L.QC_OK
    close(F, ref.F(Z), "list F")
    close(J, ref.dF(Z), "list dF")
    # the first handle leading a shorter list (its staging blocks are re-made), then the full list again
    F1 = np.full_like(F, 7.0)
    J1 = np.full_like(J, 7.0)
    assert L.lib.qc_eval_F_jac_list(hs, 1, L.dptr(Z), L.dptr(F1), L.dptr(J1)) == L.QC_OK
    own_rows, own_vals = int(dyn._parts[0][3].ddim), int(dyn._parts[0][3].jac_nnz_interval)
    np.testing.assert_array_equal(F1.reshape(5, -1)[:, :own_rows], F.reshape(5, -1)[:, :own_rows])
    np.testing.assert_array_equal(J1.reshape(5, -1)[:, :own_vals], J.reshape(5, -1)[:, :own_vals])
    assert not F1.reshape(5, -1)[:, own_rows:].any()            # rows of integrators outside the list: delivered as 0
    F2, J2 = np.zeros_like(F), np.zeros_like(J)
    assert L.lib.qc_eval_F_jac_list(hs, n, L.dptr(Z), L.dptr(F2), L.dptr(J2)) == L.QC_OK
    np.testing.assert_array_equal(F2, F)
    np.testing.assert_array_equal(J2, J)
    # single-handle host entry points refuse a composed handle, and say where to go
    assert L.lib.qc_eval_F(dyn._parts[0][2], L.dptr(Z), L.dptr(F)) == L.QC_ERR_UNSUPPORTED
    assert b"composed" in L.lib.qc_last_error(dyn._parts[0][2])
    # refusals
    assert L.lib.qc_eval_F_list(None, 0, L.dptr(Z), L.dptr(F)) == L.QC_ERR_INVALID
    assert L.lib.qc_eval_F_list(hs, n, None, L.dptr(F)) == L.QC_ERR_INVALID
    assert L.lib.qc_eval_F_list(hs, n, L.dptr(Z), None) == L.QC_ERR_INVALID
    assert b"NULL buffer" in L.lib.qc_last_error(dyn._parts[0][2])
    other = qc.QuantumDynamics(p1.integrators, p1.traj)          # a handle over another trajectory
    mixed = (C.c_void_p * 2)(dyn._parts[0][2], other._h)
    assert L.lib.qc_eval_F_list(mixed, 2, L.dptr(Z), L.dptr(F)) == L.QC_ERR_INVALID
    assert b"do not describe one problem" in L.lib.qc_last_error(dyn._parts[0][2])
    other.close()
    # new_x = false: the knots of the last call are reused, Z is not read (a poisoned vector gives the same Jacobian)
    g0 = dyn.knot_generation()
    dyn.F(Z)
    assert dyn.knot_generation() == g0 + 1
    dyn.set_new_x(False)
    J2 = dyn.dF(np.full_like(Z, np.nan), fresh=True)
    assert dyn.knot_generation() == g0 + 1
    np.testing.assert_array_equal(J2, J)
    mu = np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))
    H2 = dyn.mu_d2F(np.full_like(Z, np.nan), mu, fresh=True)
    dyn.set_new_x(True)
    close(H2, ref.mu_d2F(Z, mu), "list hessian at the device's knots", atol=1e-11)
    np.testing.assert_array_equal(dyn.mu_d2F(Z, mu, fresh=True), H2)
    dyn.close()
    # a list with a Pade and an exponential member: both have a Hessian of the Lagrangian (round 6)
    p1, p2 = direct_sum_members(qc, False, T=6, exp_second=True)
    ds = qc.unitary_direct_sum_inputs([p1, p2])
    ref = composed_oracle(ds)
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    Zm = ds.traj.datavec
    mu = np.random.default_rng(1).standard_normal(int(dyn.dims.n_rows))
    close(dyn.mu_d2F(Zm, mu), ref.mu_d2F(Zm, mu), "list hessian, Pade + exponential members", atol=1e-11)
    hr, hc = dyn.mu_d2F_structure
    orr, oc = ref.hess_structure()
    np.testing.assert_array_equal(hr, orr)
    np.testing.assert_array_equal(hc, oc)
    dyn.close()


@pytest.mark.gpu
def test_rollouts_of_a_list_are_per_member(qc, oracle):
    """`unitary_rollout_fidelity(prob.trajectory, sys_k)` of a sampling problem (unitary_sampling_problem.jl:187-193) and the members of a
    direct sum: the rollout of the list's k-th state integrator uses ITS system and ITS controls."""
    rng = np.random.default_rng(8)
    base = qc.multi_qubit_system(2)
    systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.3 * k), base.H_drives) for k in range(3)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["CNOT"], 21)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    init = qc.operator_to_iso_vec(np.eye(4, dtype=complex))
    groups = qc.split_groups(inp.integrators)
    from types import SimpleNamespace
    for k in range(3):
        prob = problem_from_inputs(SimpleNamespace(integrators=groups[k], traj=inp.traj))
        np.testing.assert_allclose(dyn.rollout(Z, init, part=k), oracle.rollout(prob, Z, init), rtol=1e-10, atol=1e-11)
    assert not np.allclose(dyn.rollout(Z, init, part=0), dyn.rollout(Z, init, part=2))
    dyn.close()
    p1, p2 = direct_sum_members(qc, False, T=9)
    ds = qc.unitary_direct_sum_inputs([p1, p2])
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    for k, p in enumerate((p1, p2)):
        n = p.system.levels
        own = qc.QuantumDynamics(p.integrators, p.traj)
        i0 = qc.operator_to_iso_vec(np.eye(n, dtype=complex))
        np.testing.assert_array_equal(dyn.rollout(ds.traj.datavec, i0, part=k), own.rollout(p.traj.datavec, i0))
        own.close()
    dyn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sampling3", "directsum", "sampling_cfg5", "sampling_order6"])
def test_integrator_lists_over_several_devices_are_bit_identical(qc, oracle, kind):
    """The integrator lists of the sampling / direct-sum templates on a multi-device evaluator (missing 3): every member
    created over the same device list (qc_create_multi on a composed descriptor), the "_list" entry points evaluating shard by shard.
    devices = [0, 0, 0] on the one-GPU box: the same arrays as one device, bit for bit, whatever kernels serve the members
    (reference unitary_sampling_problem.jl:134-155, unitary_direct_sum_problem.jl:127-130)."""
    rng = np.random.default_rng(3)
    if kind == "sampling3":          # three 3-qubit systems, shared controls: the batched 2N = 16 launch on every shard
        base = qc.multi_qubit_system(3)
        systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.1 * k), base.H_drives) for k in range(3)]
        inp = qc.unitary_sampling_inputs(systems, qc.GATES["TOFFOLI"], 41)
    elif kind == "sampling_cfg5":    # two 4-qubit systems: the sparse-drive 2N = 32 kernels, one launch per member
        base = qc.multi_qubit_system(4)
        systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.2 * k), base.H_drives) for k in range(2)]
        inp = qc.unitary_sampling_inputs(systems, qc.GATES["QFT16"], 11)
    elif kind == "sampling_order6":  # any-order kernels
        base = qc.multi_qubit_system(2)
        systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.3 * k), base.H_drives) for k in range(2)]
        inp = qc.unitary_sampling_inputs(systems, qc.GATES["CNOT"], 14, pade_order=6)
    else:
        p1, p2 = direct_sum_members(qc, True, T=23)
        inp = qc.unitary_direct_sum_inputs([p1, p2])
    Z = inp.traj.datavec
    one = qc.QuantumDynamics(inp.integrators, inp.traj)
    many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0, 0, 0])
    assert isinstance(many, qc.ComposedQuantumDynamics) and many.n_shards == 3
    n_int = inp.traj.T - 1
    chunk = -(-n_int // 3)
    assert [many.shard_info(i)[1:] for i in range(3)] == [(min(i * chunk, n_int), min((i + 1) * chunk, n_int)) for i in range(3)]
    for a, b in zip(one.dF_structure + one.mu_d2F_structure, many.dF_structure + many.mu_d2F_structure):
        np.testing.assert_array_equal(a, b)
    mu = rng.standard_normal(int(one.dims.n_rows))
    F1, J1 = one.F_dF(Z, fresh=True)
    H1 = one.mu_d2F(Z, mu, fresh=True)
    ref = composed_oracle(inp)
    close(F1, ref.F(Z), kind + " F")
    close(J1, ref.dF(Z), kind + " dF")
    close(H1, ref.mu_d2F(Z, mu), kind + " mu_d2F", atol=1e-11)
    for rep in range(2):             # (the second round runs on warm staging: pinned ring blocks re-armed, parameter blocks cached)
        Fm, Jm = many.F_dF(Z, fresh=True)
        np.testing.assert_array_equal(Fm, F1)
        np.testing.assert_array_equal(Jm, J1)
        np.testing.assert_array_equal(many.mu_d2F(Z, mu, fresh=True), H1)
        np.testing.assert_array_equal(many.F(Z, fresh=True), F1)
        np.testing.assert_array_equal(many.dF(Z, fresh=True), J1)
    # Ipopt's new_x = false on the sharded list: every shard reuses the knots it has
    g0 = many.knot_generation()
    many.F(Z)
    assert many.knot_generation() == g0 + 1
    many.set_new_x(False)
    np.testing.assert_array_equal(many.dF(np.full_like(Z, np.nan), fresh=True), J1)
    np.testing.assert_array_equal(many.mu_d2F(np.full_like(Z, np.nan), mu, fresh=True), H1)
    assert many.knot_generation() == g0 + 1
    many.set_new_x(True)
    # mixing single- and multi-device members is refused with a message
    L = qc._lib
    import ctypes as C
    mixed = (C.c_void_p * 2)(many._parts[0][2], one._parts[1][2])
    assert L.lib.qc_eval_F_list(mixed, 2, L.dptr(Z), L.dptr(np.zeros(int(one.dims.F_len)))) == L.QC_ERR_INVALID
    with pytest.raises(ValueError):
        import torch
        many.F_dF_device(torch.from_numpy(Z).cuda(), None, torch.empty(int(one.dims.jac_nnz), dtype=torch.float64, device="cuda"))
    many.close()
    one.close()


@pytest.mark.gpu
def test_a_shorter_list_never_sees_the_longer_lists_values(qc, oracle):
    """Rows and values that no handle of a list owns are delivered as 0 -- also when the handle that leads the list led ANOTHER list just
    before (the zeroed-once device blocks change hands: ADVICE round 4), on the plain-copy path (F alone, mu_d2F) as on the watched one."""
    import ctypes as C
    L = qc._lib
    p1, p2 = direct_sum_members(qc, True, T=9)
    ds = qc.unitary_direct_sum_inputs([p1, p2])
    dyn = qc.QuantumDynamics(ds.integrators, ds.traj)
    hs, n = dyn._handles, len(dyn._parts)
    Z = ds.traj.datavec
    n_int = ds.traj.T - 1
    F, J = dyn.F_dF(Z, fresh=True)
    mu = np.random.default_rng(1).standard_normal(int(dyn.dims.n_rows))
    H = dyn.mu_d2F(Z, mu, fresh=True)
    Fa = dyn.F(Z, fresh=True)
    np.testing.assert_array_equal(Fa, F)
    own_rows, own_vals, own_h = (int(dyn._parts[0][3].ddim), int(dyn._parts[0][3].jac_nnz_interval), int(dyn._parts[0][3].hess_nnz_interval))
    for rep in range(2):
        F1, J1, H1 = np.full_like(F, 7.0), np.full_like(J, 7.0), np.full_like(H, 7.0)
        assert L.lib.qc_eval_F_list(hs, 1, L.dptr(Z), L.dptr(F1)) == L.QC_OK                    # plain-copy path
        assert L.lib.qc_eval_hess_list(hs, 1, L.dptr(Z), L.dptr(mu), L.dptr(H1)) == L.QC_OK
        assert L.lib.qc_eval_jac_list(hs, 1, L.dptr(Z), L.dptr(J1)) == L.QC_OK                  # watched path
        for what, got, full, own in (("F", F1, F, own_rows), ("dF", J1, J, own_vals), ("mu_d2F", H1, H, own_h)):
            g2, f2 = got.reshape(n_int, -1), full.reshape(n_int, -1)
            np.testing.assert_array_equal(g2[:, :own], f2[:, :own], err_msg=what)
            # what the list does not own: zeros (rows; values on the plain-copy path) or left as the caller had it (values on the
            # watched path, which writes the members' segments only) -- never the longer list's numbers
            rest = g2[:, own:]
            assert not rest.any() or (what == "dF" and (rest == 7.0).all()), what
        F2, J2 = np.zeros_like(F), np.zeros_like(J)                                              # ... and the whole list again
        assert L.lib.qc_eval_F_jac_list(hs, n, L.dptr(Z), L.dptr(F2), L.dptr(J2)) == L.QC_OK
        np.testing.assert_array_equal(F2, F)
        np.testing.assert_array_equal(J2, J)
        np.testing.assert_array_equal(dyn.mu_d2F(Z, mu, fresh=True), H)
        np.testing.assert_array_equal(dyn.F(Z, fresh=True), F)
    dyn.close()


@pytest.mark.gpu
def test_own_call_after_leading_a_by_component_list_sees_structural_zeros(qc):
    """two QC_ROWS_BY_COMPONENT handles (C API only: every handle's rows at its state component's position inside
    Z.dims.states rows per interval) evaluated as a LIST leave the members' rows in the leader's zeroed-once device vector; the leader's
    OWN residual call afterwards must deliver structural zeros there, not the other member's rows of the list call."""
    import ctypes as C
    L = qc._lib
    s1 = qc.multi_qubit_system(1)
    s1b = qc.QuantumSystem(1.3 * s1.H_drift, s1.H_drives)
    inp = qc.unitary_sampling_inputs([s1, s1b], qc.GATES["H"], 9)
    groups = qc.split_groups(inp.integrators)
    assert len(groups) == 2
    hs = (C.c_void_p * 2)()
    keep = []
    for k, grp in enumerate(groups):
        d, ka = qc.make_desc(grp[:1], inp.traj, rows="by_component")     # the two unitary integrators alone: equal value blocks, as a list needs
        keep.append(ka)
        h = C.c_void_p()
        L.check(L.lib.qc_create(C.byref(d), C.byref(h)))
        hs[k] = h
    dims = L.qc_dims_t()
    L.check(L.lib.qc_dims(hs[0], C.byref(dims)), hs[0])
    rows = int(inp.traj.dims.states)
    n_int = inp.traj.T - 1
    assert dims.F_len == rows * n_int
    Z = inp.traj.datavec
    F_own_before = np.full(dims.F_len, 7.0)
    L.check(L.lib.qc_eval_F(hs[0], L.dptr(Z), L.dptr(F_own_before)), hs[0])
    F_list = np.full(dims.F_len, 7.0)
    assert L.lib.qc_eval_F_list(hs, 2, L.dptr(Z), L.dptr(F_list)) == L.QC_OK
    F_own_after = np.full(dims.F_len, 7.0)
    L.check(L.lib.qc_eval_F(hs[0], L.dptr(Z), L.dptr(F_own_after)), hs[0])
    np.testing.assert_array_equal(F_own_after, F_own_before)
    own = 8                                                       # the first system's iso-vec rows
    a, b = F_own_after.reshape(n_int, rows), F_list.reshape(n_int, rows)
    np.testing.assert_array_equal(a[:, :own], b[:, :own])
    assert not a[:, own:].any() and b[:, own:2 * own].any()       # the list wrote the second member's rows; the own call shows zeros there
    # ... and the list again behind the own call
    F_list2 = np.full(dims.F_len, 7.0)
    assert L.lib.qc_eval_F_list(hs, 2, L.dptr(Z), L.dptr(F_list2)) == L.QC_OK
    np.testing.assert_array_equal(F_list2, F_list)
    for k in range(2):
        L.lib.qc_destroy(hs[k])
