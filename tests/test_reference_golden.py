"""Reference OUTPUTS, when somebody has produced them: `julia/reconcile.jl` (run on a machine with Julia and the reference's
packages) evaluates QuantumCollocationCore 0.3's own `QuantumDynamics` on the reference's fixture and on BASELINE configs 1 - 2
and writes tests/golden/ref_*.json.  These tests pick the files up -- the oracle on the CPU, the HIP path on the GPU -- and
skip while they are absent (they cannot be produced in the build container: no Julia).  With the files committed, "parity
unpinned"  becomes a pinned statement.

Comparison: residuals entry by entry; Jacobian / Hessian values as COO sets -- {(row, col): summed value} -- because the
ORDER of the entries inside an interval is Core's own and need not be this library's; structures are 1-based in
the files.  Tolerance: north_star's 1e-10 relative."""
import glob
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FILES = sorted(f for f in glob.glob(os.path.join(GOLD, "ref_*.json")) if not f.endswith("_exponential.json"))
RTOL = 1e-10

pytestmark = pytest.mark.skipif(not FILES, reason="no tests/golden/ref_*.json: run julia/reconcile.jl where Julia and QuantumCollocationCore 0.3 exist")


def coo_sum(rows, cols, vals, one_based):
    """{(row, col): value} with duplicates summed (reference test/test_utils.jl:14-20), zero-valued entries dropped."""
    out = {}
    k = 1 if one_based else 0
    for r, c, v in zip(np.asarray(rows).tolist(), np.asarray(cols).tolist(), np.asarray(vals).tolist()):
        out[(r - k, c - k)] = out.get((r - k, c - k), 0.0) + v
    return out


def assert_coo_equal(ours, ref, what, scale):
    keys = set(ours) | set(ref)
    worst = max(abs(ours.get(k, 0.0) - ref.get(k, 0.0)) for k in keys)
    assert worst <= RTOL * max(1.0, scale), f"{what}: max |difference| {worst:.3e} over {len(keys)} positions (scale {scale:.3e})"
    # structure: every position the reference lists must be listed here too (this library may list more: dense blocks of B / F
    # where Core detects structural zeros), and extra positions must hold zeros -- implied by the value check above
    missing = [k for k in ref if k not in ours and ref[k] != 0.0]
    assert not missing, f"{what}: {len(missing)} non-zero reference positions are not in this library's structure, e.g. {missing[:3]}"


def problem_from_record(qc, rec):
    """(integrators, traj) of a reconcile.jl record, through the mirror constructors (reference call order
    unitary_smooth_pulse_problem.jl:163-179)."""
    N = int(rec["levels"])
    cplx = lambda re, im: (np.asarray(re, dtype=float) + 1j * np.asarray(im, dtype=float)).reshape(N, N, order="F")
    system = qc.QuantumSystem(cplx(rec["H_drift_re"], rec["H_drift_im"]),
                              [cplx(r, i) for r, i in zip(rec["H_drives_re"], rec["H_drives_im"])])
    T, dim = int(rec["T"]), int(rec["dim"])
    Z = np.asarray(rec["Z"], dtype=float)
    data = Z[:dim * T].reshape(dim, T, order="F")
    comps = {name: data[np.asarray(rec["components"][name]) - 1, :] for name in rec["names"]}      # 1-based rows, trajectory order
    row = 0
    for name in rec["names"]:                      # the mirror lays components out in `names` order: the file must agree
        idx = np.asarray(rec["components"][name]) - 1
        assert idx.tolist() == list(range(row, row + idx.size)), f"component {name} is not contiguous / in order"
        row += idx.size
    ts = rec["timestep"]
    traj = qc.NamedTrajectory(comps, controls=("dda",), timestep=ts if isinstance(ts, str) else float(ts))
    integ = [qc.UnitaryPadeIntegrator("Ũ⃗", "a", system, traj, order=int(rec["pade_order"])), qc.DerivativeIntegrator("a", "da", traj),
             qc.DerivativeIntegrator("da", "dda", traj)]
    return integ, traj, Z


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_against_reference_outputs(qc, oracle, path):
    """The CPU oracle against Core's numbers: this is what pins the oracle (and, through the GPU parity tests, the kernels)."""
    from oracle_bridge import problem_from_inputs
    rec = json.load(open(path))
    integ, traj, Z = problem_from_record(qc, rec)
    prob = problem_from_inputs(type("I", (), {"integrators": integ, "traj": traj})())
    prob.hess_align = 1
    F = oracle.F(prob, Z)
    Fr = np.asarray(rec["F"], dtype=float)
    assert F.size == Fr.size == int(rec["rows_declared"])
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=RTOL * max(1.0, np.abs(Fr).max()))
    jr, jc = oracle.jac_structure(prob)
    Jr = np.asarray(rec["dF"], dtype=float)
    assert_coo_equal(coo_sum(jr, jc, oracle.dF(prob, Z), False), coo_sum(rec["dF_rows"], rec["dF_cols"], Jr, True), "dF", np.abs(Jr).max())
    if "mu_d2F" in rec:
        mu = np.asarray(rec["mu"], dtype=float)
        hr, hc = oracle.hess_structure(prob)
        Hr = np.asarray(rec["mu_d2F"], dtype=float)
        assert_coo_equal(coo_sum(hr, hc, oracle.mu_d2F(prob, Z, mu), False), coo_sum(rec["mu_d2F_rows"], rec["mu_d2F_cols"], Hr, True), "mu_d2F",
                         np.abs(Hr).max())


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_path_against_reference_outputs(qc, path):
    """The HIP path, through the C ABI, directly against Core's numbers (hess_align = 1: exactly the structural entries)."""
    rec = json.load(open(path))
    integ, traj, Z = problem_from_record(qc, rec)
    dyn = qc.QuantumDynamics(integ, traj, hess_align=1)
    F, J = dyn.F_dF(Z)
    Fr, Jr = np.asarray(rec["F"], dtype=float), np.asarray(rec["dF"], dtype=float)
    np.testing.assert_allclose(F, Fr, rtol=RTOL, atol=RTOL * max(1.0, np.abs(Fr).max()))
    jr, jc = dyn.dF_structure
    assert_coo_equal(coo_sum(jr, jc, J, False), coo_sum(rec["dF_rows"], rec["dF_cols"], Jr, True), "dF", np.abs(Jr).max())
    if "mu_d2F" in rec:
        mu = np.asarray(rec["mu"], dtype=float)
        hr, hc = dyn.mu_d2F_structure
        Hr = np.asarray(rec["mu_d2F"], dtype=float)
        assert_coo_equal(coo_sum(hr, hc, dyn.mu_d2F(Z, mu), False), coo_sum(rec["mu_d2F_rows"], rec["mu_d2F_cols"], Hr, True), "mu_d2F", np.abs(Hr).max())
        # bit-exact sparsity structure, as north_star words it: the same SET of positions (modulo explicit zeros either side lists)
        ours = {(int(r), int(c)) for r, c in zip(hr, hc)}
        ref = {(int(r) - 1, int(c) - 1) for r, c in zip(rec["mu_d2F_rows"], rec["mu_d2F_cols"])}
        assert ref <= ours or ours <= ref, "Hessian structures are not nested"
    dyn.close()


def test_scalar_definitions_against_reference(qc, oracle):
    """The definitions INTEGRATION.md lists as unverifiable, from ref_fixture.json: regulariser weighting, fidelity form."""
    path = os.path.join(GOLD, "ref_fixture.json")
    if not os.path.exists(path):
        pytest.skip("no ref_fixture.json")
    rec = json.load(open(path))
    L = rec["regularizer_a_R1"]
    # the library's default must be the reference's definition
    assert abs(L - rec["regularizer_dt_scaled"]) <= 1e-12 * max(1.0, abs(L)), "QuadraticRegularizer is not dt-scaled: flip the bindings' default to QC_REG_PLAIN"
    fid, gfid, hfid = oracle.fidelity_value_grad_hess(np.asarray(rec["fidelity_state"]), np.asarray(rec["fidelity_goal"]))
    assert abs(fid - rec["fidelity"]) <= 1e-12, "iso_vec_unitary_fidelity is not |tr|/n: the objectives' default form must become 'abs2'"
