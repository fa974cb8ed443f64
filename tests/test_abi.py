"""CPU-side checks of the drop-in boundary: the library loads, exports exactly what include/qcolloc.h
declares, struct layouts agree between C and ctypes, the host-only entry points (layout, structure,
iso helpers) agree with the oracle, and errors surface as codes + messages.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle_bridge import problem_from_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qcolloc.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qc_[a-z_A-Z0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(qc):
    declared = header_functions()
    assert len(declared) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", qc._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (qc_\w+)", out))
    assert set(declared) <= exported, set(declared) - exported
    assert set(declared) == set(qc._lib.SYMBOLS), set(declared) ^ set(qc._lib.SYMBOLS)
    assert b"gfx950" in qc._lib.lib.qc_version()


def test_struct_layout_matches_c(qc, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "qcolloc.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(qc_desc), offsetof(qc_desc, G_drift), '
        'offsetof(qc_desc, t_begin), offsetof(qc_desc, deriv_dim), offsetof(qc_desc, dt_fixed), '
        'sizeof(qc_dims_t), offsetof(qc_dims_t, kernel), sizeof(qc_terms_desc), offsetof(qc_terms_desc, reg_baseline), '
        'offsetof(qc_terms_desc, device));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    d, dm, td = qc._lib.qc_desc, qc._lib.qc_dims_t, qc._lib.qc_terms_desc
    assert got == [C.sizeof(d), d.G_drift.offset, d.t_begin.offset, d.deriv_dim.offset, d.dt_fixed.offset,
                   C.sizeof(dm), dm.kernel.offset, C.sizeof(td), td.reg_baseline.offset, td.device.offset]


def test_iso_helpers_match_oracle(qc, oracle):
    rng = np.random.default_rng(0)
    for N in (1, 2, 3, 8):
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        np.testing.assert_array_equal(qc.operator_to_iso_vec(A), oracle.operator_to_iso_vec(A))
        np.testing.assert_array_equal(qc.iso_vec_to_operator(qc.operator_to_iso_vec(A)), A)
        H = (A + A.conj().T) / 2
        np.testing.assert_array_equal(qc.iso_generator(H), oracle.generator(H))
    for order in range(2, 21, 2):
        np.testing.assert_allclose(qc.pade_coefficients(order), oracle.pade_coeffs(order), rtol=1e-15)


@pytest.mark.parametrize("cfg,T", [(1, 50), (2, 200), (3, 40), (5, 4)])
def test_dims_and_structure_match_oracle(qc, oracle, cfg, T):
    inp = qc.config_inputs(cfg, T=T)
    desc, keep = qc.make_desc(inp.integrators, inp.traj)
    dims = qc.desc_dims(desc)
    prob = problem_from_inputs(inp)
    assert dims.n_rows == prob.n_rows and dims.n_cols == prob.n_vars and dims.ddim == prob.ddim
    assert dims.jac_nnz_interval == oracle.jac_nnz_interval(prob)
    assert dims.hess_nnz_interval == oracle.hess_nnz_interval(prob)
    assert dims.n_intervals == T - 1
    jr, jc, hr, hc = qc.desc_structures(desc)
    orr, oc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, orr)      # bit-exact, including order
    np.testing.assert_array_equal(jc, oc)
    ohr, ohc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, ohr)
    np.testing.assert_array_equal(hc, ohc)
    jr1, jc1, _, _ = qc.desc_structures(desc, one_based=True)
    np.testing.assert_array_equal(jr1, jr + 1)
    np.testing.assert_array_equal(jc1, jc + 1)


@pytest.mark.parametrize("cfg", [3, 5])
def test_hessian_matrix_blocks_sit_on_whole_lines_and_the_scalar_entries_are_one_run(qc, cfg):
    """The per-interval value layout of mu_d2F: the four kinds of matrix blocks first -- each a multiple of 16
    entries when 2N x N is, so that with hess_align = 16 every block store covers whole 128-byte lines --, then every scalar entry
    ((a,a), (a,dt), (dt,dt), (dx,dt)) as ONE contiguous run, which the one-call kernel stores in one piece."""
    inp = qc.config_inputs(cfg, T=4)
    traj = inp.traj
    desc, keep = qc.make_desc(inp.integrators, traj, hess_align=16)
    dims = qc.desc_dims(desc)
    _, _, hr, hc = qc.desc_structures(desc)
    k = dims.hess_nnz_interval
    assert k % 16 == 0
    r, c = hr[:k], hc[:k]                      # interval 0: local indices over [z_0 ; z_1]
    zd = traj.dim
    U = set(range(traj.components["Ũ⃗"].start, traj.components["Ũ⃗"].stop)) if hasattr(traj, "components") else None
    if U is None:
        pytest.skip("trajectory without a component table")
    s = len(U)
    while k > 1 and (r[k - 1], c[k - 1]) == (r[0], c[0]):      # the alignment padding: duplicates of the first entry at the very end
        k -= 1
    kinds = np.array([(r[i] % zd in U) != (c[i] % zd in U) for i in range(k)])     # a matrix-block entry pairs a state entry with a drive or dt
    n_matrix = int(kinds.sum())
    assert n_matrix % s == 0 and s % 16 == 0
    assert kinds[:n_matrix].all() and not kinds[n_matrix:].any()        # matrix blocks first, the scalar run behind them
    # every block of s entries has one fixed partner index (a drive or dt) and runs over the whole state block
    for b0 in range(0, n_matrix, s):
        rows, cols = r[b0:b0 + s], c[b0:b0 + s]
        state_side = rows if len(set(rows)) == s else cols
        other = cols if state_side is rows else rows
        assert len(set(other)) == 1 and sorted(x % zd for x in state_side) == sorted(U)


def test_structure_of_a_shard(qc, oracle):
    inp = qc.config_inputs(2, T=20)
    desc, keep = qc.make_desc(inp.integrators, inp.traj, t_range=(7, 13))
    dims = qc.desc_dims(desc)
    assert dims.n_intervals == 6 and dims.Z_len == inp.traj.dim * 20
    jr, jc, hr, hc = qc.desc_structures(desc)
    prob = problem_from_inputs(inp)
    orr, oc = oracle.jac_structure(prob, t_begin=7, t_end=13)
    np.testing.assert_array_equal(jr, orr)
    np.testing.assert_array_equal(jc, oc)
    ohr, ohc = oracle.hess_structure(prob, t_begin=7, t_end=13)
    np.testing.assert_array_equal(hr, ohr)
    np.testing.assert_array_equal(hc, ohc)


def test_exponential_and_fixed_time_structure(qc, oracle):
    sys_ = qc.multi_qubit_system(1)
    inp = qc.unitary_smooth_pulse_inputs(sys_, qc.GATES["H"], 10, free_time=False, integrator="exponential")
    desc, keep = qc.make_desc(inp.integrators, inp.traj)
    prob = problem_from_inputs(inp)
    jr, jc, hr, hc = qc.desc_structures(desc)
    orr, oc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, orr)
    np.testing.assert_array_equal(jc, oc)
    # mu_d2F of the exponential integrator (fixed timestep: (U_t, a) and (a, a) only; nothing touches knot t+1)
    ohr, ohc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, ohr)
    np.testing.assert_array_equal(hc, ohc)
    s, m = prob.s, prob.m
    assert qc.desc_dims(desc).hess_nnz_interval == s * m + m * (m + 1) // 2


def test_invalid_descriptors_are_rejected_with_a_message(qc):
    inp = qc.config_inputs(1, T=5)
    lib = qc._lib.lib

    def rc_of(mut):
        desc, keep = qc.make_desc(inp.integrators, inp.traj)
        mut(desc)
        dims = qc._lib.qc_dims_t()
        rc = lib.qc_desc_dims(C.byref(desc), C.byref(dims))
        return rc, lib.qc_last_error(None).decode()

    for mut in (lambda d: setattr(d, "T", 1), lambda d: setattr(d, "pade_order", 5), lambda d: setattr(d, "off_U", 12),
                lambda d: setattr(d, "off_a", 3), lambda d: setattr(d, "integrator", 7), lambda d: setattr(d, "zdim", 9),
                lambda d: setattr(d, "t_end", 99), lambda d: setattr(d, "n_deriv", 9), lambda d: setattr(d, "N", 0)):
        rc, msg = rc_of(mut)
        assert rc == qc._lib.QC_ERR_INVALID and msg
    assert lib.qc_desc_dims(None, None) == qc._lib.QC_ERR_INVALID


@pytest.mark.skipif(torch.cuda.is_available(), reason="this box has a GPU")
def test_no_cpu_fallback_without_a_gpu(qc):
    """The product path must fail loudly when there is no device; it never routes through the oracle."""
    inp = qc.config_inputs(1, T=5)
    with pytest.raises(qc.QCollocError) as e:
        qc.QuantumDynamics(inp.integrators, inp.traj)
    assert e.value.code == qc._lib.QC_ERR_NO_DEVICE


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "quantumcollocation.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"qc_oracle", r"libqc_oracle", r"oracle/", r"load_oracle", r"qco_"):
                    assert not re.search(pat, txt), f"{fn} references the oracle ({pat})"
                if re.search(r"dlopen|CDLL", txt):
                    # run-time loading exists in two places only: the ctypes binding loads libqcolloc_hip.so, and the library
                    # loads RCCL on first use of the in-library all-gather; every shared-object name in such a file must be one of those
                    assert fn in ("_lib.py", "qc_host_eval.cpp"), f"{fn} loads shared objects at run time"
                    for so in re.findall(r'"([^"\s]*\.so[.\d]*)"', txt):
                        assert "rccl" in so or "libqcolloc_hip" in so, f"{fn} may load {so}"


def test_ket_integrators_build_a_descriptor(qc, oracle):
    sys_ = qc.multi_qubit_system(2)
    psi0 = [np.eye(4)[:, 0], np.eye(4)[:, 1]]
    psi1 = [np.eye(4)[:, 1], np.eye(4)[:, 0]]
    inp = qc.quantum_state_smooth_pulse_inputs(sys_, psi0, psi1, 9)
    desc, keep = qc.make_desc(inp.integrators, inp.traj)
    assert desc.state_cols == 2 and desc.N == 4 and desc.n_deriv == 2
    dims = qc.desc_dims(desc)
    prob = problem_from_inputs(inp)
    assert prob.ncol == 2 and dims.ddim == prob.ddim == 2 * 8 + 8
    jr, jc, hr, hc = qc.desc_structures(desc)
    rr, rc = oracle.jac_structure(prob)
    np.testing.assert_array_equal(jr, rr)
    np.testing.assert_array_equal(jc, rc)
    ohr, ohc = oracle.hess_structure(prob)
    np.testing.assert_array_equal(hr, ohr)
    np.testing.assert_array_equal(hc, ohc)


def test_sampling_problem_groups_and_placement(qc, oracle):
    """Several unitary integrators: one descriptor per integrator, rows/values placed inside shared blocks."""
    base = qc.multi_qubit_system(1)
    systems = [qc.QuantumSystem(base.H_drift * f, base.H_drives) for f in (0.9, 1.0, 1.1)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["H"], 7)
    groups = qc.split_groups(inp.integrators)
    assert [len(g) for g in groups] == [1, 1, 3]
    assert inp.traj.names[:3] == ("Ũ⃗_system_1", "Ũ⃗_system_2", "Ũ⃗_system_3") and inp.traj.dims.states == 3 * 8 + 4
    own = [qc.desc_dims(qc.make_desc(g, inp.traj)[0]) for g in groups]
    rows = sum(x.ddim for x in own)
    jac = sum(x.jac_nnz_interval for x in own)
    assert rows == 28
    desc, keep = qc.make_desc(groups[1], inp.traj, placement=dict(rows_per_interval=rows, row_offset=8, jac_per_interval=jac,
                                                                   jac_offset=own[0].jac_nnz_interval))
    d = qc.desc_dims(desc)
    assert d.n_rows == rows * 6 and d.ddim == 8
    jr, jc, _, _ = qc.desc_structures(desc)
    assert jr.min() == 8 and jr.max() == 5 * rows + 15          # rows 8..15 of every interval block
    assert jc.min() == inp.traj.offset("Ũ⃗_system_2")
    # an offset that does not fit is rejected
    bad, keep2 = qc.make_desc(groups[1], inp.traj, placement=dict(rows_per_interval=10, row_offset=8))
    dims = qc._lib.qc_dims_t()
    assert qc._lib.lib.qc_desc_dims(C.byref(bad), C.byref(dims)) == qc._lib.QC_ERR_INVALID


def test_random_descriptors_structure_properties(qc, oracle):
    """Host-only property sweep over random layouts: the structure vectors are duplicate-free, in range, upper-triangular
    (Hessian), equal to the oracle's bit for bit, shard-wise concatenable, and their 1-based form is the 0-based one plus 1."""
    from oracle_bridge import random_problem
    rng = np.random.default_rng(2025)
    L = qc._lib
    for trial in range(40):
        N = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 20]))
        m = int(rng.integers(0, 5))
        T = int(rng.integers(2, 7))
        order = int(rng.choice([2, 4, 6]))
        free_time = bool(rng.integers(0, 2))
        integ = oracle.EXPONENTIAL if rng.random() < 0.25 else oracle.PADE
        ncol = int(rng.integers(1, N + 1)) if rng.random() < 0.3 else 0
        prob, _ = random_problem(oracle, N=N, m=max(m, 1), T=T, order=order, free_time=free_time, integrator=integ,
                                 seed=int(rng.integers(1 << 30)), ncol=ncol, layout=str(rng.choice(["standard", "shuffled"])))
        if m == 0:
            prob.m = 0
            prob.G_drives = prob.G_drives[:0]
            prob.derivs = []
        d = L.qc_desc()
        d.N, d.m, d.T, d.zdim, d.global_dim = prob.N, prob.m, prob.T, prob.zdim, 0
        d.off_U, d.off_a, d.off_dt, d.dt_fixed = prob.off_U, prob.off_a, prob.off_dt, prob.dt_fixed
        d.integrator, d.pade_order, d.n_deriv, d.state_cols = prob.integrator, prob.order, len(prob.derivs), ncol
        for i, dv in enumerate(prob.derivs):
            d.deriv_x_off[i], d.deriv_dx_off[i], d.deriv_dim[i] = dv.x_off, dv.dx_off, dv.dim
        d.hess_align = int(rng.choice([0, 1, 8, 16, 24]))          # 0 = the library default: exactly the structural entries (ABI 0.5)
        prob.hess_align = d.hess_align or 1
        tag = f"trial {trial}: N={N} m={m} T={T} order={order} ft={free_time} integ={integ} ncol={ncol} align={d.hess_align}"
        dims = qc.desc_dims(d)
        assert dims.n_rows == prob.n_rows and dims.n_cols == prob.n_vars, tag
        jr, jc, hr, hc = qc.desc_structures(d)
        orr, oc = oracle.jac_structure(prob)
        np.testing.assert_array_equal(jr, orr, err_msg=tag)
        np.testing.assert_array_equal(jc, oc, err_msg=tag)
        assert jr.size == dims.jac_nnz and len(set(zip(jr.tolist(), jc.tolist()))) == jr.size, tag
        assert jr.min() >= 0 and jr.max() < dims.n_rows and jc.min() >= 0 and jc.max() < dims.n_cols, tag
        if True:        # both integrators have a Hessian structure (the exponential one's without entries at knot t+1)
            ohr, ohc = oracle.hess_structure(prob)
            np.testing.assert_array_equal(hr, ohr, err_msg=tag)
            np.testing.assert_array_equal(hc, ohc, err_msg=tag)
            if hr.size:
                # the alignment padding repeats an interval's first entry (explicit zeros); everything else is duplicate-free
                pad = oracle.hess_pad(prob) * (prob.T - 1)
                assert dims.hess_nnz_interval % prob.hess_align == 0 and dims.hess_nnz_interval - oracle.hess_pad(prob) == len(oracle.hess_structure_local(prob)), tag
                assert np.all(hr <= hc) and len(set(zip(hr.tolist(), hc.tolist()))) == hr.size - pad and hc.max() < dims.n_cols, tag
                if integ != oracle.PADE:
                    assert (hc.reshape(prob.T - 1, -1) < (np.arange(prob.T - 1)[:, None] + 1) * prob.zdim).all(), tag
        jr1, jc1, hr1, hc1 = qc.desc_structures(d, one_based=True)
        np.testing.assert_array_equal(jr1, jr + 1)
        np.testing.assert_array_equal(hc1, hc + 1)
        if T >= 4:      # two shards concatenate to the whole
            cut = int(rng.integers(1, T - 1))
            parts = []
            for tb, te in ((0, cut), (cut, T - 1)):
                d.t_begin, d.t_end = tb, te
                parts.append(qc.desc_structures(d))
            d.t_begin, d.t_end = 0, 0
            np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), jr, err_msg=tag)
            np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), jc, err_msg=tag)
            np.testing.assert_array_equal(np.concatenate([p[2] for p in parts]), hr, err_msg=tag)


def _build_c_example(tmp_path):
    exe = tmp_path / "c_abi_example"
    csrc = os.path.join(ROOT, "quantumcollocation.jl_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"),
                    "-o", str(exe), "-L", csrc, "-lqcolloc_hip", f"-Wl,-rpath,{csrc}", "-lm"], check=True)
    return exe


@pytest.mark.skipif(torch.cuda.is_available(), reason="this box has a GPU")
def test_c_example_compiles_as_c99_and_fails_loudly_without_a_device(qc, tmp_path):
    """include/qcolloc.h is plain C: the example builds with gcc -std=c99 -Werror, gets dims from the host-only entry points
    and stops at qc_create with QC_ERR_NO_DEVICE (no CPU path)."""
    exe = _build_c_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 1 and "rows 60 cols 90 jac_nnz 520 hess_nnz 290" in r.stdout   # 58 values per interval: exactly the structural entries (hess_align = 0)
    assert "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_example_matches_the_python_mirror(qc, tmp_path):
    exe = _build_c_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, check=True)
    m = re.search(r"checksums F (\S+) dF (\S+) mu_d2F (\S+) first entry \((\d+),(\d+)\)", r.stdout)
    assert m, r.stdout
    assert "pinned residuals: same values" in r.stdout, r.stdout                       # qc_host_alloc: residuals written in place by the kernel
    assert "ipopt order: same values" in r.stdout, r.stdout          # qc_set_new_x(h, 0): Jacobian / Hessian at the point of the last F call
    N, M, T = 2, 2, 6
    S = 2 * N * N
    zdim = S + 3 * M + 1
    Z = np.zeros((T, zdim))
    for t in range(T):
        th = 0.3 * t
        Z[t, :8] = [np.cos(th), 0, 0, -np.sin(th), 0, np.cos(th), -np.sin(th), 0]
        Z[t, S:S + 3 * M] = [0.1 * np.sin(1.0 + t + 0.7 * k) for k in range(3 * M)]
        Z[t, S + 3 * M] = 0.2
    comps = {"Ũ⃗": Z[:, :8].T, "a": Z[:, 8:10].T, "da": Z[:, 10:12].T, "dda": Z[:, 12:14].T, "Δt": Z[:, 14:15].T}
    traj = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    sys_ = qc.QuantumSystem(0.1 * qc.PAULIS["Z"], [qc.PAULIS["X"], qc.PAULIS["Y"]])
    dyn = qc.QuantumDynamics([qc.UnitaryPadeIntegrator("Ũ⃗", "a", sys_, traj), qc.DerivativeIntegrator("a", "da", traj),
                              qc.DerivativeIntegrator("da", "dda", traj)], traj)
    F, J = dyn.F_dF(traj.datavec)
    H = dyn.mu_d2F(traj.datavec, np.ones(int(dyn.dims.n_rows)))
    sF = float(np.sum(F * (1 + np.arange(F.size) % 7)))
    sJ = float(np.sum(J * (1 + np.arange(J.size) % 11)))
    sH = float(np.sum(H * (1 + np.arange(H.size) % 13)))
    for got, ref in zip(m.groups()[:3], (sF, sJ, sH)):
        assert abs(float(got) - ref) <= 1e-11 * max(1.0, abs(ref)), (got, ref)
    jr, jc = dyn.dF_structure
    assert (int(m.group(4)), int(m.group(5))) == (int(jr[0]) + 1, int(jc[0]) + 1)
    dyn.close()
    # the example's integrator list (two systems, shared controls) against the same list through the Python mirror
    ml = re.search(r"list checksums F (\S+) dF (\S+) mu_d2F (\S+) sizes (\d+) (\d+) (\d+)", r.stdout)
    assert ml, r.stdout
    Z2 = np.zeros((T, 2 * S + 3 * M + 1))
    for t in range(T):
        for k in range(2):
            th = (0.3 + 0.05 * k) * t
            Z2[t, k * S:(k + 1) * S] = [np.cos(th), 0, 0, -np.sin(th), 0, np.cos(th), -np.sin(th), 0]
        Z2[t, 2 * S:2 * S + 3 * M] = [0.1 * np.sin(1.0 + t + 0.7 * k) for k in range(3 * M)]
        Z2[t, 2 * S + 3 * M] = 0.2
    comps = {"Ũ⃗_system_1": Z2[:, :8].T, "Ũ⃗_system_2": Z2[:, 8:16].T, "a": Z2[:, 16:18].T, "da": Z2[:, 18:20].T, "dda": Z2[:, 20:22].T,
             "Δt": Z2[:, 22:23].T}
    traj2 = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    sys_b = qc.QuantumSystem(-0.1 * qc.PAULIS["Z"], [qc.PAULIS["X"], qc.PAULIS["Y"]])
    lst = qc.QuantumDynamics([qc.UnitaryPadeIntegrator("Ũ⃗_system_1", "a", sys_, traj2), qc.UnitaryPadeIntegrator("Ũ⃗_system_2", "a", sys_b, traj2),
                              qc.DerivativeIntegrator("a", "da", traj2), qc.DerivativeIntegrator("da", "dda", traj2)], traj2, hess_align=1)
    F2, J2 = lst.F_dF(traj2.datavec)
    H2 = lst.mu_d2F(traj2.datavec, np.ones(int(lst.dims.n_rows)))
    assert (int(ml.group(4)), int(ml.group(5)), int(ml.group(6))) == (F2.size, J2.size, H2.size)
    for got, arr, mod in zip(ml.groups()[:3], (F2, J2, H2), (7, 11, 13)):
        ref = float(np.sum(arr * (1 + np.arange(arr.size) % mod)))
        assert abs(float(got) - ref) <= 1e-11 * max(1.0, abs(ref)), (got, ref)
    lst.close()


def test_descriptor_fuzz_never_crashes(qc):
    """3000 descriptors with random (often nonsensical) fields through the host-only entry points: every call returns QC_OK or
    QC_ERR_INVALID with a message; accepted descriptors produce structures of exactly the announced size and range.
    (tests/run_asan_host.sh runs this file under AddressSanitizer.)"""
    L = qc._lib
    rng = np.random.default_rng(31337)
    ok = 0
    for trial in range(3000):
        # a valid smooth-pulse layout [U, a, da, dda, dt] ...
        d = L.qc_desc()
        N = int(rng.choice([1, 2, 3, 4, 8]))
        m = int(rng.choice([0, 1, 2, 6]))
        T = int(rng.choice([2, 3, 7]))
        s = 2 * N * N
        zdim = s + 3 * m + 1
        d.N, d.m, d.T, d.zdim, d.global_dim = N, m, T, zdim, int(rng.choice([0, 3]))
        d.off_U, d.off_a, d.off_dt, d.dt_fixed = 0, s, s + 3 * m, 0.0
        d.integrator, d.pade_order = int(rng.choice([0, 0, 1])), int(rng.choice([2, 4, 6]))
        d.n_deriv = 2 if m else 0
        if m:
            d.deriv_x_off[0], d.deriv_dx_off[0], d.deriv_dim[0] = s, s + m, m
            d.deriv_x_off[1], d.deriv_dx_off[1], d.deriv_dim[1] = s + m, s + 2 * m, m
        # ... with zero to three fields corrupted
        for _ in range(int(rng.integers(0, 4))):
            f = str(rng.choice(["N", "m", "T", "zdim", "global_dim", "off_U", "off_a", "off_dt", "integrator", "pade_order", "n_deriv", "deriv",
                                "state_cols", "t_range", "placement", "dt_fixed"]))
            if f == "deriv":
                i = int(rng.integers(0, 8))
                d.deriv_x_off[i] = int(rng.integers(-2, zdim + 3))
                d.deriv_dx_off[i] = int(rng.integers(-2, zdim + 3))
                d.deriv_dim[i] = int(rng.choice([0, 1, max(m, 1), zdim, -1]))
            elif f == "t_range":
                d.t_begin, d.t_end = int(rng.choice([0, 1, -1, T])), int(rng.choice([0, T - 1, T + 3, -2]))
            elif f == "placement":
                d.rows_per_interval, d.row_offset = int(rng.integers(-5, 60)), int(rng.integers(-5, 60))
                d.jac_per_interval, d.jac_offset = int(rng.integers(-5, 600)), int(rng.integers(-5, 600))
            elif f == "dt_fixed":
                d.dt_fixed = float(rng.choice([0.2, -1.0, np.nan]))
                d.off_dt = -1
            else:
                setattr(d, f, int(rng.choice([0, 1, 2, 3, 5, 8, 9, 17, 33, 70, -1, -4, zdim, zdim - 1, s])))
        dims = L.qc_dims_t()
        rc = L.lib.qc_desc_dims(C.byref(d), C.byref(dims))
        assert rc in (L.QC_OK, L.QC_ERR_INVALID), rc
        if rc != L.QC_OK:
            assert L.lib.qc_last_error(None)
            continue
        ok += 1
        assert dims.n_rows >= 0 and dims.jac_nnz >= 0 and dims.hess_nnz >= 0
        if dims.jac_nnz > 2_000_000:
            continue
        jr = np.full(dims.jac_nnz + 8, -77, dtype=np.int64)
        jc = np.full(dims.jac_nnz + 8, -77, dtype=np.int64)
        assert L.lib.qc_desc_jac_structure(C.byref(d), L.iptr(jr), L.iptr(jc), 0) == L.QC_OK
        assert np.all(jr[dims.jac_nnz:] == -77) and np.all(jc[dims.jac_nnz:] == -77)          # nothing written past the end
        if dims.jac_nnz:
            assert jr[:dims.jac_nnz].min() >= 0 and jr[:dims.jac_nnz].max() < dims.n_rows
            assert jc[:dims.jac_nnz].min() >= 0 and jc[:dims.jac_nnz].max() < dims.n_cols
        hr = np.full(dims.hess_nnz + 8, -77, dtype=np.int64)
        hc = np.full(dims.hess_nnz + 8, -77, dtype=np.int64)
        assert L.lib.qc_desc_hess_structure(C.byref(d), L.iptr(hr), L.iptr(hc), 0) == L.QC_OK
        assert np.all(hr[dims.hess_nnz:] == -77)
        if dims.hess_nnz:
            assert hr[:dims.hess_nnz].min() >= 0 and hc[:dims.hess_nnz].max() < dims.n_cols and np.all(hr[:dims.hess_nnz] <= hc[:dims.hess_nnz])
    assert ok > 20      # the generator does produce valid descriptors too


def test_julia_struct_mirrors_match_the_ctypes_mirrors(qc):
    """julia/QCollocHIP.jl cannot be run here (no Julia in the image): at least its struct mirrors are checked field by field --
    name, order and type -- against the ctypes mirrors, whose sizes the library itself confirms at import (qc_sizeof_*)."""
    import ctypes as C
    txt = open(os.path.join(ROOT, "julia", "QCollocHIP.jl"), encoding="utf-8").read()
    L = qc._lib
    tmap = {"Int32": C.c_int32, "Int64": C.c_int64, "Float64": C.c_double}

    def julia_fields(name):
        body = re.search(r"^struct " + name + r"\n(.*?)^end", txt, flags=re.S | re.M).group(1)
        body = re.sub(r"#[^\n]*", "", body)
        out = []
        for decl in re.split(r"[;\n]", body):
            decl = decl.strip()
            if not decl:
                continue
            fname, ftype = [x.strip() for x in decl.split("::")]
            if ftype.startswith("Ptr{"):
                ct = "ptr"
            elif ftype.startswith("NTuple{"):
                count, elem = ftype[len("NTuple{"):-1].split(",")
                ct = tmap[elem.strip()] * (L.QC_MAX_DERIV if count.strip() == "QC_MAX_DERIV" else int(count))
            else:
                ct = tmap[ftype]
            out.append((fname, ct))
        return out

    for jname, mirror in (("QCDesc", L.qc_desc), ("QCDims", L.qc_dims_t), ("QCTermsDesc", L.qc_terms_desc), ("QCFidelityDesc", L.qc_fidelity_desc)):
        jf = julia_fields(jname)
        assert [f for f, _ in jf] == [f for f, _ in mirror._fields_], jname
        for (fname, jt), (_, ct) in zip(jf, mirror._fields_):
            if jt == "ptr":
                assert issubclass(ct, C._Pointer), (jname, fname)
            else:
                assert C.sizeof(jt) == C.sizeof(ct) and (jt is ct or jt._type_ == getattr(ct, "_type_", None)), (jname, fname)
    assert "qc_sizeof_desc" in txt and "qc_create_multi" in txt and "F!" in txt


def _c_prototypes():
    """name -> (return kind, [argument kinds]) of every function include/qcolloc.h declares; kinds: 'ptr', 'i32', 'i64', 'f64', 'void'."""
    txt = open(os.path.join(ROOT, "include", "qcolloc.h"), encoding="utf-8").read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)

    def kind(t):
        t = t.strip()
        if "*" in t:
            return "ptr"
        base = t.replace("const", "").split()
        if not base or base == ["void"]:
            return "void"
        if len(base) > 1 and re.fullmatch(r"\w+", base[-1]) and base[-1] not in ("int", "int32_t", "int64_t", "double"):
            base = base[:-1]                     # drop the parameter name
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64"}[" ".join(base)]

    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?(?:int|int32_t|int64_t|void|char|double)\s*\*?)\s*(qc_\w+)\s*\(([^;{}]*)\)\s*;", txt, flags=re.M):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a for a in (x.strip() for x in args.replace("\n", " ").split(",")) if a and a != "void"]
        protos[name] = (kind(ret), [kind(a) for a in args])
    return protos


def test_julia_ccalls_match_the_header():
    """julia/*.jl cannot be run here; every `ccall((:qc_..., LIB[]), Ret, (Args...), ...)` in them is checked against the C prototype
    of include/qcolloc.h: the function exists, the arity matches, integers have the right width, pointers are pointers."""
    protos = _c_prototypes()
    assert {"qc_create", "qc_eval_F_jac", "qc_eval_hess", "qc_set_new_x", "qc_abi_version", "qc_terms_eval"} <= set(protos), sorted(protos)[:5]

    def jkind(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Clonglong": "i64", "Float64": "f64", "Cdouble": "f64", "Cvoid": "void"}[t]

    def split_top(s):            # split on commas outside braces / parentheses
        out, depth, cur = [], 0, ""
        for ch in s:
            depth += ch in "{(" 
            depth -= ch in "})"
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur)
        return [x.strip() for x in out if x.strip()]

    seen = 0
    for fn in ("QCollocHIP.jl", "reconcile.jl"):
        txt = open(os.path.join(ROOT, "julia", fn), encoding="utf-8").read()
        for m in re.finditer(r"ccall\(\(:(qc_\w+),\s*LIB\[\]\),\s*([\w{}]+),\s*\(", txt):
            name, ret = m.group(1), m.group(2)
            depth, i = 1, m.end()
            while depth:                          # the argument-type tuple, to its closing parenthesis
                depth += txt[i] == "("
                depth -= txt[i] == ")"
                i += 1
            jargs = split_top(txt[m.end():i - 1])
            assert name in protos, f"{fn}: {name} is not declared in include/qcolloc.h"
            cret, cargs = protos[name]
            assert jkind(ret) == cret, (fn, name, ret, cret)
            assert [jkind(a) for a in jargs] == cargs, (fn, name, jargs, cargs)
            seen += 1
    assert seen >= 20, seen


def test_block_orders_permute_the_structure_blockwise(qc, oracle):
    """qc_desc.jac_block_order / hess_block_order (ABI 0.6; SURVEY 7 "permutation hook"): the value blocks of an interval in any order --
    the same COO set, every block contiguous and internally unchanged, the blocks in the order asked for; descriptors that are not
    permutations are refused."""
    import ctypes as C
    L = qc._lib
    rng = np.random.default_rng(11)
    for integ in ("pade", "exponential"):
        for free_time in (True, False):
            inp = qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(2), qc.GATES["CNOT"], 5, free_time=free_time, integrator=integ)
            d0, _k0 = qc.make_desc(inp.integrators, inp.traj)
            jr0, jc0, hr0, hc0 = qc.desc_structures(d0)
            n_int = inp.traj.T - 1
            dims0 = qc.desc_dims(d0)
            for trial in range(4):
                jo, ho = rng.permutation(L.QC_JAC_BLOCKS), rng.permutation(L.QC_HESS_BLOCKS)
                d1, _k1 = qc.make_desc(inp.integrators, inp.traj, jac_block_order=jo, hess_block_order=ho)
                dims1 = qc.desc_dims(d1)
                assert (dims1.jac_nnz, dims1.hess_nnz, dims1.n_rows) == (dims0.jac_nnz, dims0.hess_nnz, dims0.n_rows)
                jr1, jc1, hr1, hc1 = qc.desc_structures(d1)
                for (r0, c0, r1, c1, order, what) in ((jr0, jc0, jr1, jc1, jo, "dF"), (hr0, hc0, hr1, hc1, ho, "mu_d2F")):
                    a = np.stack([r0, c0], 1).reshape(n_int, -1, 2)[0]
                    b = np.stack([r1, c1], 1).reshape(n_int, -1, 2)[0]
                    assert sorted(map(tuple, a)) == sorted(map(tuple, b)), what            # the same COO set
                    # default-order block lengths from the default structure: cut a at its block boundaries, reassemble in `order`
                    prob = problem_from_inputs(inp)
                    s, m, ft, pade = prob.s, prob.m, prob.free_time, prob.integrator == oracle.PADE
                    dl = sum(dv.dim for dv in prob.derivs)
                    if what == "dF":
                        lens = [prob.nc * prob.n ** 2, prob.nc * prob.n ** 2 if pade else s, s * m, s if ft else 0, (4 if ft else 3) * dl]
                    else:
                        ub = 1 if pade else 0
                        lens = [s * m, ub * s * m, s if ft else 0, ub * s if ft else 0, m * (m + 1) // 2, m if ft else 0, 1 if ft else 0, dl if ft else 0]
                    assert sum(lens) == a.shape[0], what
                    starts = np.concatenate([[0], np.cumsum(lens)])
                    want = np.concatenate([a[starts[k]:starts[k + 1]] for k in order]) if a.size else a
                    np.testing.assert_array_equal(b, want, err_msg=f"{what} {integ} ft={free_time} order={list(order)}")
    bad, _ = qc.make_desc(inp.integrators, inp.traj)
    bad.jac_block_order[0] = 1                     # (1, 0, 0, 0, 0): not a permutation
    out = L.qc_dims_t()
    assert L.lib.qc_desc_dims(C.byref(bad), C.byref(out)) == L.QC_ERR_INVALID and b"permutation" in L.lib.qc_last_error(None)
    with pytest.raises(ValueError):
        qc.make_desc(inp.integrators, inp.traj, hess_block_order=[0, 1, 2])
