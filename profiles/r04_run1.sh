# round 4, run 1: the closure ring / knot-generation tests, the driver-style N = 2 start, the default bench line, an 8-rank gloo dry run
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_closures.py tests/test_end_to_end.py tests/test_evaluator.py tests/test_multi_device.py tests/test_rollout.py -x -q -m gpu > gpurun_out/r04/run1_tests.txt 2>&1
tail -5 gpurun_out/r04/run1_tests.txt
python bench.py 2>gpurun_out/r04/bench_n1.err | tail -1 > gpurun_out/r04/bench_n1_a.json
QC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 2>gpurun_out/r04/bench_gloo_n8.err | tail -1 > gpurun_out/r04/bench_gloo_n8.json
python - <<'PY'
import json
for f in ("gpurun_out/r04/bench_n1_a.json", "gpurun_out/r04/bench_gloo_n8.json"):
    try:
        d = json.load(open(f))
        hv = d.get("host_visible") or {}
        print(f, d["n_gpus"], round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("hess_us"), d.get("F_dF_hess_one_call_us"))
        print("  host:", {k: hv.get(k) for k in ("F_dF_ms", "F_ms", "hess_ms", "jac_same_x_ms", "ipopt_sequence_ms", "closure_ms", "closure_fresh_ms", "error")})
        print("  c5:", {k: v for k, v in (d.get("config5") or {}).items() if k.endswith("_us") or k.endswith("frac")})
    except Exception as e:
        print(f, "unreadable:", e)
PY
