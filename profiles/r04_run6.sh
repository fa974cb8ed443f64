cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_closures.py tests/test_evaluator.py tests/test_end_to_end.py tests/test_multi_device.py -x -q -m gpu > gpurun_out/r04/run6_tests.txt 2>&1
tail -12 gpurun_out/r04/run6_tests.txt
python bench.py --steps 200 --warmup 20 2>gpurun_out/r04/bench_n1_b.err | tail -1 > gpurun_out/r04/bench_n1_b.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_n1_b.json"))
hv = d["host_visible"]
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("hess_us"), d.get("F_dF_hess_one_call_us"), d.get("F_only_us"))
print({k: hv.get(k) for k in ("F_dF_ms", "F_ms", "F_registered_ms", "hess_ms", "jac_same_x_ms", "ipopt_sequence_ms", "closure_ms", "closure_fresh_ms")})
print(d.get("config5"))
PY
