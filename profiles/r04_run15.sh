cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu -x > gpurun_out/r04/run15_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r04/run15_tests.txt | tail -2
python bench.py 2>gpurun_out/r04/bench_n1_d.err | tail -1 > gpurun_out/r04/bench_n1_d.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_n1_d.json"))
hv = d["host_visible"]
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d.get("hess_us"), d.get("F_dF_hess_one_call_us"), d.get("F_only_us"))
print({k: hv.get(k) for k in ("F_dF_ms", "F_ms", "F_pinned_ms", "hess_ms", "jac_same_x_ms", "ipopt_sequence_ms", "closure_ms")})
print({k: v for k, v in d["config5"].items() if not isinstance(v, (dict, list))})
PY
