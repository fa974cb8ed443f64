cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
bash profiles/collect.sh r04b > gpurun_out/r04/collect_r04b.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/prof_r04b/summary.json"))
for k, v in d.items():
    print(k[28:96], round(v["avg_us"], 2), v.get("hbm_bytes_per_launch_corrected"))
b = json.loads(open("gpurun_out/prof_r04b/bench_trace.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"])
PY
for s in 21 22 23; do python tests/stress_host_gpu.py 400 $s 2>&1 | tail -1 | cut -c1-110; done
python tests/stress_gpu.py 400 31 2>&1 | tail -1 | cut -c1-200
