cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --no-host-visible --no-config5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('live:', round(d['value']), d['ms_per_step'], d['roofline']['step_us_stream_events'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
bash profiles/collect.sh r04c > gpurun_out/collect_r04c.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/prof_r04c/summary.json"))
for k, v in d.items():
    print(k[28:96], round(v["avg_us"], 2), v.get("hbm_bytes_per_launch_corrected"))
PY
