# K = 20 (the driver's command) three times each way: the closing event polled before the synchronize (rounds 3 - 5) or not (round 6)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  QC_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-visible --no-config5 > gpurun_out/k20_$i.json 2> gpurun_out/k20_$i.err
  QC_BENCH_POLL=1 QC_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-visible --no-config5 > gpurun_out/k20p_$i.json 2> gpurun_out/k20p_$i.err
done
grep -H "timed region" gpurun_out/k20_*.err gpurun_out/k20p_*.err
python - <<'PY'
import json
for tag in ("k20", "k20p"):
    for i in (1,2,3):
        d=json.loads(open(f'gpurun_out/{tag}_{i}.json').read().strip().splitlines()[-1])
        print(tag, d['value'], d['ms_per_step'], d['roofline']['step_us_stream_events'], d['roofline']['kernel_us_event_pairs'])
PY
