cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r03_bench_n1.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_n1_default.json
for n in 2 4; do
  QC_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_gloo_n$n.json
done
for f in gpurun_out/r03_bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f'))
print('$f', d['n_gpus'], round(d['value']), d['ms_per_step'], d['roofline']['frac'], d.get('hess_us'), d.get('F_dF_hess_one_call_us'), (d.get('host_visible') or {}).get('F_dF_ms'), (d.get('host_visible') or {}).get('hess_ms'), (d.get('host_visible') or {}).get('ms_per_ipopt_iter'), d['cpu_baseline']['value'])
"; done
