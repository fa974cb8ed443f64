#!/bin/bash
# bench.py (F + dF, config 3) for the product library and each experiment build; one line per variant
for v in "" "$@"; do
  out=$(QCOLLOC_HIP_VARIANT=$v python bench.py --steps 1500 --warmup 100 --cpu-seconds 0 --no-hessian --no-host-visible 2>/dev/null)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', '${v:-product}', 'step_us', round(d['roofline']['step_us_stream_events'],3), 'pairs_us', round(d['roofline']['kernel_us_event_pairs'],3))"
done
