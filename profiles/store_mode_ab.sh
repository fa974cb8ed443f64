#!/bin/bash
# Store flavour of the F + dF kernel's outputs, re-measured on round 5's kernel: QC_STORE_MODE 0 plain, 1 write-through (sc1), 2 non-temporal (default)
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2 3; do for mde in 2 1 0; do QC_STORE_MODE=$mde python bench.py --no-host-visible --no-config4 --no-config5 --cpu-seconds 0 --steps 1000 --warmup 100 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('store mode $mde: step', round(d['roofline']['step_us_stream_events'],3), 'us; F only', round(d['F_only_us'],2), '; mu_d2F', round(d['hess_us'],2), '; one call', round(d['F_dF_hess_one_call_us'],2), d['F_dF_hess_kernel'])"; done; done
