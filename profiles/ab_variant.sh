#!/bin/bash
# A/B of the product library against csrc/libqcolloc_hip.<name>.so (profiles/build_variant.sh): bash profiles/ab_variant.sh <name>
cd ${GRAFT_REPO_ROOT:-.}
v=${1:-base}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config_parity or random_problem or layouts or new_x or ket" 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2 3; do for lib in $v ""; do QCOLLOC_HIP_VARIANT=$lib python bench.py --no-host-visible --cpu-seconds 0 --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant=[$lib]', round(d['value']), 'F+dF', round(d['ms_per_step']*1e3,3), 'hess', round(d['hess_us'],2), 'F only', round(d['F_only_us'],2), 'one call', round(d['F_dF_hess_one_call_us'],2), 'config1/2', d.get('small_configs'))"; done; done
