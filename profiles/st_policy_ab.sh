#!/bin/bash
# Cache policy of the F + dF kernel's stores (variant builds -DQC_ST_POLICY_ID=k put policy k on QC_STORE_MODE=1): 1 "sc1 nt", 2 "sc0 nt", 3 "sc0 sc1 nt", 4 "sc0 sc1", 5 "sc0"
cd ${GRAFT_REPO_ROOT:-.}
run() { QCOLLOC_HIP_VARIANT=$1 QC_STORE_MODE=$2 python bench.py --no-host-visible --no-config4 --no-config5 --no-hessian --cpu-seconds 0 --steps 1000 --warmup 100 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3: step', round(d['roofline']['step_us_stream_events'],3), 'us')"; }
for r in 1 2; do run "" 2 "nt (product)"; run st1 1 "sc1 nt"; run st2 1 "sc0 nt"; run st3 1 "sc0 sc1 nt"; run st4 1 "sc0 sc1"; run st5 1 "sc0"; done
