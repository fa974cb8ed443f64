#!/bin/bash
# A/B of the two-wave Hessian kernel (qc_mfma_hess2.hip) against the one-wave kernel (QC_HESS_TWO_WAVES=0).
set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/hess2_ab.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or hess or Hess or mu_d2F" 2>&1 | tail -3 >> $O
for T in 0 250 500; do
for v in 0 1; do
  echo "== QC_HESS_TWO_WAVES=$v T=$T" >> $O
  QC_HESS_TWO_WAVES=$v timeout 300 python profiles/fused_bench.py 3 $T 2>&1 | grep config | tail -2 >> $O
done
done
timeout 300 python profiles/stamps_fused.py 1000 hess 2>&1 | grep -v amdgpu.ids >> $O
cat $O
