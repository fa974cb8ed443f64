#!/bin/bash
# rocprofv3 kernel-trace statistics of the secondary kernels (any-order Pade, 5-qubit MFMA64 kernels):
#   gpurun -- 'bash profiles/collect_round1b.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r01b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/variants -- python3 $R/profiles/time_variants.py > $OUT/variants.txt 2> $OUT/variants.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma64 -- python3 $R/profiles/time_mfma64.py > $OUT/mfma64.txt 2> $OUT/mfma64.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mfma64h -- python3 $R/profiles/time_mfma64_hess.py > $OUT/mfma64h.txt 2> $OUT/mfma64h.log
for d in variants mfma64 mfma64h; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d: $f"; head -25 "$f" | cut -c1-220
  cp "$f" $OUT/${d}_kernel_stats.csv
done
cat $OUT/variants.txt
