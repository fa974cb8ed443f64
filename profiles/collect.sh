#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh <tag> [extra bench.py args]'
# Writes gpurun_out/prof_<tag>/{trace,fetch,write}/...csv.  Kernel trace and each PMC counter are
# separate passes (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one
# pass; gpurun refuses --pmc combined with other trace domains).
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--prewarm-seconds 0: one pass over the output ring only -- tens of thousands of back-to-back launches under the profiler
#  inflate the traced kernel durations by 5 - 10 %: 9.39 against 8.57 us for the same kernel in the same call)
ARGS="--steps 300 --warmup 30 --cpu-seconds 0 --prewarm-seconds 0 --no-host-visible $*"   # (host-buffer calls launch the same kernels in their compact form: kept out of the per-kernel averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
python3 $R/profiles/summarize.py $OUT
