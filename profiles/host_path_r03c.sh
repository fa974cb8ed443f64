#!/bin/bash
# third sweep: team members pinned to distinct core complexes on the GPU's NUMA node (default) vs left to the scheduler
out=${1:-gpurun_out/r03d}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 >> $out/host_path.txt 2>> $out/host_path.err; }
export QC_HOST_HESS_CHUNKS=1
for rep in 1 2; do
run QC_HOST_AFFINITY=1
run QC_HOST_AFFINITY=0
run QC_HOST_AFFINITY=1 QC_HOST_THREADS=7
run QC_HOST_AFFINITY=1 QC_HOST_THREADS=12
run QC_HOST_AFFINITY=1 QC_HOST_THREADS=15
run QC_HOST_AFFINITY=0 QC_HOST_THREADS=15
done
QC_HOST_TRACE=1 QC_HOST_AFFINITY=1 python profiles/host_path_r03.py 3 2> $out/host_trace_aff1.txt > /dev/null
QC_HOST_TRACE=1 QC_HOST_AFFINITY=0 python profiles/host_path_r03.py 3 2> $out/host_trace_aff0.txt > /dev/null
env python profiles/host_path_r03.py 5 >> $out/host_path.txt 2>> $out/host_path.err
