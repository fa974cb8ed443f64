#!/bin/bash
# fifth sweep: re-arm intensity (jobs) x ring size; tight-loop new-x calls are the sensitive ones
out=${1:-gpurun_out/r03g}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 >> $out/host_path.txt 2>> $out/host_path.err; }
for rep in 1 2; do
run QC_HOST_NBUF=4 QC_HOST_REARM_JOBS=1
run QC_HOST_NBUF=4 QC_HOST_REARM_JOBS=2
run QC_HOST_NBUF=3 QC_HOST_REARM_JOBS=2
run QC_HOST_NBUF=2 QC_HOST_REARM_JOBS=8
run QC_HOST_REARM=inline
done
QC_HOST_TRACE=1 QC_HOST_NBUF=4 QC_HOST_REARM_JOBS=1 python profiles/host_path_r03.py 3 2> $out/host_trace_j1.txt > /dev/null
python profiles/host_path_r03.py 5 >> $out/host_path.txt 2>> $out/host_path.err
