#!/bin/bash
# multi-shard handle (4 shards on one device, T = 4000): re-arm / affinity / worker variants
out=${1:-gpurun_out/r03k}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 4000 4 >> $out/host_path_multi.txt 2>> $out/host_path.err; }
run QC_X=1
run QC_X=2
run QC_HOST_REARM=inline
run QC_HOST_AFFINITY=0
run QC_HOST_THREADS=8
run QC_HOST_THREADS=12
run QC_HOST_REARM_JOBS=1
run QC_HOST_LANDING=0
QC_HOST_TRACE=1 python profiles/host_path_r03.py 3 4000 4 2> $out/host_trace_multi.txt > /dev/null
python profiles/host_path_r03.py 3 4000 >> $out/host_path_multi.txt 2>> $out/host_path.err
