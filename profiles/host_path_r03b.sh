#!/bin/bash
# second sweep: refill flavour, thread counts, trace with copy-completion stamps
out=${1:-gpurun_out/r03c}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 >> $out/host_path.txt 2>> $out/host_path.err; }
export QC_HOST_HESS_CHUNKS=1
run QC_X=1
run QC_X=2
run QC_HOST_FILL_NT=0
run QC_HOST_FILL_NT=0
for t in 3 5 12 15; do run QC_HOST_THREADS=$t; done
for kb in 128 512 1024; do run QC_HOST_PIECE_KB=$kb QC_HOST_THREADS=5; done
numactl -H > $out/numa.txt 2>&1
for node in 0 1; do numactl --cpunodebind=$node --membind=$node python profiles/host_path_r03.py 3 >> $out/host_path_numa$node.txt 2>> $out/host_path.err; done
QC_HOST_TRACE=1 python profiles/host_path_r03.py 3 2> $out/host_trace.txt > /dev/null
QC_HOST_TRACE=1 QC_HOST_THREADS=5 python profiles/host_path_r03.py 3 2> $out/host_trace_t5.txt > /dev/null
QC_HOST_TRACE=1 QC_HOST_FILL_NT=0 python profiles/host_path_r03.py 3 2> $out/host_trace_fill0.txt > /dev/null
