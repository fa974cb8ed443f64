#!/bin/bash
# Sweep of the host path's switches at BASELINE config 3 (and 5): one process per setting.  Output: gpurun_out/r03b/host_path.txt
out=${1:-gpurun_out/r03b}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 >> $out/host_path.txt 2>> $out/host_path.err; }
run QC_HOST_LANDING=1
run QC_HOST_LANDING=1
run QC_HOST_LANDING=0
for t in 4 6 10 12 15; do run QC_HOST_THREADS=$t; done
for kb in 32 64 256 512; do run QC_HOST_PIECE_KB=$kb; done
for c in 1 2 8; do run QC_HOST_HESS_CHUNKS=$c; done
run QC_HOST_NT=0
env python profiles/host_path_r03.py 5 >> $out/host_path.txt 2>> $out/host_path.err
env QC_HOST_LANDING=0 python profiles/host_path_r03.py 5 >> $out/host_path.txt 2>> $out/host_path.err
env python profiles/host_path_r03.py 1 >> $out/host_path.txt 2>> $out/host_path.err
env python profiles/host_path_r03.py 2 >> $out/host_path.txt 2>> $out/host_path.err
env python profiles/host_path_r03.py 3 4000 >> $out/host_path.txt 2>> $out/host_path.err
env python profiles/host_path_r03.py 3 4000 4 >> $out/host_path.txt 2>> $out/host_path.err
QC_HOST_TRACE=1 python profiles/host_path_r03.py 3 2> $out/host_trace.txt > /dev/null
