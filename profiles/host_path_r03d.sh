#!/bin/bash
# fourth sweep: deferred vs inline re-arm, raw copy time (NOWATCH)
out=${1:-gpurun_out/r03e}; mkdir -p $out
run() { env "$@" python profiles/host_path_r03.py 3 >> $out/host_path.txt 2>> $out/host_path.err; }
export QC_HOST_HESS_CHUNKS=1
for rep in 1 2; do
run QC_HOST_REARM=deferred
run QC_HOST_REARM=inline
run QC_HOST_REARM=deferred QC_HOST_THREADS=12
done
run QC_HOST_NOWATCH=1
QC_HOST_TRACE=1 python profiles/host_path_r03.py 3 2> $out/host_trace.txt > /dev/null
QC_HOST_TRACE=1 QC_HOST_NOWATCH=1 python profiles/host_path_r03.py 3 2> $out/host_trace_nowatch.txt > /dev/null
python profiles/host_path_r03.py 5 >> $out/host_path.txt 2>> $out/host_path.err
python profiles/host_path_r03.py 3 8000 >> $out/host_path.txt 2>> $out/host_path.err
