#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged).  usage: gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
