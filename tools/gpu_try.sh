#!/bin/bash
# Retry ACQUIRING a GPU box while gpurun answers "no slot free" (exit 3: nothing ran, nothing charged).  A command that
# did run is never repeated.   usage: tools/gpu_try.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 90
done
exit 3
