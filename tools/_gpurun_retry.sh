#!/bin/bash
# usage: tools/_gpurun_retry.sh <timeout_s> '<command>'   -- retries while the pod's GPU slots are busy (nothing is charged then).
# Exit status: gpurun's own for the call that ran (so scripts chained behind this stop on a failed or refused run), 75 when it gave up.
T=$1; shift
for i in $(seq 1 40); do
    out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
    rc=$?
    if [ $rc -eq 3 ] || echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
    echo "$out" | tail -${GPURUN_TAIL:-40}
    exit $rc
done
echo "gave up: slots busy"
exit 75
