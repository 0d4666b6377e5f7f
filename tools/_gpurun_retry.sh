#!/bin/bash
# usage: tools/_gpurun_retry.sh <timeout_s> '<command>'   -- retries while the pod's GPU slots are busy (nothing is charged then)
T=$1; shift
for i in $(seq 1 40); do
    out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
    if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
    echo "$out" | tail -40
    exit 0
done
echo "gave up: slots busy"
