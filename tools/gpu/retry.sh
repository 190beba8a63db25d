#!/bin/bash
# tools/gpu/retry.sh <log> <timeout s> '<command>': gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
LOG=$1; T=$2; CMD=$3
for i in $(seq 1 20); do
  gpurun --timeout $T -- "$CMD" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
