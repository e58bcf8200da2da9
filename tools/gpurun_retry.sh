#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3 = nothing charged): tools/gpurun_retry.sh <timeout s> '<command>'
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
