#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   -- gpurun, retried while the pod has no free GPU slot (nothing is charged for those)
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1)
  echo "$out"
  echo "$out" | grep -q "status=transient" || exit 0
  sleep 150
done
