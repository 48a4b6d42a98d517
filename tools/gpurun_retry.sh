#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'
# gpurun, retried ONLY while the pod has no free box / GPU slot (gpurun's exit code 3: the job never started and nothing is
# charged).  Every other outcome -- the command's own failure, a run that started and died, a refusal -- is passed on with
# gpurun's exit status and is NOT re-run: a job that died on the GPU is something to read, not to repeat.
t=$1; shift
rc=3
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -eq 3 ] || exit $rc
  echo "gpurun_retry: no box or slot free (attempt $i), waiting" >&2
  sleep 150
done
exit $rc
