#!/bin/bash
# tools/gpu_retry.sh <timeout_s> <logfile> '<command>': gpurun with retries while no GPU slot is free (exit code 3)
to=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
