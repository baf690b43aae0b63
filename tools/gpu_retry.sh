#!/bin/bash
# usage: tools/gpu_retry.sh <timeout-seconds> <command...>   - retries gpurun while no box / slot is free (exit code 3: nothing charged)
T=$1; shift
for k in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
