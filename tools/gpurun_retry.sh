#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing ran, nothing charged).  usage: gpurun_retry.sh <log> <timeout> <cmd>
LOG=$1; TMO=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
