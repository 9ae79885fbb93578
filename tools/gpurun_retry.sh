#!/bin/bash
# tools/gpurun_retry.sh <timeout> <command...>: gpurun, retried while the pod's GPU slots are busy (exit code 3: nothing charged)
to=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
