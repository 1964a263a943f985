#!/bin/bash
# Submit one gpurun call; when no box/slot is free (exit 3: nothing ran, nothing charged) wait and submit again.
# usage: tools/gpurun_retry.sh TIMEOUT 'command'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
