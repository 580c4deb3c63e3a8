#!/bin/bash
# usage: scripts/gpurun_retry.sh <timeout> '<command>'   -- retries only while gpurun reports "no slot / no box" (exit 3: nothing ran, nothing charged)
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
