#!/bin/bash
# Waits for a free GPU box: repeats a gpurun call only while it is refused as transient (exit 3 = nothing ran, nothing charged).
# usage: tools/gpu_when_free.sh <timeout-seconds> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 150
done
exit 3
