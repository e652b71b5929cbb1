#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3): tools/gr.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
