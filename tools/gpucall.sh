#!/bin/bash
# tools/gpucall.sh <timeout-seconds> <script> <log>: runs a command file through gpurun, retrying while no GPU slot is free (exit code 3)
t=$1; script=$2; log=$3
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "bash $script" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $log; then exit $rc; fi
  sleep 90
done
exit 3
