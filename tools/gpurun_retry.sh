#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3): tools/gpurun_retry.sh [--timeout N] -- '<command>'
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
