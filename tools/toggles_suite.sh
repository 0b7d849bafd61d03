#!/bin/bash
# full GPU suite under each opt-out switch; stops at the first abnormal exit (not at ordinary test failures)
for e in "RT_FRAMES_IN_FLIGHT=1" "RT_GATHER_BATCH=0" "RT_GATHER_FUSED=0" "RT_PROBE_SIDE=0" "RT_FUSE_QUEUED=0" "RT_FUSE_PRIMARY=0" "RT_FUSE_TAIL=0"; do
  echo "== $e" >> gpurun_out/toggles.log
  env $e timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -15 >> gpurun_out/toggles.log
  rc=${PIPESTATUS[0]}
  echo "rc=$rc" >> gpurun_out/toggles.log
  echo "$e rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
