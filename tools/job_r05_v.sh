#!/bin/bash
# the whole GPU suite under heavier contention, twice: anything that only shows when processes share the GPU?
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
  timeout 2400 python -m pytest tests -q -m gpu -n 8 -p no:cacheprovider > gpurun_out/r05v_tests_$i.log 2>&1
  tail -4 gpurun_out/r05v_tests_$i.log | cut -c1-200
  grep "^FAILED" gpurun_out/r05v_tests_$i.log | head
done
