#!/bin/bash
# slab finish tests + pass X / Y item-order A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ionize.py -x -q -m gpu -k "slab or shard or sharded or two_radii or parity or config4 or edge" > gpurun_out/r05b_tests.log 2>&1
tail -15 gpurun_out/r05b_tests.log
for cfg in "0 0" "1 0" "1 1" "1 2" "0 1" "0 0"; do
  set -- $cfg
  echo "XORDER=$1 YORDER=$2" >> gpurun_out/r05b_order.txt
  C21CM_XORDER=$1 C21CM_YORDER=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05b_order.txt
done
cat gpurun_out/r05b_order.txt
