#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for e in "A=1" "C21CM_LIB=variants/xp2/lib21cmfast_hip.so"; do echo "$e"; env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py; done
done
