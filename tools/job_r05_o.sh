#!/bin/bash
# process-to-process variation of the line passes on ONE box (is the "box spread" an allocation effect?)
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py; done
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py; done
