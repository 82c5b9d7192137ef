#!/bin/bash
export TMPDIR=/tmp
run() { echo "== $*"; for i in 1 2 3 4 5; do env "$@" timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), end=' ')"; done; echo; }
run A=1
run C21CM_ARENA=0
run C21CM_ARENA=4096
run C21CM_ARENA=1048576
run C21CM_ARENA=33554432
