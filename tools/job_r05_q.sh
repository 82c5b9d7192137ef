#!/bin/bash
export TMPDIR=/tmp
run() { echo -n "== $* : "; for i in 1 2 3; do env "$@" timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), end=' ')"; done; echo; }
run A=1
for k in 128 256 384 512 640 768 1024 1152 2048 2176 4224 8320 65664 1048704; do run C21CM_ARENA=$k; done
