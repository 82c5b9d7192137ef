#!/bin/bash
# placement shopping (refined rule): consecutive processes, with and without
for rep in 1 2 3 4 5; do for pl in 1 0; do
  C21CM_WS_PLACE=$pl python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('512 place=$pl', 'ms', round(d['ms_per_step'],2), r['kernel'][:26], round(r['ms_per_launch'],4), [round(k['ms'],4) for k in r['other_kernels']])"
done; done
for rep in 1 2 3; do for pl in 1 0; do
  C21CM_WS_PLACE=$pl python bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('1024 place=$pl', 'ms', round(d['ms_per_step'],1), r['kernel'][:26], round(r['ms_per_launch'],3), [round(k['ms'],3) for k in r['other_kernels']])" || tail -3 /tmp/err.txt
done; done
