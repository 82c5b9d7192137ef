#!/bin/bash
# placement walk: a few processes on whatever box this is
for rep in 1 2 3; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('512', 'ms', round(d['ms_per_step'],2), r['kernel'][:26], round(r['ms_per_launch'],4), [round(k['ms'],4) for k in r['other_kernels']])"
done
python bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('1024', 'ms', round(d['ms_per_step'],1), r['kernel'][:26], round(r['ms_per_launch'],3), [round(k['ms'],3) for k in r['other_kernels']])"
