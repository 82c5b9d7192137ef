#!/bin/bash
# placement walk with geometric spacers: consecutive processes, traced
for rep in 1 2 3 4; do
  C21CM_WS_TRACE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('512', 'ms', round(d['ms_per_step'],2), r['kernel'][:26], round(r['ms_per_launch'],4), [round(k['ms'],4) for k in r['other_kernels']])"
  grep "\[place\]" /tmp/err.txt | sed 's/\[place\] slot //; s/candidate [^ ]* after //; s/ GB: /:/; s/ ms//' | tr '\n' ';'; echo
done
for rep in 1 2; do
  C21CM_WS_TRACE=1 python bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('1024', 'ms', round(d['ms_per_step'],1), r['kernel'][:26], round(r['ms_per_launch'],3), [round(k['ms'],3) for k in r['other_kernels']])"
  grep "\[place\]" /tmp/err.txt | sed 's/\[place\] slot //; s/candidate [^ ]* after //; s/ GB: /:/; s/ ms//' | tr '\n' ';'; echo
done
