#!/bin/bash
# the second radius' density work spectrum placed against the first's too (two-radius pass X writes both): A/B
for rep in 1 2 3 4; do for px in 1 0; do
  C21CM_WS_PLACE_X=$px python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('512 place_x=$px', 'ms', round(d['ms_per_step'],2), r['kernel'][:26], round(r['ms_per_launch'],4), [round(k['ms'],4) for k in r['other_kernels']])"
done; done
