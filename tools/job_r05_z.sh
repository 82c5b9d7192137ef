#!/bin/bash
# single-grid pass Y out of place (pass X into a scratch spectrum, pass Y from there into the work spectrum): A/B
for rep in 1 2; do for o in 1 0; do
  echo "== C21CM_Y_OOP=$o"
  C21CM_Y_OOP=$o python bench.py --mode erfc --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('erfc ms', round(d['ms_per_step'],2), r['kernel'][:30], round(r['ms_per_launch'],4), [(k['kernel'][:24], round(k['ms'],4)) for k in r['other_kernels']], 'xH', d['config'].get('global_xH'))"
  C21CM_Y_OOP=$o PYTHONPATH=. python tools/time_abi_ionize.py 512 0 2>/dev/null | tail -1 | cut -c1-150
done; done
for o in 1 0; do echo "== C21CM_Y_OOP=$o"; C21CM_Y_OOP=$o python tools/time_coeval_ts.py 512 1024 6.0 2>/dev/null | tail -1 | cut -c1-260; done
