#!/bin/bash
# virtual-memory-API buffers with LARGE chunks: is there a penalty?  (1024^3 bench, placement off so that slots stay VMM-made)
for mb in 0 2 64 1024 2200; do
  if [ $mb = 0 ]; then al=plain; else al=scatter; fi
  C21CM_WS_PLACE=0 C21CM_WS_ALLOC=$al C21CM_WS_SCATTER_MB=$mb python bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('1024 chunk_mb=$mb', 'ms', round(d['ms_per_step'],1), r['kernel'][:26], round(r['ms_per_launch'],3), [round(k['ms'],3) for k in r['other_kernels']])" || tail -3 /tmp/err.txt
done
