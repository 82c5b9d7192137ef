#!/bin/bash
# what the driver runs at round end: smoke(), the gpu suite without xdist (timed), the default bench line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python -m pytest tests/ -x -q -m gpu > gpurun_out/suite_serial.out 2>&1 ) 2>&1 | grep real
grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/suite_serial.out | tail -2
( time python bench.py > gpurun_out/bench_default.json 2>/dev/null ) 2>&1 | grep real
python -c "
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['ms_per_step'],2), r['kernel'][:30], round(r['ms_per_launch'],4), round(r['frac'],3), r.get('traffic_profile_stale'), 'rloop', round(r['r_loop']['frac'],3), 'cpu', d['cpu_baseline']['value'], d['config']['work_spectra_placement'])"
