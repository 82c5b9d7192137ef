#!/bin/bash
# what the driver runs at round end: smoke(), the gpu suite without xdist (time it), the default bench line
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5
( time python -m pytest tests/ -x -q -m gpu > gpurun_out/suite_serial.out 2>&1 ) 2>&1 | tail -4
grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/suite_serial.out | tail -3
( time python bench.py > gpurun_out/bench_default.json 2>/dev/null ) 2>&1 | tail -4
cut -c1-250 gpurun_out/bench_default.json
