#!/bin/bash
# full GPU suite with the two-phase placement walk: xdist (short walks), then serially (full walks), timed
python -m pytest tests -x -q -m gpu -n 6 > gpurun_out/full_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/full_tests.out | tail -2
( time python -m pytest tests -x -q -m gpu > gpurun_out/suite_serial.out 2>&1 ) 2>&1 | grep real
grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/suite_serial.out | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
