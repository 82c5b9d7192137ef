#!/bin/bash
# x_e work spectrum of the Eulerian models and the second grid of the spin-temperature filter stage placed against
# their sweep partners: tests, then E-INTEGRAL + x_e through the ABI and config 5 with / without
python -m pytest tests -x -q -m gpu -n 6 > gpurun_out/full_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/full_tests.out | tail -3
for pl in 1 0 1 0; do echo "== C21CM_WS_PLACE=$pl"; C21CM_WS_PLACE=$pl PYTHONPATH=. python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 | cut -c1-330; done
for pl in 1 0; do echo "== C21CM_WS_PLACE=$pl"; C21CM_WS_PLACE=$pl python tools/time_coeval_ts.py 512 1024 6.0 2>/dev/null | tail -1 | cut -c1-330; done
