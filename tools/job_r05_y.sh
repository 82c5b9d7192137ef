#!/bin/bash
# recombination loop with the whalo_sfr work spectrum placed against its sweep partner: tests + timing, with / without
python -m pytest tests/test_gpu_recomb.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -3
for pl in 1 0 1 0; do echo "== C21CM_WS_PLACE=$pl"; C21CM_WS_PLACE=$pl python tools/time_recomb.py 512 3 2>&1 | tail -1; done
