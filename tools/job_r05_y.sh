#!/bin/bash
# recombination loop: the last variant (filtered N_rec + x_e, four spectra) on the fused loop
python -m pytest tests/test_gpu_recomb.py tests/test_gpu_reference_fixtures.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -12
python tools/time_recomb.py 512 3 > gpurun_out/recomb_y.out 2>&1; tail -1 gpurun_out/recomb_y.out
C21CM_RECOMB_FUSED_NREC=0 python tools/time_recomb.py 512 3 inhomogeneous_filtered_xe 2>&1 | tail -1
