#!/bin/bash
# recombination loop: three-line barrier kernel at two waves per SIMD (N_rec rows parked in LDS, 16-byte mask
# rows), Gamma_12 pass with 16-byte mask rows, x_e + whalo_sfr as one two-grid sweep
python -m pytest tests/test_gpu_recomb.py tests/test_gpu_reference_fixtures.py tests/test_gpu_ionize.py tests/test_gpu_abi.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -5
python tools/time_recomb.py 512 3 > gpurun_out/recomb_y.out 2>&1; tail -1 gpurun_out/recomb_y.out
