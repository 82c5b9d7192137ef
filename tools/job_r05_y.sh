#!/bin/bash
# Gamma_12 inside the recombination barrier kernel (only waves whose lines crossed transform whalo_sfr): tests + A/B
python -m pytest tests/test_gpu_recomb.py tests/test_gpu_reference_fixtures.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -6
for f in 1 0; do
  echo "== C21CM_RECOMB_G12_FUSED=$f"
  C21CM_RECOMB_G12_FUSED=$f python tools/time_recomb.py 512 3 > gpurun_out/recomb_g12_$f.out 2>&1; tail -1 gpurun_out/recomb_g12_$f.out
done
