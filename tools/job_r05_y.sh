#!/bin/bash
# the fused recombination shard phases with a filtered N_rec (CELL_RECOMB = false), with and without x_e
python -m pytest tests/test_gpu_recomb.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -12
