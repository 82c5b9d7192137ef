#!/bin/bash
# the fused recombination shard phases with an x_e grid
python -m pytest tests/test_gpu_recomb.py tests/test_gpu_ts_shard.py tests/test_gpu_config5.py -x -q -m gpu -n 4 > gpurun_out/recomb_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/recomb_tests.out | tail -8
python tools/time_recomb_shard.py 2>&1 | tail -3
