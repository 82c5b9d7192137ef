#!/bin/bash
# full GPU suite with the placement walk (xdist, then the ionize / shard tests serially)
python -m pytest tests -x -q -m gpu -n 6 > gpurun_out/full_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/full_tests.out | tail -5
python -m pytest tests/test_gpu_ionize.py tests/test_gpu_abi.py tests/test_gpu_bench_shard.py -x -q -m gpu > gpurun_out/serial_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/serial_tests.out | tail -4
