#!/bin/bash
# full GPU suite under xdist after the recombination / reverse pass Z changes, then config 2 and config 5 timings
python -m pytest tests -x -q -m gpu -n 6 > gpurun_out/full_tests.out 2>&1
echo "rc=$?"; grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" gpurun_out/full_tests.out | tail -6
python bench.py --mode icpf --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-600
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
