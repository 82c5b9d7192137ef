#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ionize.py -x -q -m gpu -k config4_1024 2>&1 | tail -2; done
echo "--- plain order"
for i in 1 2; do C21CM_XORDER=0 C21CM_YORDER=0 timeout 600 python -m pytest tests/test_gpu_ionize.py -x -q -m gpu -k config4_1024 2>&1 | tail -2; done
echo "--- nopk variant (SLP on, scalar source)"
for i in 1 2; do C21CM_LIB=variants/nopk/lib21cmfast_hip.so timeout 600 python -m pytest tests/test_gpu_ionize.py -x -q -m gpu -k config4_1024 2>&1 | tail -2; done
PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 1024 8 2>&1 | tail -1 > gpurun_out/r05h_slab_1024.json; cat gpurun_out/r05h_slab_1024.json
PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 512 8 2>&1 | tail -1 > gpurun_out/r05h_slab_512.json; cat gpurun_out/r05h_slab_512.json
timeout 300 python bench.py --mode icpf 2>/dev/null > gpurun_out/r05h_bench_icpf.json; cat gpurun_out/r05h_bench_icpf.json
