#!/bin/bash
export TMPDIR=/tmp
echo "== in-tree (fixed)"
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_contention.py -x -q -m gpu 2>&1 | tail -1; done
echo "== variant with the barrier removed"
for i in 1 2 3; do C21CM_LIB=variants/notwbar/lib21cmfast_hip.so timeout 900 python -m pytest tests/test_gpu_contention.py -x -q -m gpu 2>&1 | tail -1; done
