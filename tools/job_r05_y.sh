#!/bin/bash
python -m pytest tests/test_gpu_placement.py -x -q -m gpu 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -5
