#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05u_tests.log 2>&1
tail -8 gpurun_out/r05u_tests.log
