#!/bin/bash
# reproduce the one-off sharded != single mismatch of the 1024^3 test under GPU contention
export TMPDIR=/tmp
mkdir -p gpurun_out
stress() { # label, env
  echo "== $1"; shift
  for round in 1 2 3; do
    for p in 1 2 3 4; do
      ( env "$@" timeout 900 python -m pytest tests/test_gpu_ionize.py -x -q -m gpu -k config4_1024 2>&1 | tail -1 ) &
    done
    wait
  done
}
stress "default" A=1
stress "plain order" C21CM_XORDER=0 C21CM_YORDER=0
stress "nopk variant" C21CM_LIB=variants/nopk/lib21cmfast_hip.so C21CM_XORDER=0 C21CM_YORDER=0
