#!/bin/bash
export TMPDIR=/tmp
stress() { # label, env...
  lab=$1; shift
  for p in 1 2 3 4; do ( env "$@" PYTHONPATH=. REPS=8 timeout 1200 python tools/scratch/diag_1024_race.py > gpurun_out/r05k_${lab}_$p.txt 2>&1 ) & done
  wait
  bad=$(cat gpurun_out/r05k_${lab}_*.txt | grep -c "NOT reproducible\|not reproducible\|[1-9][0-9]* cells differ")
  ok=$(cat gpurun_out/r05k_${lab}_*.txt | grep -c " 0 cells differ")
  echo "== $lab: anomalies $bad, clean sharded-vs-single comparisons $ok"
  cat gpurun_out/r05k_${lab}_*.txt | grep "NOT reproducible\|not reproducible\|[1-9][0-9]* cells differ" | cut -c1-200 | head -6
}
stress default A=1
stress serialize AMD_SERIALIZE_KERNEL=3
stress nosdma HSA_ENABLE_SDMA=0
stress n512 N=512 REPS=40
stress nopk C21CM_LIB=variants/nopk/lib21cmfast_hip.so
