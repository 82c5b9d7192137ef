#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
stress() { # label, env...
  lab=$1; shift
  for p in 1 2 3 4; do ( env "$@" PYTHONPATH=. REPS=10 timeout 1200 python tools/scratch/diag_1024_race.py > gpurun_out/r05l_${lab}_$p.txt 2>&1 ) & done
  wait
  bad=$(cat gpurun_out/r05l_${lab}_*.txt | grep -c "NOT reproducible\|not reproducible\|[1-9][0-9]* cells differ")
  ok=$(cat gpurun_out/r05l_${lab}_*.txt | grep -c " 0 cells differ")
  echo "== $lab: anomalies $bad, clean sharded-vs-single comparisons $ok"
  cat gpurun_out/r05l_${lab}_*.txt | grep "NOT reproducible\|not reproducible\|[1-9][0-9]* cells differ" | cut -c1-200 | head -6
}
stress fixed A=1
stress fixed_nosdma HSA_ENABLE_SDMA=0
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05l_tests.log 2>&1
tail -6 gpurun_out/r05l_tests.log
echo "== plane_yz re-timed (experimental variant)"
C21CM_LIB=variants/exp/lib21cmfast_hip.so C21CM_YZ=2 PYTHONPATH=. timeout 300 python tools/time_yz.py 2>/dev/null | tail -1 | cut -c1-600
for e in "A=1" "C21CM_YZ=2" "C21CM_YZ=1"; do echo "$e"; env $e C21CM_LIB=variants/exp/lib21cmfast_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py; done
