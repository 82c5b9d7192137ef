#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r05f_ab.txt
run() { echo "$1" >> gpurun_out/r05f_ab.txt; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05f_ab.txt; }
run "pk (in-tree)" A=1
run "noslp" C21CM_LIB=variants/noslp/lib21cmfast_hip.so
run "pk (in-tree)" A=1
run "noslp" C21CM_LIB=variants/noslp/lib21cmfast_hip.so
echo "1024:" >> gpurun_out/r05f_ab.txt
for e in "A=1" "C21CM_LIB=variants/noslp/lib21cmfast_hip.so"; do echo "$e" >> gpurun_out/r05f_ab.txt; env $e timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05f_ab.txt; done
echo "erfc:" >> gpurun_out/r05f_ab.txt
for e in "A=1" "C21CM_LIB=variants/noslp/lib21cmfast_hip.so"; do echo "$e" >> gpurun_out/r05f_ab.txt; env $e timeout 300 python bench.py --mode erfc --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05f_ab.txt; done
cat gpurun_out/r05f_ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05f_tests.log 2>&1
tail -12 gpurun_out/r05f_tests.log
for acc in fixed double; do echo "CIC_ACC=$acc"; C21CM_CIC_ACC=$acc PYTHONPATH=. timeout 300 python tools/time_cic.py 2>&1 | tail -1 | cut -c1-700; done > gpurun_out/r05f_cic.txt 2>&1
cat gpurun_out/r05f_cic.txt
for src in 1 0; do PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1; done > gpurun_out/r05f_abi_eulerian.jsonl
PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 >> gpurun_out/r05f_abi_eulerian.jsonl
cat gpurun_out/r05f_abi_eulerian.jsonl
