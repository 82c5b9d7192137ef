#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05d_tests.log 2>&1
tail -12 gpurun_out/r05d_tests.log
tools/bin/valu_rate_probe > gpurun_out/r05_valu_rate_probe.txt 2>&1
cat gpurun_out/r05_valu_rate_probe.txt
rm -f gpurun_out/r05d_ab.txt
run() { echo "$1" >> gpurun_out/r05d_ab.txt; shift; env "$@" timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05d_ab.txt; }
run "1024 default (X rot, Y rot)" A=1
run "1024 X plain, Y rot" C21CM_XORDER=0
run "1024 X rot, Y plain" C21CM_YORDER=0
run "1024 both plain" C21CM_XORDER=0 C21CM_YORDER=0
echo "256:" >> gpurun_out/r05d_ab.txt
for e in "A=1" "C21CM_XORDER=0 C21CM_YORDER=0"; do echo "$e" >> gpurun_out/r05d_ab.txt; env $e timeout 300 python bench.py --hii-dim 256 --steps 20 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py >> gpurun_out/r05d_ab.txt; done
cat gpurun_out/r05d_ab.txt
for acc in fixed double; do echo "CIC_ACC=$acc"; C21CM_CIC_ACC=$acc PYTHONPATH=. timeout 300 python tools/time_cic.py 2>&1 | tail -3; done > gpurun_out/r05d_cic.txt 2>&1
cat gpurun_out/r05d_cic.txt
for src in 1 0; do PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1; done > gpurun_out/r05d_abi_eulerian.jsonl
PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 >> gpurun_out/r05d_abi_eulerian.jsonl
cat gpurun_out/r05d_abi_eulerian.jsonl
