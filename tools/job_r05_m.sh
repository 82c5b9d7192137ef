#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
REPO=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05m_tests.log 2>&1
tail -5 gpurun_out/r05m_tests.log
for d in 1 0; do for src in 1 0; do
  C21CM_EUL_DENSE_ROWS=$d PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1 | sed "s/^{/{\"dense_rows\": $d, /"
done; done > gpurun_out/r05m_abi_eulerian.jsonl
PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 >> gpurun_out/r05m_abi_eulerian.jsonl
cat gpurun_out/r05m_abi_eulerian.jsonl
(cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_abi1m -o abi1 -- python $REPO/tools/time_abi_ionize.py 512 1 9.0 > /dev/null 2>&1)
python tools/kernel_stats_brief.py $(find gpurun_out/prof_abi1m -name "*kernel_stats.csv" | head -1) 8
