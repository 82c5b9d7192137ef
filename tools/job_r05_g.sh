#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
REPO=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu -n 4 > gpurun_out/r05g_tests.log 2>&1
tail -8 gpurun_out/r05g_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null | python tools/bench_brief.py
PYTHONPATH=. timeout 300 python tools/time_ic_pf.py 2>/dev/null | tail -3 > gpurun_out/r05g_icpf.txt; cat gpurun_out/r05g_icpf.txt
PYTHONPATH=. timeout 300 python tools/time_recomb.py 512 2>/dev/null | tail -1 > gpurun_out/r05g_recomb_timing.json; cat gpurun_out/r05g_recomb_timing.json
PYTHONPATH=. timeout 600 python tools/time_coeval_ts.py 512 1024 6.0 2>/dev/null | tail -1 > gpurun_out/r05g_config5_timing.json; cat gpurun_out/r05g_config5_timing.json | cut -c1-1500
(cd /tmp && PYTHONPATH=$REPO timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_abi1 -o abi1 -- python $REPO/tools/time_abi_ionize.py 512 1 9.0 > /dev/null 2>&1)
python tools/kernel_stats_brief.py $(find gpurun_out/prof_abi1 -name "*kernel_stats.csv" | head -1) 12
