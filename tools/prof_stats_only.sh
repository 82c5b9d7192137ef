#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command only (no PMC passes)
REPO=$PWD
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_quick -o q -- \
    python $REPO/bench.py --no-cpu-baseline "$@" > $REPO/gpurun_out/prof_quick_bench.json 2> $REPO/gpurun_out/prof_quick.err
cd $REPO
timeout 20 python tools/kernel_stats_brief.py gpurun_out/prof_quick/q_kernel_stats.csv 14
