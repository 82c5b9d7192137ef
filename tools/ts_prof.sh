#!/bin/bash
# Per-kernel times of a spin-temperature evolution at config 5's grid (HII_DIM 512, DIM 1024):
# rocprofv3 kernel trace of tools/time_coeval_ts.py (Philox ICs so that the profile is the device's).
# usage (GPU box): tools/ts_prof.sh [tag] [z_end]   -> gpurun_out/prof_ts_<tag>/, brief on stdout
tag=${1:-r03}
zend=${2:-20}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
C21CM_IC_RNG=philox timeout 600 rocprofv3 --kernel-trace --stats --output-format csv \
  -d $root/gpurun_out/prof_ts_$tag -o ts -- env PYTHONPATH=$root python $root/tools/time_coeval_ts.py 512 1024 $zend 1.02 16 \
  > $root/gpurun_out/prof_ts_$tag.json 2> $root/gpurun_out/prof_ts_$tag.err
cd $root
python tools/kernel_stats_brief.py $(find gpurun_out/prof_ts_$tag -name "*kernel_stats.csv" | head -1) 24
