#!/bin/bash
# kernel breakdown of the recombination loops (which launches separate 124 ms from 2 x 44)
REPO=$PWD
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
for v in inhomogeneous_cell_xe inhomogeneous_cell homogeneous; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_rec_$v -o q -- \
      python $REPO/tools/time_recomb.py 512 3 $v > $REPO/gpurun_out/prof_rec_$v.out 2> $REPO/gpurun_out/prof_rec_$v.err
  cd $REPO
  echo "== $v"; tail -2 gpurun_out/prof_rec_$v.out
  timeout 20 python tools/kernel_stats_brief.py gpurun_out/prof_rec_$v/q_kernel_stats.csv 16
done
