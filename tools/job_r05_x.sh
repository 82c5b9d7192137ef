#!/bin/bash
# kernel breakdown of the Eulerian table modes through the ABI (CONST-ION-EFF with tables, E-INTEGRAL)
REPO=$PWD
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
for m in 0 1; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_eul_$m -o q -- \
      python $REPO/tools/time_abi_ionize.py 512 $m > $REPO/gpurun_out/prof_eul_$m.out 2> $REPO/gpurun_out/prof_eul_$m.err
  cd $REPO
  echo "== source model $m"; tail -1 gpurun_out/prof_eul_$m.out | cut -c1-400
  timeout 20 python tools/kernel_stats_brief.py gpurun_out/prof_eul_$m/q_kernel_stats.csv 12
done
