#!/bin/bash
# spin-temperature accumulate kernel: scalar lookup (default now) against the packed 2-vector form (variants/tspk)
python -m pytest tests/test_gpu_ts.py tests/test_gpu_ts_shard.py tests/test_gpu_config5.py -x -q -m gpu 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3
REPO=$PWD; export TMPDIR=/tmp
for v in default tspk; do
  lib=$REPO/variants/$v/lib21cmfast_hip.so; [ $v = default ] && lib=$REPO/21cmfast_amd/lib21cmfast_hip.so
  cd /tmp
  C21CM_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_ts_$v -o q -- \
      python $REPO/tools/time_coeval_ts.py 512 1024 12.0 > $REPO/gpurun_out/prof_ts_$v.out 2>/dev/null
  cd $REPO
  echo "== $v"; tail -1 gpurun_out/prof_ts_$v.out | cut -c1-230
  python tools/kernel_stats_brief.py gpurun_out/prof_ts_$v/q_kernel_stats.csv 40 | grep "ts_accumulate\|sfrd_sum\|ts_cell"
done
