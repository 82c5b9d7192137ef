#!/bin/bash
# Round-end evidence set (GPU box): gpurun --timeout 1500 -- 'bash tools/job_profiles.sh r04'
TAG=${1:-r04}
REPO=$PWD
export TMPDIR=/tmp
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
# G = 1 closed form: bench line + kernel stats
timeout 300 python bench.py --mode erfc > gpurun_out/${TAG}_bench_erfc.json 2> gpurun_out/bench_erfc.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_erfc -o erfc -- \
    python $REPO/bench.py --mode erfc --steps 5 --warmup 2 --no-cpu-baseline --no-abi > /dev/null 2>&1)
# config 4 on one GPU: bench line + kernel stats
timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_1024.json 2> gpurun_out/bench_1024.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_1024 -o k1024 -- \
    python $REPO/bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline > /dev/null 2>&1)
# Eulerian models through the drop-in entry point, banded and dense
for b in 1 0; do for src in 1 0; do
  C21CM_EUL_BAND=$b PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1 | sed "s/^{/{\"banded\": $b, /"
done; done > gpurun_out/${TAG}_abi_eulerian.jsonl
timeout 300 python tools/time_recomb.py 512 2>/dev/null | tail -1 > gpurun_out/${TAG}_recomb_timing.json
timeout 600 python tools/time_coeval_ts.py 512 1024 6.0 2> gpurun_out/config5.err | tail -1 > gpurun_out/${TAG}_config5_timing.json
ls -la gpurun_out | tail -30
