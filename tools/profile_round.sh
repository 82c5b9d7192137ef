#!/bin/bash
# Round profile set, run on the GPU box:  gpurun --timeout 1800 -- 'bash tools/profile_round.sh r01'
# Produces under gpurun_out/: the rocprofv3 kernel-trace stats of the default bench command, the
# two PMC passes (FETCH_SIZE / WRITE_SIZE, separately, kernel-trace only) reduced by
# tools/collect_pmc.py, and a plain bench line.  Copy the summaries into profiles/.
set -x
TAG=${1:-rNN}
REPO=$PWD
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -o $TAG -- \
    python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
    > $REPO/gpurun_out/prof_stats_bench.json 2> $REPO/gpurun_out/prof_stats.err
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$C -o pmc -- \
        python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline \
        > /dev/null 2> $REPO/gpurun_out/pmc_$C.err
done
cd $REPO
F=$(dirname $(find gpurun_out/pmc_FETCH_SIZE -name pmc_counter_collection.csv | head -1))
W=$(dirname $(find gpurun_out/pmc_WRITE_SIZE -name pmc_counter_collection.csv | head -1))
python tools/collect_pmc.py $F $W gpurun_out/pmc_$TAG.json gpurun_out/${TAG}_pmc
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +2M -delete
cp gpurun_out/pmc_$TAG.json profiles/   # (on the GPU box: the bench line below reads the counters of THIS run's kernels)
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_final.err
