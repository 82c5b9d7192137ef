#!/bin/bash
# First GPU job of round 5: baseline bench line, the copy-rate ceiling from the same box, kernel stats.
TAG=${1:-r05a}
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/bin/copy_bench > gpurun_out/${TAG}_copy_bench.txt 2>&1
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tools/bin/copy_bench >> gpurun_out/${TAG}_copy_bench.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG} -o $TAG -- \
    python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi > $REPO/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null)
python tools/kernel_stats_brief.py $(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1) > gpurun_out/${TAG}_kernel_stats_brief.txt 2>&1
timeout 300 python bench.py --mode erfc --no-cpu-baseline > gpurun_out/${TAG}_bench_erfc.json 2> gpurun_out/${TAG}_bench_erfc.err
timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_1024.json 2> gpurun_out/${TAG}_bench_1024.err
for src in 1 0; do
  PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1
done > gpurun_out/${TAG}_abi_eulerian.jsonl
ls -la gpurun_out | tail -20
cat gpurun_out/${TAG}_copy_bench.txt
cat gpurun_out/${TAG}_bench.json
