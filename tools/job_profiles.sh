#!/bin/bash
# Round-end evidence set (GPU box): gpurun --timeout 2400 -- 'bash tools/job_profiles.sh r05'
TAG=${1:-r05}
REPO=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/bin/copy_bench > gpurun_out/${TAG}_copy_bench.txt 2>&1
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
tools/bin/copy_bench >> gpurun_out/${TAG}_copy_bench.txt 2>&1
# G = 1 closed form: bench line + kernel stats
timeout 300 python bench.py --mode erfc > gpurun_out/${TAG}_bench_erfc.json 2> gpurun_out/bench_erfc.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_erfc -o erfc -- \
    python $REPO/bench.py --mode erfc --steps 5 --warmup 2 --no-cpu-baseline --no-abi > /dev/null 2>&1)
# config 2: IC + PerturbedField
timeout 300 python bench.py --mode icpf > gpurun_out/${TAG}_bench_icpf.json 2> gpurun_out/bench_icpf.err
# config 4 on one GPU: kernel stats, the two PMC passes, then the bench line (reads the PMC summary)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_1024 -o k1024 -- \
    python $REPO/bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline > /dev/null 2>&1)
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc1024_$C -o pmc -- \
        python $REPO/bench.py --hii-dim 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline > /dev/null 2> $REPO/gpurun_out/pmc1024_$C.err)
done
F=$(dirname $(find gpurun_out/pmc1024_FETCH_SIZE -name pmc_counter_collection.csv | head -1))
W=$(dirname $(find gpurun_out/pmc1024_WRITE_SIZE -name pmc_counter_collection.csv | head -1))
python tools/collect_pmc.py $F $W gpurun_out/pmc1024_$TAG.json gpurun_out/${TAG}_1024_pmc 1024 > gpurun_out/pmc1024.log 2>&1
cp gpurun_out/pmc1024_$TAG.json profiles/ 2>/dev/null   # (the bench line below reads profiles/pmc1024_*.json)
find gpurun_out/pmc1024_FETCH_SIZE gpurun_out/pmc1024_WRITE_SIZE -name "*.csv" -size +2M -delete
timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_1024.json 2> gpurun_out/bench_1024.err
# the sharded code path on one rank (RCCL communicator of one), incl. the config4 object
timeout 600 python bench.py --force-shard --steps 5 --warmup 2 --no-cpu-baseline --no-abi --config4-dim 1024 --config4-steps 2 > gpurun_out/bench_${TAG}_force_shard_one_rank.json 2> gpurun_out/bench_force_shard.err
# measured pieces of the multi-GPU budgets
PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 1024 8 2>/dev/null | tail -1 > gpurun_out/${TAG}_slab_finish_pieces_1024x8.json
PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 512 8 2>/dev/null | tail -1 > gpurun_out/${TAG}_slab_finish_pieces_512x8.json
# Eulerian models through the drop-in entry point, banded and dense
for b in 1 0; do for src in 1 0; do
  C21CM_EUL_BAND=$b PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1 | sed "s/^{/{\"banded\": $b, /"
done; done > gpurun_out/${TAG}_abi_eulerian.jsonl
PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 >> gpurun_out/${TAG}_abi_eulerian.jsonl
timeout 300 python tools/time_recomb.py 512 2>/dev/null | tail -1 > gpurun_out/${TAG}_recomb_timing.json
timeout 300 python tools/time_cic.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_cic_timing.json
timeout 600 python tools/time_coeval_ts.py 512 1024 6.0 2> gpurun_out/config5.err | tail -1 > gpurun_out/${TAG}_config5_timing.json
tools/bin/valu_rate_probe > gpurun_out/${TAG}_valu_rate_probe.txt 2>&1
ls -la gpurun_out | tail -40
