#!/bin/bash
# Round evidence set (GPU box):  gpurun --timeout 3000 -- 'bash tools/job_profiles.sh r06'
# Everything lands in gpurun_out/profiles_<TAG>/ with the names profiles/ uses; copy what is to be judged.
# Sections can be selected: bash tools/job_profiles.sh r06 "bench pmc icpf ts sizes eul shard misc"
TAG=${1:-r06}
WHAT=${2:-"bench pmc icpf ts sizes eul shard misc"}
REPO=$PWD
OUT=$REPO/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
has() { [[ " $WHAT " == *" $1 "* ]]; }
stats() { # stats <name> <command...>: rocprofv3 kernel-trace stats of a command -> $OUT/<name>_kernel_stats.csv
    local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$name -o k -- "$@" \
        > $REPO/gpurun_out/prof_$name.out 2> $REPO/gpurun_out/prof_$name.err)
    local f=$(find $REPO/gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${TAG}_${name}_kernel_stats.csv && python tools/kernel_stats_brief.py $f 12
    find $REPO/gpurun_out/prof_$name -name "*.csv" -size +2M -delete
}
pmc() { # pmc <name> <command...>: the two counter passes, separately, kernel-trace only
    local name=$1; shift
    for C in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${name}_$C -o pmc -- "$@" \
            > /dev/null 2> $REPO/gpurun_out/pmc_${name}_$C.err)
    done
    PMC_F=$(dirname $(find gpurun_out/pmc_${name}_FETCH_SIZE -name pmc_counter_collection.csv | head -1))
    PMC_W=$(dirname $(find gpurun_out/pmc_${name}_WRITE_SIZE -name pmc_counter_collection.csv | head -1))
}
pmc_clean() { find gpurun_out/pmc_$1_FETCH_SIZE gpurun_out/pmc_$1_WRITE_SIZE -name "*.csv" -size +2M -delete; }

if has bench; then
    tools/bin/copy_bench > $OUT/${TAG}_copy_bench.txt 2>&1
    stats bench python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline
    cp gpurun_out/prof_bench.out $OUT/${TAG}_bench_under_rocprof.json
    mv $OUT/${TAG}_bench_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
fi
if has pmc; then
    pmc 512 python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-abi
    python tools/collect_pmc.py $PMC_F $PMC_W $OUT/pmc_$TAG.json $OUT/${TAG}_pmc
    pmc_clean 512
    cp $OUT/pmc_$TAG.json profiles/   # (on the GPU box only: the bench lines below read THIS run's counters)
fi
if has icpf; then
    stats icpf python $REPO/bench.py --mode icpf --steps 5 --warmup 2 --no-cpu-baseline
    pmc icpf python $REPO/bench.py --mode icpf --steps 2 --warmup 1 --no-cpu-baseline
    python tools/collect_pmc.py $PMC_F $PMC_W $OUT/pmc_icpf_$TAG.json $OUT/${TAG}_icpf_pmc total:3
    pmc_clean icpf
    cp $OUT/pmc_icpf_$TAG.json profiles/
    timeout 300 python bench.py --mode icpf > $OUT/${TAG}_bench_icpf.json 2> gpurun_out/bench_icpf.err
fi
if has bench; then
    python bench.py --steps 10 --warmup 3 > $OUT/bench_${TAG}_final.json 2> gpurun_out/bench_final.err
    timeout 300 python bench.py --mode erfc > $OUT/${TAG}_bench_erfc.json 2> gpurun_out/bench_erfc.err
    stats erfc python $REPO/bench.py --mode erfc --steps 5 --warmup 2 --no-cpu-baseline --no-abi
fi
if has ts; then
    # per-kernel times of a spin-temperature evolution at config 5's grid, and the run's own timing line
    C21CM_IC_RNG=philox stats ts env PYTHONPATH=$REPO python $REPO/tools/time_coeval_ts.py 512 1024 20 1.02 16
    timeout 900 python tools/time_coeval_ts.py 512 1024 6.0 2> gpurun_out/config5.err | tail -1 > $OUT/${TAG}_config5_timing.json
    timeout 300 python tools/time_ts.py 512 2>/dev/null | tail -1 > $OUT/${TAG}_ts_timing.json
fi
if has sizes; then
    # north_star's 256^3 -> 1024^3 sweep on one GPU
    timeout 300 python bench.py --hii-dim 256 --no-cpu-baseline --no-abi > $OUT/bench_${TAG}_256.json 2> gpurun_out/bench_256.err
    stats 1024 python $REPO/bench.py --hii-dim 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline
    pmc 1024 python $REPO/bench.py --hii-dim 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-abi --no-kernel-roofline
    python tools/collect_pmc.py $PMC_F $PMC_W $OUT/pmc1024_$TAG.json $OUT/${TAG}_1024_pmc 1024
    pmc_clean 1024
    cp $OUT/pmc1024_$TAG.json profiles/
    # (with the CPU leg: the oracle runs the 1024^3 box on the same fields -> parity_1024 in the line; ~3 min on 64 threads)
    timeout 2000 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-abi > $OUT/bench_${TAG}_1024.json 2> gpurun_out/bench_1024.err
fi
if has eul; then
    # Eulerian models through the drop-in entry point: fused table sweep + pair sweeps (default), round 5's loop
    for v in "1 1" "0 0"; do set -- $v
      for src in 1 0; do
        C21CM_EUL_TABLE_FUSED=$1 C21CM_EUL_PAIR=$2 PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 2>/dev/null | tail -1 | sed "s/^{/{\"table_sweep_fused\": $1, \"pair_sweeps\": $2, /"
      done
      C21CM_EUL_TABLE_FUSED=$1 C21CM_EUL_PAIR=$2 PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1 | sed "s/^{/{\"table_sweep_fused\": $1, \"pair_sweeps\": $2, /"
    done > $OUT/${TAG}_abi_eulerian.jsonl
    stats abi_e_integral env PYTHONPATH=$REPO python $REPO/tools/time_abi_ionize.py 512 1 9.0
    stats abi_e_integral_xe env PYTHONPATH=$REPO python $REPO/tools/time_abi_ionize.py 512 1 9.0 1
fi
if has shard; then
    # the sharded code path: one rank over real RCCL; 2, 3 and 8 ranks on the one GPU over the test transport
    timeout 600 python bench.py --force-shard --steps 5 --warmup 2 --no-cpu-baseline --no-abi --config4-dim 1024 --config4-steps 2 \
        > $OUT/bench_${TAG}_force_shard_one_rank.json 2> gpurun_out/bench_force_shard.err
    make -C tests/shim > /dev/null
    for N in 2 3 8; do
      C21CM_RCCL_LIB=$REPO/tests/shim/librccl_shim.so RCCL_SHIM_SLOT_KB=256 C21CM_WS_PLACE=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
        --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --backend gloo --steps 3 --warmup 1 --no-cpu-baseline \
        --no-abi --no-kernel-roofline --config4-dim 256 2> gpurun_out/bench_shim_$N.err | grep "^{" > $OUT/bench_${TAG}_shim_transport_${N}_ranks_one_gpu.json
    done
    PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 1024 8 2>/dev/null | tail -1 > $OUT/${TAG}_slab_finish_pieces_1024x8.json
    PYTHONPATH=. timeout 600 python tools/time_slab_finish.py 512 8 2>/dev/null | tail -1 > $OUT/${TAG}_slab_finish_pieces_512x8.json
fi
if has misc; then
    timeout 300 python tools/time_recomb.py 512 2>/dev/null | tail -1 > $OUT/${TAG}_recomb_timing.json
    timeout 300 python tools/time_cic.py 2>/dev/null | tail -1 > $OUT/${TAG}_cic_timing.json
fi
ls -la $OUT
