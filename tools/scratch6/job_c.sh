python -m pytest tests/test_gpu_ionize.py -q -x -k "table or banded or eulerian or closed_form or erfc or slab_finish or shard" 2>&1 | tail -15
python -m pytest tests/test_gpu_reference_fixtures.py tests/test_gpu_abi.py tests/test_gpu_golden.py tests/test_gpu_run_coeval.py -q -x 2>&1 | tail -5
for m in 1 0; do for f in "1 1" "1 0" "0 0"; do set -- $f; echo "model $m fused $1 pair $2"; C21CM_EUL_TABLE_FUSED=$1 C21CM_EUL_PAIR=$2 python tools/time_abi_ionize.py 512 $m 2>&1 | tail -1; done; done
for p in 1 0; do echo "erfc pair $p"; C21CM_EUL_PAIR=$p python bench.py --mode erfc --no-cpu-baseline --no-abi --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['config']['global_xH'])"; done
