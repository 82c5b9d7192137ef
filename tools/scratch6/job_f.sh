export PYTHONPATH=.
for v in "default" "default" "C21CM_WS_PLACE=0"; do
  echo "== $v"
  env $( [ "$v" != default ] && echo $v ) C21CM_WS_TRACE=1 python tools/time_coeval_ts.py 512 1024 6.0 2> gpurun_out/c5.err | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print({k:d[k] for k in ('n_snapshots','evolution_s','ts_ms','ionize_ms','perturb_ms')})"
  grep -c "\[place\]" gpurun_out/c5.err
done
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-abi --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), round(d['first_call_ms'],1), d['placement'])"; done
python bench.py --hii-dim 1024 --no-cpu-baseline --no-abi --steps 3 --warmup 1 --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), round(d['first_call_ms'],1), d['placement'])"
