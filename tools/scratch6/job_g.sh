python -m pytest tests/test_gpu_placement.py -q -x 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-abi 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), round(d['first_call_ms'],1), d['placement'], d['default_allocation'], d['config']['work_spectra_placement'][:60])"
PYTHONPATH=. python tools/time_coeval_ts.py 512 1024 6.0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print({k:d[k] for k in ('n_snapshots','evolution_s','ts_ms','ionize_ms','perturb_ms')})"
