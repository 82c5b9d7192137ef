python -m pytest tests/test_gpu_ts.py tests/test_gpu_ts_shard.py tests/test_gpu_shard_shim.py -q -x -k "ts" 2>&1 | tail -4
PYTHONPATH=. python tools/time_ts_shard_pieces.py 512 8 16 2>/dev/null | tail -1
PYTHONPATH=. python tools/time_ts_shard_pieces.py 512 8 64 2>/dev/null | tail -1
