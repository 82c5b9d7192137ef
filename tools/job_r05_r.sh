#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_fixtures.py -x -q -m gpu -k "deviates or reference_stream" 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_final.err
timeout 300 python bench.py --hii-dim 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r05_1024.json 2> gpurun_out/bench_1024.err
python tools/bench_brief.py gpurun_out/bench_r05_final.json; python tools/bench_brief.py gpurun_out/bench_r05_1024.json
python - <<'PY'
import json
for f in ("gpurun_out/bench_r05_final.json","gpurun_out/bench_r05_1024.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, r.get("traffic"), r.get("traffic_profile_stale"), r.get("achievable_GBs"))
PY
