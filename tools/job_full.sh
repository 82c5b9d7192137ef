python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_r04_final.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > gpurun_out/bench_default.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/bench_default.json'));print('default', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_profile_stale'])"
