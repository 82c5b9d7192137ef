python -m pytest tests/test_gpu_ionize.py -m gpu -x -q -k "closed_form or erfc or const_ion" 2>&1 | tail -5
python bench.py --mode erfc --no-cpu-baseline --no-abi > gpurun_out/erfc_band.json 2> gpurun_out/erfc_band.err; python -c "import json;d=json.load(open('gpurun_out/erfc_band.json'));print(d['ms_per_step'], d['roofline']['r_loop'])"
bash tools/prof_stats_only.sh --mode erfc --steps 5 --warmup 2 --no-abi 2>&1 | head -6
C21CM_EUL_BAND_DEBUG=1 python bench.py --mode erfc --steps 1 --warmup 0 --no-cpu-baseline --no-abi 2> gpurun_out/band_debug.log > /dev/null; grep 'fail' gpurun_out/band_debug.log|head -3; head -8 gpurun_out/band_debug.log
