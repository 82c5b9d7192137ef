timeout 600 python -m pytest tests/test_gpu_ionize.py -m gpu -x -q -k "xe_grid or banded" 2>&1 | grep -v '^band r' | tail -4
for z in 22.0 20.0 18.0 15.0 12.0 9.0 7.0; do
for b in 1 0; do echo "z=$z band=$b"; C21CM_EUL_BAND=$b PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 $z 1 2>/dev/null | tail -1 | cut -c60-200; done
C21CM_EUL_BAND_DEBUG=1 PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 $z 1 2>&1 | grep -cE "band miss" ; true
done
python bench.py --mode erfc --steps 10 --warmup 3 --no-cpu-baseline --no-abi 2>/dev/null > gpurun_out/r04_bench_erfc.json; python -c "import json;d=json.load(open('gpurun_out/r04_bench_erfc.json'));print('erfc', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:60])"
for src in 1 0; do PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 $src 9.0 0 2>/dev/null | tail -1 | cut -c60-200; done
