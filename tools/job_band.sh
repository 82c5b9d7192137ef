timeout 600 python -m pytest tests/test_gpu_ionize.py -m gpu -x -q -k "xe_grid or banded" 2>&1 | grep -v '^band r' | tail -6
for q in 1 0; do echo "quad=$q"
C21CM_EUL_BAND_QUAD=$q PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>/dev/null | tail -1
C21CM_EUL_BAND_QUAD=$q PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 0 2>/dev/null | tail -1
C21CM_EUL_BAND_QUAD=$q python bench.py --mode erfc --steps 10 --warmup 3 --no-cpu-baseline --no-abi --no-kernel-roofline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('erfc', d['ms_per_step'])"
done
C21CM_EUL_BAND_DEBUG=1 PYTHONPATH=. timeout 200 python tools/time_abi_ionize.py 512 1 9.0 1 2>&1 | grep -E 'fail|band r= *(1|2|3|20|36) ' | head -6
