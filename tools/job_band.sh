timeout 600 python -m pytest tests/test_gpu_ionize.py -m gpu -x -q -k "banded" 2>&1 | grep -v '^band r' | tail -40
