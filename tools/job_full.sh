python -m pytest tests -m gpu -q -x 2>&1 | tail -8
