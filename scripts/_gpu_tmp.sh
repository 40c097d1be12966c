timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 200 python scripts/lu_sweep.py 4096 8192 16384
