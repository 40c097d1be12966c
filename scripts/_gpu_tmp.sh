timeout 400 python -m pytest tests -m gpu -q -x -k "lu or mldivide or linsolve or blk or trsm or cyclic" 2>&1 | tail -6
timeout 100 python scripts/lu_time.py 2048 8192 16384 2>&1 | grep "rep=1"
