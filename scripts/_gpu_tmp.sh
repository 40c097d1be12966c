timeout 300 python -m pytest tests -m gpu -q -x -k "lu or mldivide or linsolve or blk or cyclic" 2>&1 | tail -3
timeout 200 python scripts/trsm_time.py 2>&1 | grep "w=128\|w=64 "
timeout 100 python scripts/lu_time.py 16384 2>&1 | grep "rep=1"
