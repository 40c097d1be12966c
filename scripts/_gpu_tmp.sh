timeout 300 python -m pytest tests -m gpu -q -x -k "lu or mldivide or linsolve or blk or cyclic" 2>&1 | tail -3
timeout 200 python scripts/solve_time.py 2>&1 | tail -5
