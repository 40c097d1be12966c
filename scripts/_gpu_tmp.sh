timeout 500 python -m pytest tests -m gpu -q -x -k "matmul or syrk or lu or mldivide or dgemm" 2>&1 | tail -4
timeout 300 python scripts/gemm_trans.py 2>&1 | tail -5
