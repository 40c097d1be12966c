timeout 300 python -m pytest tests -m gpu -q -x -k "lu or mldivide or linsolve or blk or cyclic" 2>&1 | tail -6
echo "== rows 256 (P=1)"; RMHIP_LU_PANEL_DEBUG=1 timeout 100 python scripts/panel_time.py 256 2>&1 | tail -9
echo "== rows 16384 (P=64)"; RMHIP_LU_PANEL_DEBUG=1 timeout 100 python scripts/panel_time.py 16384 2>&1 | tail -9
timeout 100 python scripts/lu_time.py 2048 8192 16384 2>&1 | grep "rep=1"
