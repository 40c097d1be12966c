timeout 300 python -m pytest tests -m gpu -q -x -k "lu or mldivide or linsolve or blk or cyclic" 2>&1 | tail -3
RMHIP_LU_LOOKAHEAD=1 RMHIP_LU_PANEL=columns timeout 100 python scripts/la_diff.py dump 8192 /tmp/ref.npy
RMHIP_LU_LOOKAHEAD=1 timeout 100 python scripts/la_diff.py dump 8192 /tmp/p1.npy
echo "== ref vs p1"; python scripts/la_diff.py cmp /tmp/ref.npy /tmp/p1.npy
echo "== lookahead 16384"; RMHIP_LU_LOOKAHEAD=1 timeout 100 python scripts/la_determinism.py 16384 | tail -3
RMHIP_LU_LOOKAHEAD=1 RMHIP_LU_NB=512 timeout 60 python scripts/lu_time.py 8192 16384 2>&1 | grep "rep=1"
timeout 60 python scripts/lu_time.py 8192 16384 2>&1 | grep "rep=1"
