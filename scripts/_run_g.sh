cd /tmp
for T in 64 256; do echo "== RB_THREADS=$T"; RMHIP_LU_RB_THREADS=$T RMHIP_LU_PANEL_DEBUG=1 python $GRAFT_REPO_ROOT/scripts/lu_trace.py 16384 2 2>&1 | tail -18; done
python $GRAFT_REPO_ROOT/scripts/solvepath_check.py quick 2>&1 | tail -14
python $GRAFT_REPO_ROOT/scripts/lu_trace.py 16384 4 2>&1 | tail -3
python $GRAFT_REPO_ROOT/scripts/lu_trace.py 8192 4 2>&1 | tail -2
