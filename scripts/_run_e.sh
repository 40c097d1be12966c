bash scripts/sp_dbg.sh r03e 2>&1 | tail -32
for T in 128 256; do echo "== RB_THREADS=$T"; RMHIP_LU_RB_THREADS=$T python scripts/lu_trace.py 16384 3 2>&1 | tail -2; done
python scripts/solvepath_check.py quick 2>&1 | tail -14
