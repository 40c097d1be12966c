for d in 264 280 296 328 392 504; do
echo "== unblocked n=16384 dbg=$d (+16 no col, +32 no dgemm, +64 no trsm, +128 no laswp)"
RMHIP_LU_DEBUG=$d python scripts/lu_time.py 16384 2>&1 | grep "rmhip lu" | tail -1
done
for d in 8 40; do
echo "== blocked nb=512 n=16384 dbg=$d"
RMHIP_LU_DEBUG=$d python scripts/lu_time.py 16384 2>&1 | grep "rmhip lu" | tail -1
done
