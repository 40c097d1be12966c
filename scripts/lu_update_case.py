"""Developer tool (GPU box): ONE shape of the blocked LU's trailing update (or a deep product on the same workspace) launched a few times, for
counter passes under rocprofv3 (scripts/lu_update_pmc.sh).  Usage: lu_update_case.py <rank512|rank256|rank128|deep> [beta]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
case, beta = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
prov = HipProvider(0)
N, LD = 16384, 16416
w = prov.fill_uniform(7, -1e-3, 1e-3, (LD, N))
if case == "deep":
    a, b, c, dims = (w, 0, 0, 8192, 8192), (w, 0, 8192, 8192, 8192), (w, 8192, 8192, 8192, 8192), (8192, 8192, 8192)
else:
    k = int(case[4:])
    mm = N - k
    a, b, c, dims = (w, k, 0, mm, k), (w, 0, k, k, mm), (w, k, k, mm, mm), (mm, mm, k)
for _ in range(2): prov.blk_gemm(-1.0, a, b, beta, c)
prov.timer_begin()
for _ in range(4): prov.blk_gemm(-1.0, a, b, beta, c)
t = prov.timer_end() / 4
print(f"{case} beta={beta}: {t*1e3:.1f} us {2.0*dims[0]*dims[1]*dims[2]/t/1e9:.1f} TFLOP/s")
