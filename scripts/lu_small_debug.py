"""Developer tool: panel phase ticks (RMHIP_LU_PANEL_DEBUG=1) for small factorizations: lu_small_debug.py n [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = prov.fill_uniform(41, -1, 1, (n, n))
for rep in range(reps):
    prov.synchronize(); t0 = time.perf_counter()
    p, info = prov.blk_lu((a, 0, 0, n, n)) if False else (None, None)
    r = prov.lu(a)
    prov.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} rep={rep}: lu {dt*1e3:.3f} ms = {dt*1e6/n:.2f} us/column", flush=True)
