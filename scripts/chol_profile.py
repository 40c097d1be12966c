"""Developer tool (GPU box): one chol at the given order, twice (for a kernel trace).  Usage: chol_profile.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
hm = prov.fill_uniform(20, -1.0, 1.0, (m, m))
spd = prov.elem_add(prov.syrk(hm), prov.scalar_mul(prov.eye((m, m)), float(m)))
for _ in range(2):
    prov.free(prov.chol(spd).factor)
prov.synchronize()
prov.timer_begin(); prov.free(prov.chol(spd).factor); print(f"chol {m}: {prov.timer_end():.3f} ms")
