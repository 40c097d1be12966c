"""Developer tool (GPU box): covariance / syrk of tall-skinny matrices (the VALU Gram kernel, special.hip) - us per call; RMHIP_GRAM_BPC
sets its workgroups per CU, RMHIP_NO_GRAM_SKINNY=1 shows the MFMA route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
for (r, c) in [(1 << 20, 8), (1 << 18, 32), (100003, 17), (1 << 22, 3), (50000, 24)]:
    x = prov.fill_uniform(2, -1.0, 1.0, (r, c))
    out = []
    for f in (prov.covariance, prov.syrk):
        for _ in range(3):
            prov.free(f(x))
        prov.timer_begin()
        for _ in range(10):
            prov.free(f(x))
        out.append(prov.timer_end() / 10 * 1e3)
    print(f"{r:8d} x {c:3d}   cov {out[0]:7.1f} us   syrk {out[1]:7.1f} us   ({8.0*r*c/out[1]/1e3:6.0f} GB/s)", flush=True)
    prov.free(x)
