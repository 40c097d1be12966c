"""Developer tool (GPU box): repeat x = A\\b at one size and report every solve whose residual or solution differs from the first.
Usage: lu_stress.py [n] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a = prov.fill_uniform(41, -1, 1, (n, n))
b = prov.fill_uniform(42, -1, 1, (n, 1))
x0 = None
bad = 0
for rep in range(reps):
    x = prov.mldivide(a, b)
    r = prov.elem_sub(prov.matmul(a, x), b)
    res = float(np.abs(prov.download(r)).max())
    xh = prov.download(x).ravel()
    if x0 is None: x0 = xh
    same = bool(np.array_equal(xh, x0))
    if res > 1e-6 or not same:
        bad += 1
        d = np.nonzero(xh != x0)[0]
        print(f"rep {rep}: residual {res:.3e} identical={same} first differing row {d[0] if d.size else -1} of {d.size}", flush=True)
    prov.free(x); prov.free(r)
print(f"n={n} reps={reps} bad={bad}", flush=True)
