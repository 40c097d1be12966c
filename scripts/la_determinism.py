"""Developer tool: is the look-ahead LU deterministic?  Factor the same matrix several times, compare bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = prov.fill_uniform(7, -1, 1, (n, n))
b = prov.fill_uniform(8, -1, 1, (n, 1))
ref = None
for rep in range(6):
    x = prov.download(prov.mldivide(a, b))
    if ref is None: ref = x
    d = np.max(np.abs(x - ref)); nbits = int(np.sum(x.view(np.uint64) != ref.view(np.uint64)))
    print(f"rep {rep}: max|x - x0| = {d:.3e}  differing elements {nbits}", flush=True)
