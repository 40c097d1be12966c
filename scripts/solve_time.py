"""Developer tool: cost of the substitution phase -- x = A\B for several right-hand-side counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
a = prov.fill_uniform(31, -1, 1, (n, n))
for nrhs in (1, 2, 128, 1):
    b = prov.fill_uniform(32, -1, 1, (n, nrhs))
    for rep in range(2):
        prov.synchronize(); t0 = time.perf_counter()
        x = prov.mldivide(a, b)
        prov.synchronize(); dt = time.perf_counter() - t0
        prov.free(x)
    print(f"n={n} nrhs={nrhs}: {dt*1e3:.1f} ms", flush=True)
    prov.free(b)
w = prov.fill_uniform(31, -1, 1, (n, n))
for rep in range(2):
    prov.synchronize(); t0 = time.perf_counter()
    perm, info = prov.blk_lu((w, 0, 0, n, n))
    prov.synchronize(); dt = time.perf_counter() - t0
print(f"n={n} factor only (blk_lu in place, unpadded ld): {dt*1e3:.1f} ms", flush=True)
