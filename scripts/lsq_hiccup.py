"""Developer tool (GPU box): repeated rectangular solves of one shape, wall time per call and the LU counters - to catch sporadic slow calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
m, n, nrhs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000x10000x2").split("x")]
a = prov.fill_uniform(1, -1.0, 1.0, (m, n))
b = prov.fill_uniform(2, -1.0, 1.0, (m, nrhs))
prev = prov.lu_stats()
for i in range(40):
    t0 = time.perf_counter()
    prov.free(prov.mldivide(a, b))
    prov.synchronize()
    w = (time.perf_counter() - t0) * 1e3
    st = prov.lu_stats()
    diff = {k: st[k] - prev[k] for k in st if isinstance(st[k], (int, float)) and st[k] != prev[k]}
    prev = st
    print(f"{i:3d} {w:8.2f} ms  {diff}", flush=True)
