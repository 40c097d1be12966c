"""Developer tool (GPU box): rectangular `A\\b` (least squares through the Gram matrix, rmhip_ops.cpp lstsq_full_rank) on tall and wide
shapes - ms per solve and the residual check against numpy's lstsq on the smaller ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1 << 20, 8, 1), (1 << 18, 32, 1), (100000, 100, 1), (100003, 17, 3), (10000, 1000, 1), (8, 1 << 20, 1), (1000, 10000, 2)]
for (m, n, nrhs) in shapes:
    a = prov.fill_uniform(1, -1.0, 1.0, (m, n))
    b = prov.fill_uniform(2, -1.0, 1.0, (m, nrhs))
    for _ in range(2):
        prov.free(prov.mldivide(a, b))
    prov.synchronize()
    prov.timer_begin()
    for _ in range(5):
        prov.free(prov.mldivide(a, b))
    ms = prov.timer_end() / 5
    import time
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        prov.free(prov.mldivide(a, b))
        prov.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
    line = f"{m:8d} x {n:7d} \\ {nrhs}   {ms*1e3:9.1f} us   wall ms " + " ".join(f"{w:.2f}" for w in walls)
    if m * n <= (1 << 23):
        A, B = prov.download_matrix(a), prov.download_matrix(b)
        x = prov.download_matrix(prov.mldivide(a, b))
        want = np.linalg.lstsq(A, B, rcond=None)[0]
        line += f"   |x - lstsq| / |lstsq| {np.linalg.norm(x - want) / np.linalg.norm(want):.1e}"
    print(line, flush=True)
    prov.free(a); prov.free(b)
