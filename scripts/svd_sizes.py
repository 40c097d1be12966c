"""Developer tool (GPU box): time of the Jacobi-SVD fallback of `A\\b` on rank-deficient square systems of growing order (run with
RMHIP_SVD_MAX_COLS at or above the largest order), residual and minimum-norm check against numpy's lstsq at the smaller ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
rng = np.random.default_rng(5)
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2048]:
    r = n - max(3, n // 50)
    A = rng.standard_normal((n, r)) @ rng.standard_normal((r, n)) / np.sqrt(n)
    b = A @ rng.standard_normal((n, 2))
    ha, hb = prov.upload(A), prov.upload(b)
    prov.synchronize()
    t0 = time.perf_counter()
    hx = prov.mldivide(ha, hb)
    prov.synchronize()
    dt = time.perf_counter() - t0
    x = prov.download(hx).reshape((n, 2), order="F")
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    line = f"n={n:5d} rank={r:5d}  {dt*1e3:9.1f} ms  residual {res:.2e}  svd_solves {prov.lu_stats()['svd_solves']}"
    if n <= 2048:
        want = np.linalg.lstsq(A, b, rcond=None)[0]
        line += f"  |x - lstsq| / |lstsq| {np.linalg.norm(x - want) / np.linalg.norm(want):.2e}"
    print(line, flush=True)
