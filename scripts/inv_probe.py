import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from runmat_amd import HipProvider
prov = HipProvider(0)
for n in (64, 1000, 2048, 4096, 8192):
    a = prov.fill_uniform(3, -1.0, 1.0, (n, n))
    A = prov.download_matrix(a) + n * np.eye(n) if n <= 4096 else None
    if A is not None:
        prov.free(a); a = prov.upload(A)
    e = prov.eye((n, n))
    x = prov.mldivide(a, e); prov.free(x)
    prov.synchronize(); prov.timer_begin()
    x = prov.mldivide(a, e)
    ms = prov.timer_end()
    if A is not None:
        X = prov.download_matrix(x)
        err = np.max(np.abs(A @ X - np.eye(n)))
    else:
        err = float('nan')
    print(f"n={n}: A\\I {ms:.2f} ms  ({(2/3+2)*n**3/ms/1e9:.1f} TFLOP/s)  max|A X - I| = {err:.2e}", flush=True)
