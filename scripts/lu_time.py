"""Developer tool: time mldivide / lu at a few sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
sizes = [int(s) for s in (sys.argv[1:] or ["2048", "4096", "8192", "16384"])]
for n in sizes:
    hu = prov.fill_uniform(31, -1.0, 1.0, (n, n))
    eye_scaled = prov.upload(np.full((n, 1), float(n)))
    # A = U + n*I : build on host only for small n; for big n add n to the diagonal via a broadcast trick
    d = prov.download(hu).reshape(n, n, order="F") if n <= 8192 else None
    if d is not None:
        d[np.arange(n), np.arange(n)] += n
        ha = prov.upload(d)
        b = d @ np.ones((n, 1))
    else:
        ha = hu  # random matrix (well conditioned enough for timing)
        b = np.ones((n, 1))
    hb = prov.upload(b)
    for rep in range(2):
        prov.synchronize(); t0 = time.perf_counter()
        hx = prov.mldivide(ha, hb)
        prov.synchronize(); dt = time.perf_counter() - t0
        x = prov.download(hx)
        flops = (2.0/3.0)*n**3 + 2.0*n*n
        err = float(np.max(np.abs(x - 1.0))) if d is not None else float('nan')
        print(f"n={n} rep={rep} mldivide {dt*1e3:.1f} ms  {flops/dt/1e12:.2f} TFLOP/s  max|x-1|={err:.2e}", flush=True)
    prov.free(ha); prov.free(hb)
print(prov.telemetry_snapshot())
