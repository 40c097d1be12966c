"""Developer tool: a few x = A\\b solves at one size (default 16384) for rocprofv3 --kernel-trace timelines
and RMHIP_LU_PANEL_DEBUG=1 phase ticks.  Usage: lu_trace.py [n] [solves]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = prov.fill_uniform(41, -1, 1, (n, n))
b = prov.fill_uniform(42, -1, 1, (n, 1))
for rep in range(reps):
    prov.synchronize(); t0 = time.perf_counter()
    try:
        x = prov.mldivide(a, b)
        prov.synchronize(); dt = time.perf_counter() - t0
        if rep == reps - 1 and not os.environ.get("RMHIP_LU_SKIP"):  # residual of the last solve: max |A x - b|
            import numpy as np
            r = prov.elem_sub(prov.matmul(a, x), b)
            print(f"max |A x - b| = {np.abs(prov.download(r)).max():.3e}", flush=True)
        prov.free(x)
    except Exception as e:  # RMHIP_LU_SKIP runs produce singular garbage: the time still counts
        prov.synchronize(); dt = time.perf_counter() - t0
    flops = (2.0 / 3.0) * n ** 3 + 2.0 * n * n
    print(f"n={n} rep={rep}: {dt*1e3:.2f} ms  {flops/dt/1e12:.2f} TFLOP/s", flush=True)
