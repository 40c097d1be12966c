"""Developer tool (GPU box): `A\\b`, `lu` and `matmul` at small orders - us per call (wall clock with a synchronize per call, and
back-to-back through the timer): the latency floor of the solve path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
for n in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 100, 128, 200, 256, 384, 512, 768, 1024, 1536, 2048]:
    a = prov.fill_uniform(1, -1.0, 1.0, (n, n))
    b = prov.fill_uniform(2, -1.0, 1.0, (n, 1))
    out = []
    for f in (lambda: prov.mldivide(a, b), lambda: prov.lu(a), lambda: prov.matmul(a, a)):
        try:
            for _ in range(2):
                r = f()
            prov.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                r = f()
            prov.synchronize()
            out.append((time.perf_counter() - t0) / 10 * 1e6)
        except Exception as e:
            out.append(float("nan"))
    print(f"n={n:5d}   A\\b {out[0]:9.1f} us   lu {out[1]:9.1f} us   matmul {out[2]:8.1f} us", flush=True)
