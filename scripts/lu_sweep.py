"""Developer tool: x = A\b timing over sizes, device-generated matrices, back-to-back reps (no host work between)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
for n in [int(a) for a in (sys.argv[1:] or ["2048", "4096", "6144", "8192", "12288", "16384"])]:
    a = prov.fill_uniform(31, -1, 1, (n, n)); b = prov.fill_uniform(32, -1, 1, (n, 1))
    ts = []
    for rep in range(5):
        prov.synchronize(); t0 = time.perf_counter()
        x = prov.mldivide(a, b)
        prov.synchronize(); ts.append(time.perf_counter() - t0)
        prov.free(x)
    best = min(ts[1:]); flops = (2.0 / 3.0) * n ** 3 + 2.0 * n * n
    print(f"n={n}: best {best*1e3:.1f} ms  median {sorted(ts[1:])[2]*1e3:.1f} ms  {flops/best/1e12:.2f} TFLOP/s", flush=True)
    prov.free(a); prov.free(b)
