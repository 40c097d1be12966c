"""Developer tool: dgemm rate vs leading dimension / size (checks power-of-two stride aliasing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
cases = [(8192, 0), (8192, 32), (8192, 16), (8192, 128), (8064, 0), (8320, 0), (4096, 0), (4096, 32), (16384, 0), (16384, 32)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for n, pad in cases:
    ld = n + pad
    a = prov.fill_uniform(11, -1, 1, (ld, n)); b = prov.fill_uniform(12, -1, 1, (ld, n)); c = prov.zeros((ld, n))
    va, vb, vc = (a, 0, 0, n, n), (b, 0, 0, n, n), (c, 0, 0, n, n)
    reps = 10 if n <= 8320 else 3
    for _ in range(2): prov.blk_gemm(1.0, va, vb, 0.0, vc)
    prov.timer_begin()
    for _ in range(reps): prov.blk_gemm(1.0, va, vb, 0.0, vc)
    ms = prov.timer_end() / reps
    print(f"n={n} ld={ld}: {ms:.3f} ms  {2.0*n**3/ms/1e9:.1f} TFLOP/s", flush=True)
    for h in (a, b, c): prov.free(h)
