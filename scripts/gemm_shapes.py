"""Developer tool: time C -= A*B (the LU update form) for a list of m:n:k shapes inside a 16384+32 ld buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
N = 16384
ld = N + 32
buf = prov.fill_uniform(11, -1, 1, (ld, N))
shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [
    (16320, 64, 64), (8192, 64, 64), (16256, 128, 128), (8192, 128, 128), (128, 8192, 128), (128, 128, 128), (128, 1, 128),
    (256, 8192, 256), (8192, 256, 256), (512, 8192, 512), (8192, 512, 512), (1024, 8192, 1024), (8192, 1024, 1024),
    (2048, 8192, 2048), (8192, 2048, 2048), (12288, 4096, 4096), (8192, 8192, 8192)]
for m, n, k in shapes:
    # A: rows [k, k+m) x cols [0, k); B: rows [0, k) x cols [k, k+n); C: rows [k, k+m) x cols [k, k+n)   (LU layout)
    va, vb, vc = (buf, k, 0, m, k), (buf, 0, k, k, n), (buf, k, k, m, n)
    reps = 50 if 2.0 * m * n * k < 1e10 else 5
    for _ in range(3): prov.blk_gemm(-1e-9, va, vb, 1.0, vc)
    prov.timer_begin()
    for _ in range(reps): prov.blk_gemm(-1e-9, va, vb, 1.0, vc)
    ms = prov.timer_end() / reps
    print(f"m={m} n={n} k={k}: {ms*1e3:.1f} us  {2.0*m*n*k/ms/1e9:.2f} TFLOP/s", flush=True)
