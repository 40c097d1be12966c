"""Developer tool: the rank-k update shapes of the blocked LU (m x n x k, k = 256 / 512) through rmhip_matmul under different
kernel / LDS-pad knobs (set in the environment): TFLOP/s from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
tag = " ".join(f"{k[6:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RMHIP_GEMM"))
for (m, n, k) in ((16384, 16384, 512), (12288, 12288, 512), (8192, 8192, 256), (4096, 4096, 256), (8192, 8192, 8192)):
    a = prov.fill_uniform(3, -1, 1, (m, k)); b = prov.fill_uniform(4, -1, 1, (k, n))
    reps = 3 if k > 1024 else 10
    for _ in range(2): prov.free(prov.matmul(a, b))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(prov.matmul(a, b))
        best = min(best, prov.timer_end() / reps)
    print(f"[{tag or 'default'}] {m}x{n}x{k}: {best*1e3:8.1f} us  {2.0*m*n*k/best/1e9:6.1f} TFLOP/s", flush=True)
    prov.free(a); prov.free(b)
