"""Developer tool: k_dgemm (four waves per block) against k_dgemm_w8 (RMHIP_GEMM_W8=2) over product sizes, TFLOP/s from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
tag = "w8=" + os.environ.get("RMHIP_GEMM_W8", "1")
out = []
for (m, n, k) in ((2048, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 1024), (16384, 16384, 128), (16384, 16384, 256), (4096, 4096, 16384), (12288, 4096, 2048)):
    a = prov.fill_uniform(3, -1, 1, (m, k)); b = prov.fill_uniform(4, -1, 1, (k, n))
    reps = 3 if m * n * k > 3e11 else 10
    for _ in range(2): prov.free(prov.matmul(a, b))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(prov.matmul(a, b))
        best = min(best, prov.timer_end() / reps)
    out.append(f"{m}x{n}x{k}: {2.0*m*n*k/best/1e9:5.1f}")
    prov.free(a); prov.free(b)
print(f"[{tag}] " + " | ".join(out), flush=True)
