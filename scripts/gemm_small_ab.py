"""Developer tool: small products with and without the 64x64-tile kernel (RMHIP_GEMM_SMALL), HIP-event us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
tag = "small=" + os.environ.get("RMHIP_GEMM_SMALL", "1")
for (m, n, k) in ((128, 128, 128), (256, 256, 256), (512, 512, 512), (1024, 1024, 1024), (4096, 128, 128), (8192, 256, 256), (8192, 64, 64),
                  (2048, 128, 64), (16384, 128, 128), (1024, 1024, 64)):
    a = prov.fill_uniform(3, -1, 1, (m, k)); b = prov.fill_uniform(4, -1, 1, (k, n))
    for _ in range(3): prov.free(prov.matmul(a, b))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(20): prov.free(prov.matmul(a, b))
        best = min(best, prov.timer_end() / 20)
    c = prov.download_matrix(prov.matmul(a, b))
    ref = prov.download_matrix(a) @ prov.download_matrix(b)
    print(f"{tag} {m}x{n}x{k}: {best*1e3:7.1f} us  {2.0*m*n*k/best/1e9:8.1f} GFLOP/s  max err {np.max(np.abs(c-ref)):.2e}  sum {float(np.sum(c)):.15e}", flush=True)
    prov.free(a); prov.free(b)
