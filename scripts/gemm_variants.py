"""Developer tool (GPU box): dgemm variants beside the plain product - matmul_epilogue, syrk (A'*A) and the split-K shapes
(few output tiles, long k) - TFLOP/s.  RMHIP_GEMM_W8=0 shows the four-wave kernel on the same shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider

prov = HipProvider(0)


def rate(tag, f, flops, reps=6):
    for _ in range(2):
        prov.free(f())
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps):
            prov.free(f())
        best = min(best, prov.timer_end() / reps)
    print(f"{tag:46s} {best:8.3f} ms  {flops / best / 1e9:7.2f} TFLOP/s", flush=True)


n = 8192
a = prov.fill_uniform(1, -1, 1, (n, n))
b = prov.fill_uniform(2, -1, 1, (n, n))
rs = prov.fill_uniform(3, 0.5, 1.5, (n, 1))
rate("matmul 8192^3", lambda: prov.matmul(a, b), 2.0 * n ** 3)
rate("matmul_epilogue 8192^3 (alpha, row scale, clamp)", lambda: prov.matmul_epilogue(a, b, alpha=0.5, row_scale=rs, clamp_min=-10.0), 2.0 * n ** 3)
rate("syrk 8192 x 8192", lambda: prov.syrk(a), 2.0 * n ** 3)
prov.free(a); prov.free(b); prov.free(rs)
for (m, k) in ((512, 262144), (1024, 131072), (256, 524288)):
    t = prov.fill_uniform(4, -1, 1, (k, m))
    rate(f"syrk {k} x {m} (split-K)", lambda: prov.syrk(t), 2.0 * m * m * k)
    prov.free(t)
for (m, k) in ((512, 131072), (1024, 65536)):
    x = prov.fill_uniform(5, -1, 1, (m, k))
    y = prov.fill_uniform(6, -1, 1, (k, m))
    rate(f"matmul {m} x {k} x {m} (split-K)", lambda: prov.matmul(x, y), 2.0 * m * m * k)
    prov.free(x); prov.free(y)
a = prov.fill_uniform(1, 0.1, 1, (4096, 4096))
b = prov.fill_uniform(2, 0.1, 1, (4096, 4096))
rate("matmul_epilogue 4096^3 (pow 1.5)", lambda: prov.matmul_epilogue(a, b, pow_exponent=1.5), 2.0 * 4096 ** 3)
