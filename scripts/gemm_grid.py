"""Developer tool (GPU box): dgemm over a grid of shapes (m, n, k in {32, 128, 512, 2048, 8192}) - us, TFLOP/s and the effective GB/s
of the operands + result, to spot dispatch cliffs.  GEMM_F32=1: precision-32 provider."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
F32 = os.environ.get("GEMM_F32") == "1"
prov = HipProvider(0, precision="F32") if F32 else HipProvider(0)
dims = [32, 128, 512, 2048, 8192]
bufs = {}
def buf(r, c, seed):
    key = (r, c, seed)
    if key not in bufs:
        bufs[key] = prov.fill_uniform(seed, -1, 1, (r, c))
    return bufs[key]
for m in dims:
    for n in dims:
        for k in dims:
            if m * n * k > 8192 * 8192 * 2048:
                continue
            a, b = buf(m, k, 1), buf(k, n, 2)
            for _ in range(2):
                prov.free(prov.matmul(a, b))
            reps = 20 if m * n * k < 1e9 else 4
            prov.timer_begin()
            for _ in range(reps):
                prov.free(prov.matmul(a, b))
            ms = prov.timer_end() / reps
            es = 4 if F32 else 8
            gb = es * (m * k + k * n + m * n) / ms / 1e6
            print(f"{m:5d} {n:5d} {k:5d}  {ms*1e3:9.1f} us  {2.0*m*n*k/ms/1e9:7.2f} TFLOP/s  {gb:7.0f} GB/s", flush=True)
