"""Developer tool (GPU box): dgemm on shapes that are not whole 128 x 128 x 16 tiles (the guarded kernel) beside the nearest whole-tile
shapes - TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider

F32 = os.environ.get("GEMM_F32") == "1"  # GEMM_F32=1: precision-32 provider (f32 storage, f32 matrix cores)
prov = HipProvider(0, precision="F32") if F32 else HipProvider(0)


def rate(m, n, k, reps=4):
    a = prov.fill_uniform(1, -1, 1, (m, k))
    b = prov.fill_uniform(2, -1, 1, (k, n))
    for _ in range(2):
        prov.free(prov.matmul(a, b))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps):
            prov.free(prov.matmul(a, b))
        best = min(best, prov.timer_end() / reps)
    prov.free(a); prov.free(b)
    print(f"{m:6d} x {k:6d} x {n:6d}   {best:8.3f} ms  {2.0 * m * n * k / best / 1e9:7.2f} TFLOP/s", flush=True)


shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]] or [
    (8192, 8192, 8192), (8200, 8200, 8200), (8191, 8191, 8191), (10000, 10000, 10000), (8192, 8192, 8200), (8200, 8192, 8192), (8192, 8200, 8192),
    (4096, 4096, 4096), (4100, 4100, 4100), (5000, 5000, 5000), (2048, 2048, 2048), (2000, 2000, 2000), (1000, 1000, 1000), (96, 96, 96), (200, 200, 200)]
for (m, n, k) in shapes:
    rate(m, n, k)
