"""Developer tool (GPU box): the second-tier entry points over a grid of shapes - image_normalize (batch x height x width), covariance,
dot along either dimension, reduce_moments_nd / reduce_mean_nd over dimension subsets, random_normal / random_uniform sizes, diag_extract -
us and GB/s of the algorithmic bytes, to spot dispatch cliffs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0, precision="F32") if os.environ.get("GRID_F32") == "1" else HipProvider(0)  # GRID_F32=1: precision-32 provider (byte counts printed are still those of f64)


def timed(f, reps=6):
    for _ in range(2):
        r = f()
        for h in (r if isinstance(r, tuple) else (r,)):
            prov.free(h)
    prov.timer_begin()
    for _ in range(reps):
        r = f()
        for h in (r if isinstance(r, tuple) else (r,)):
            prov.free(h)
    return prov.timer_end() / reps


print("image_normalize (B, H, W): plain / gamma   [two reads + one write of 8 B per pixel = 24 B]")
for (B, H, W) in [(16, 2160, 3840), (1, 2160, 3840), (64, 512, 512), (256, 64, 64), (4096, 32, 32), (65536, 16, 16), (3, 1001, 777), (16, 17, 100003), (2, 8192, 8192)]:
    x = prov.fill_uniform(1, 0.0, 1.0, (B, H, W))
    n = B * H * W
    a = timed(lambda: prov.image_normalize(x, B, H, W, 1e-6, gain=1.1, bias=0.1))
    g = timed(lambda: prov.image_normalize(x, B, H, W, 1e-6, gain=1.1, bias=0.1, gamma=1.8))
    print(f"  {B:5d} x {H:5d} x {W:6d}   {a*1e3:8.1f} us {24.0*n/a/1e6:6.0f} GB/s    gamma {g*1e3:8.1f} us {24.0*n/g/1e6:6.0f} GB/s", flush=True)
    prov.free(x)

print("covariance (rows x cols): [read 8 B per element + cols^2 out]")
for (r, c) in [(1 << 20, 8), (1 << 18, 32), (65536, 128), (8192, 1024), (4096, 4096), (100003, 17), (1000, 1000)]:
    x = prov.fill_uniform(2, -1.0, 1.0, (r, c))
    t = timed(lambda: prov.covariance(x))
    print(f"  {r:8d} x {c:5d}   {t*1e3:8.1f} us   {8.0*r*c/t/1e6:6.0f} GB/s  {2.0*r*c*c/t/1e9:6.2f} TFLOP/s", flush=True)
    prov.free(x)

print("dot (rows x cols, dim): [16 B per element]")
for (r, c) in [(1 << 24, 1), (8192, 8192), (32, 1 << 19), (1 << 19, 32), (1001, 4099)]:
    a, b = prov.fill_uniform(3, -1.0, 1.0, (r, c)), prov.fill_uniform(4, -1.0, 1.0, (r, c))
    line = f"  {r:8d} x {c:7d} "
    for dim in (0, 1):
        if (r, c)[dim] == 1:
            continue
        t = timed(lambda: prov.dot(a, b, dim))
        line += f"  dim {dim}: {t*1e3:8.1f} us {16.0*r*c/t/1e6:6.0f} GB/s"
    print(line, flush=True)
    prov.free(a); prov.free(b)

print("reduce_moments_nd / reduce_mean_nd (shape, dims): [8 B per element]")
for shape, dims in [((2160, 3840, 16), (0, 1)), ((16, 2160, 3840), (1, 2)), ((16, 2160, 3840), (0,)), ((64, 64, 64, 64), (0, 2)), ((64, 64, 64, 64), (1, 3)),
                    ((8192, 8192), (0, 1)), ((8192, 8192), (1,))]:
    x = prov.fill_uniform(5, -1.0, 1.0, shape)
    n = int(np.prod(shape))
    t1 = timed(lambda: prov.reduce_moments_nd(x, dims))
    t2 = timed(lambda: prov.reduce_mean_nd(x, dims))
    print(f"  {str(shape):24s} {str(dims):8s}  moments {t1*1e3:8.1f} us {8.0*n/t1/1e6:6.0f} GB/s   mean {t2*1e3:8.1f} us {8.0*n/t2/1e6:6.0f} GB/s", flush=True)
    prov.free(x)

print("random_normal / random_uniform (elements): [8 B per element]")
for n in [1 << 12, 1 << 16, 1 << 20, 1 << 24, 100000000, 1 << 28]:
    tn = timed(lambda: prov.random_normal((n, 1)))
    tu = timed(lambda: prov.random_uniform((n, 1)))
    print(f"  {n:10d}   normal {tn*1e3:8.1f} us {8.0*n/tn/1e6:6.0f} GB/s   uniform {tu*1e3:8.1f} us {8.0*n/tu/1e6:6.0f} GB/s", flush=True)

print("diag_extract (n x n)")
for n in [1024, 8192, 16384]:
    x = prov.fill_uniform(6, -1.0, 1.0, (n, n))
    t = timed(lambda: prov.diag_extract(x))
    print(f"  {n:6d}   {t*1e3:8.1f} us", flush=True)
    prov.free(x)
