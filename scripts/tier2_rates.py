"""Developer tool: the second-tier kernels of VERDICT round 1 (Monte-Carlo step, image_normalize, sum(x,2)) - ms and GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from planner_requests import monte_carlo_shaders
import numpy as np
from runmat_amd import HipProvider
from runmat_amd import sharding as sh
prov = HipProvider(0)
n = 8192
a = prov.fill_uniform(1, -1, 1, (n, n))
N = n * n * 8.0
def rate(tag, f, nbytes, reps=10):
    for _ in range(2): prov.free(f())
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(f())
        best = min(best, prov.timer_end() / reps)
    print(f"{tag:34s} {best:.4f} ms  {nbytes/best/1e6:.0f} GB/s", flush=True)
rate("reduce_sum dim1 (sum(x,2))", lambda: prov.reduce_sum_dim(a, 1), N)
rate("reduce_mean dim1", lambda: prov.reduce_mean_dim(a, 1), N)
rate("reduce_sum dim0", lambda: prov.reduce_sum_dim(a, 0), N)
rate("random_normal 1e8", lambda: prov.random_normal((100_000_000, 1)), 8e8)
B, H, W = 16, 2160, 3840
img = prov.fill_uniform(5, 0.0, 1.0, (B, H, W))
nb = B * H * W * 8.0
rate("image_normalize (no gamma)", lambda: prov.image_normalize(img, B, H, W, 1e-6, gain=1.0123, bias=-0.02, clamp_zero=True), 3 * nb)
rate("image_normalize (gamma 1.8)", lambda: prov.image_normalize(img, B, H, W, 1e-6, gain=1.0123, bias=-0.02, clamp_zero=True, gamma=1.8), 3 * nb)
prov.free(img); prov.free(a)
g = sh.Group()
for rep in range(4):
    prov.synchronize(); t0 = time.perf_counter()
    price, _ = sh.monte_carlo_price_fused(prov, g, 100_000_000, 1, monte_carlo_shaders(100.0), rng_state=0x9E3779B97F4A7C15)
    dt = time.perf_counter() - t0
    print(f"monte_carlo_price_fused 1e8: {dt*1e3:.3f} ms  {40*1e8/dt/1e9:.0f} GB/s  price {price:.9f}", flush=True)
