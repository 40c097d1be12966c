#!/usr/bin/env python3
"""Times k_rng_normal (eager random_normal of 1e8 f64 samples) and the Monte-Carlo step with a lazy / materialised Z with HIP events on
the library's stream.  python scripts/rng_time.py [reps]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / "tests"))
import numpy as np  # noqa: E402

from runmat_amd import HipProvider  # noqa: E402
from runmat_amd import sharding as sh  # noqa: E402
from planner_requests import monte_carlo_shaders  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
prov = HipProvider(0)
n = 100_000_000
prov.set_lazy_random(False)
for _ in range(3):
    prov.free(prov.random_normal((n, 1)))
prov.synchronize()
prov.timer_begin()
for _ in range(reps):
    prov.free(prov.random_normal((n, 1)))
us = prov.timer_end() / reps * 1e3
print(f"k_rng_normal 1e8 f64: {us:.1f} us  {8e8 / us / 1e3:.0f} GB/s  frac {8e8 / us / 1e3 / 8192:.3f}")
shaders = monte_carlo_shaders(100.0)
g = sh.Group()
for lazy in (False, True):
    prov.set_lazy_random(lazy)
    for _ in range(3):
        sh.monte_carlo_price_fused(prov, g, n, 1, shaders, rng_state=0x9E3779B97F4A7C15)
    prov.synchronize()
    import time

    t0 = time.perf_counter()
    for _ in range(reps):
        price, _ = sh.monte_carlo_price_fused(prov, g, n, 1, shaders, rng_state=0x9E3779B97F4A7C15)
    prov.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"monte-carlo step lazy={lazy}: {ms:.4f} ms  price {price!r}")
prov.close()
