"""Developer tool (GPU box): per-step wall time of the Monte-Carlo pricing step (a sync per step) - where do the slow steps sit?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))
from planner_requests import monte_carlo_shaders
from runmat_amd import HipProvider
from runmat_amd import sharding as sh
prov = HipProvider(0)
g = sh.Group()
shaders = monte_carlo_shaders(100.0)
ts = []
t00 = time.perf_counter()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    t0 = time.perf_counter()
    sh.monte_carlo_price_fused(prov, g, 100_000_000, 1, shaders, rng_state=0x9E3779B97F4A7C15)
    ts.append((time.perf_counter() - t0) * 1e3)
import statistics
print("median %.3f ms, mean %.3f, first 5 %s" % (statistics.median(ts[5:]), statistics.mean(ts[5:]), [round(t, 3) for t in ts[:5]]))
acc = 0.0
for i, t in enumerate(ts):
    acc += t
    if i >= 5 and t > 1.5 * statistics.median(ts[5:]):
        print(f"  step {i} at {acc:.0f} ms: {t:.3f} ms")
for lo in range(0, len(ts), 50):
    print(f"  steps {lo}-{lo+49}: mean {statistics.mean(ts[lo:lo+50]):.3f}")
