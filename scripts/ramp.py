"""Developer tool: per-batch timing of the fused kernel right after start-up (DVFS ramp check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan
prov = HipProvider(0); n = 8192
ha = prov.fill_uniform(1, -np.pi, np.pi, (n, n)); hb = prov.fill_uniform(2, -1, 1, (n, n)); hc = prov.fill_uniform(3, -1, 1, (n, n))
p, o = sin_mul_add_plan(); sh = p.generate_wgsl_for_output(o)
prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n)); prov.synchronize()
res = []
for b in range(40):
    prov.timer_begin()
    for _ in range(10): prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n))
    res.append(round(prov.timer_end()/10, 4))
print("ms per step by batch of 10:", res)
t0 = time.perf_counter()
for _ in range(200): prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n))
prov.synchronize(); print("host wall per step (200):", (time.perf_counter()-t0)/200*1e3, "ms")
# host-side cost of one call (no sync)
t0 = time.perf_counter()
for _ in range(200): prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n))
t1 = time.perf_counter(); prov.synchronize()
print("host enqueue per call:", (t1-t0)/200*1e6, "us")
