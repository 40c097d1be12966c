"""Developer tool: block-size / blocks-per-CU sweep for the headline fused kernel (complements tune_ew.py)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan
prov = HipProvider(0)
n = 8192
ins = [prov.fill_uniform(1, -np.pi, np.pi, (n, n)), prov.fill_uniform(2, -1, 1, (n, n)), prov.fill_uniform(3, -1, 1, (n, n))]
p, o = sin_mul_add_plan(); sh = p.generate_wgsl_for_output(o)
def bench(reps=10):
    for _ in range(2): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
        best = min(best, prov.timer_end() / reps)
    return best
print("default: %.4f ms  %.0f GB/s" % (bench(), 32.0 * n * n / bench() / 1e6))
rows = []
for block, bpc, unroll in itertools.product((128, 256, 512, 1024), (2, 4, 8, 16, 32, 64), (1, 2)):
    os.environ.update(RMHIP_EW_UNROLL=str(unroll), RMHIP_EW_BLOCK=str(block), RMHIP_EW_BLOCKS_PER_CU=str(bpc))
    ms = bench(8)
    rows.append((32.0 * n * n / ms / 1e6, block, bpc, unroll, ms))
rows.sort(reverse=True)
for r in rows[:10]: print("  %.0f GB/s block=%d bpc=%d unroll=%d %.4f ms" % r)
print("  worst %.0f GB/s block=%d bpc=%d unroll=%d" % rows[-1][:4])
