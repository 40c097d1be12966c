"""Developer tool (GPU box): reductions over a grid of 2-D shapes (rows, cols in {32, 512, 2048, 8192, 65536, 524288}, at most 2^27 elements):
sum / min-with-indices / std / cumsum along both dimensions - us and GB/s, to spot dispatch cliffs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0, precision="F32") if os.environ.get("GRID_F32") == "1" else HipProvider(0)  # GRID_F32=1: precision-32 provider (byte counts printed are still those of f64)
dims = [32, 512, 2048, 8192, 65536, 524288]
def free(r):
    for h in ((r.values, r.indices) if hasattr(r, "values") else (r,)):
        prov.free(h)
for rows in dims:
    for cols in dims:
        if rows * cols > 2 ** 27 or rows * cols < 2 ** 18:
            continue
        a = prov.fill_uniform(3, -1.0, 1.0, (rows, cols))
        N = rows * cols * 8.0
        line = f"{rows:7d} x {cols:7d} "
        for name, f, nb in (("sum0", lambda: prov.reduce_sum_dim(a, 0), N), ("sum1", lambda: prov.reduce_sum_dim(a, 1), N),
                            ("min0", lambda: prov.reduce_min_dim(a, 0), N), ("min1", lambda: prov.reduce_min_dim(a, 1), N),
                            ("std0", lambda: prov.reduce_std_dim(a, 0), N), ("std1", lambda: prov.reduce_std_dim(a, 1), N),
                            ("cum0", lambda: prov.cumsum_scan(a, 0), 2 * N), ("cum1", lambda: prov.cumsum_scan(a, 1), 2 * N)):
            for _ in range(2):
                free(f())
            prov.timer_begin()
            for _ in range(6):
                free(f())
            ms = prov.timer_end() / 6
            line += f" {name} {ms*1e3:7.1f}us {nb/ms/1e6:5.0f}"
        print(line, flush=True)
        prov.free(a)
