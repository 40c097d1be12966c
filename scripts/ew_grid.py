"""Developer tool (GPU box): per-op elementwise kernels over a grid of 2-D shapes with full / row / column / scalar second operands, unary
sin, transpose materialisation - us and GB/s of the streamed bytes, to spot dispatch cliffs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0, precision="F32") if os.environ.get("GRID_F32") == "1" else HipProvider(0)  # GRID_F32=1: precision-32 provider (byte counts printed are still those of f64)
dims = [32, 512, 8192, 524288]
for rows in dims:
    for cols in dims:
        if rows * cols > 2 ** 27 or rows * cols < 2 ** 18:
            continue
        a = prov.fill_uniform(3, -1.0, 1.0, (rows, cols))
        full = prov.fill_uniform(4, -1.0, 1.0, (rows, cols))
        row = prov.fill_uniform(5, -1.0, 1.0, (1, cols))
        col = prov.fill_uniform(6, -1.0, 1.0, (rows, 1))
        N = rows * cols * 8.0
        line = f"{rows:7d} x {cols:7d} "
        for name, f, nb in (("add", lambda: prov.elem_add(a, full), 3 * N), ("+row", lambda: prov.elem_add(a, row), 2 * N),
                            ("+col", lambda: prov.elem_add(a, col), 2 * N), ("*2", lambda: prov.scalar_mul(a, 2.0), 2 * N),
                            ("sin", lambda: prov.unary_sin(a), 2 * N), ("T", lambda: prov.unary_neg(prov.transpose(a)), 4 * N)):
            for _ in range(2):
                prov.free(f())
            prov.timer_begin()
            for _ in range(6):
                prov.free(f())
            ms = prov.timer_end() / 6
            line += f" {name} {ms*1e3:7.1f}us {nb/ms/1e6:5.0f}"
        print(line, flush=True)
        for h in (a, full, row, col):
            prov.free(h)
