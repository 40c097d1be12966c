"""Developer tool: rates of the general-broadcast kernels at 8192^2 f64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import FusionGroupPlan
prov = HipProvider(0)
n = 8192
a = prov.fill_uniform(1, -1, 1, (n, n)); row = prov.fill_uniform(2, -1, 1, (1, n)); col = prov.fill_uniform(3, -1, 1, (n, 1))
N = n * n * 8.0
def rate(tag, f, nbytes, reps=10):
    for _ in range(2): prov.free(f())
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(f())
        best = min(best, prov.timer_end() / reps)
    print(f"{tag:34s} {best:.4f} ms  {nbytes/best/1e6:.0f} GB/s", flush=True)
rate("elem_add(A, row)", lambda: prov.elem_add(a, row), 2 * N)
rate("elem_add(A, col)", lambda: prov.elem_add(a, col), 2 * N)
rate("elem_add(col, row) outer", lambda: prov.elem_add(col, row), N)
p = FusionGroupPlan(); x, r, c = p.input(), p.input(), p.input()
sh = p.generate_wgsl_for_output(p.primitive("Add", p.primitive("ElemMul", p.builtin("sin", x), r), c))
rate("fused sin(A).*row + col", lambda: prov.fused_elementwise(sh, [a, row, col], (n, n), n * n), 2 * N)
q = FusionGroupPlan(); x, r = q.input(), q.input()
sh2 = q.generate_wgsl_for_output(q.primitive("Sub", x, r))
rate("fused A - row (centering)", lambda: prov.fused_elementwise(sh2, [a, row], (n, n), n * n), 2 * N)
