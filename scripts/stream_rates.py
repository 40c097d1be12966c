"""Developer tool: rates of the ahead-of-time streaming kernels at 8192^2 f64 (per-op elementwise, reductions, dot, fill)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
n = 8192
a = prov.fill_uniform(1, -1, 1, (n, n)); b = prov.fill_uniform(2, -1, 1, (n, n))
N = n * n * 8.0
def rate(tag, f, nbytes, reps=10):
    for _ in range(2): prov.free(f())
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(f())
        best = min(best, prov.timer_end() / reps)
    print(f"{tag:28s} {best:.4f} ms  {nbytes/best/1e6:.0f} GB/s", flush=True)
rate("elem_add", lambda: prov.elem_add(a, b), 3 * N)
rate("unary_sin", lambda: prov.unary_sin(a), 2 * N)
rate("scalar_mul", lambda: prov.scalar_mul(a, 2.0), 2 * N)
rate("fill", lambda: prov.fill((n, n), 1.5), N)
rate("reduce_sum all", lambda: prov.reduce_sum(a), N)
rate("reduce_sum dim0", lambda: prov.reduce_sum_dim(a, 0), N)
rate("reduce_sum dim1", lambda: prov.reduce_sum_dim(a, 1), N)
rate("reduce_max all", lambda: prov.reduce_max(a), N)
rate("dot all(dim0)", lambda: prov.dot(a, b, 0), 2 * N)
rate("random_normal", lambda: prov.random_normal((n, n)), N)
rate("random_uniform", lambda: prov.random_uniform((n, n)), N)
for nm in ("abs", "sqrt", "exp", "cos", "tanh", "neg"):
    rate("unary_" + nm, (lambda f: (lambda: f(a)))(getattr(prov, "unary_" + nm)), 2 * N)
from planner_requests import FusionGroupPlan
t = FusionGroupPlan(); x = t.input(); sh = t.generate_wgsl_for_output(t.builtin("sin", x))
rate("fused sin(A)", lambda: prov.fused_elementwise(sh, [a], (n, n), n * n), 2 * N)
