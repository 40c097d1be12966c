"""Developer tool: unroll sweep (1, 2, 3, 4, 8) for representative fused bodies at 8192^2 f64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan, FusionGroupPlan, elementwise_math_plan
prov = HipProvider(0)
n = 8192
ha = prov.fill_uniform(1, -np.pi, np.pi, (n, n)); hb = prov.fill_uniform(2, -1, 1, (n, n)); hc = prov.fill_uniform(3, -1, 1, (n, n))
p, o = sin_mul_add_plan(); sh_sin = p.generate_wgsl_for_output(o)
q = FusionGroupPlan(); a, b, c = q.input(), q.input(), q.input(); sh_fma = q.generate_wgsl_for_output(q.primitive("Add", q.primitive("ElemMul", a, b), c))
r = FusionGroupPlan(); a = r.input(); sh_copy = r.generate_wgsl_for_output(r.primitive("UPlus", a))
s = FusionGroupPlan(); a, b = s.input(), s.input(); sh_add = s.generate_wgsl_for_output(s.primitive("Add", a, b))
t = FusionGroupPlan(); a = t.input(); sh_usin = t.generate_wgsl_for_output(t.builtin("sin", a))
u = FusionGroupPlan(); a, b = u.input(), u.input(); sh_div = u.generate_wgsl_for_output(u.primitive("ElemDiv", a, b))
v = FusionGroupPlan(); a, b = v.input(), v.input(); sh_expmul = v.generate_wgsl_for_output(v.primitive("ElemMul", v.builtin("exp", a), b))
def bench(sh, ins, reps=10):
    for _ in range(2): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
        best = min(best, prov.timer_end() / reps)
    return best
kernels = {"sin(A).*B+C": (sh_sin, [ha, hb, hc], 4), "A.*B+C": (sh_fma, [ha, hb, hc], 4), "A+B": (sh_add, [ha, hb], 3), "copy": (sh_copy, [ha], 2),
           "sin(A)": (sh_usin, [ha], 2), "A./B": (sh_div, [ha, hb], 3), "exp(A).*B": (sh_expmul, [hb, hc], 3)}
for name, (sh, ins, ns) in kernels.items():
    res = []
    for unroll in (1, 2, 3, 4, 8):
        os.environ["RMHIP_EW_UNROLL"] = str(unroll)
        res.append("%d:%.0f" % (unroll, ns * 8.0 * n * n / bench(sh, ins) / 1e6))
    print(f"{name:14s} GB/s by unroll  " + "  ".join(res), flush=True)
