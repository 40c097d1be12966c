"""Developer tool: interleaved A/B timing of fused-kernel tunings (order effects on this hardware are as large as the
differences between tunings, so configurations are alternated and medians compared)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan, FusionGroupPlan
prov = HipProvider(0)
n = 8192
ha = prov.fill_uniform(1, -np.pi, np.pi, (n, n)); hb = prov.fill_uniform(2, -1, 1, (n, n)); hc = prov.fill_uniform(3, -1, 1, (n, n))
p, o = sin_mul_add_plan(); sh_sin = p.generate_wgsl_for_output(o)
q = FusionGroupPlan(); a, b, c = q.input(), q.input(), q.input(); sh_fma = q.generate_wgsl_for_output(q.primitive("Add", q.primitive("ElemMul", a, b), c))
r = FusionGroupPlan(); a = r.input(); sh_copy = r.generate_wgsl_for_output(r.primitive("UPlus", a))
s = FusionGroupPlan(); a, b = s.input(), s.input(); sh_add = s.generate_wgsl_for_output(s.primitive("Add", a, b))
t = FusionGroupPlan(); a = t.input(); sh_usin = t.generate_wgsl_for_output(t.builtin("sin", a))
kernels = {"sin(A).*B+C": (sh_sin, [ha, hb, hc], 4), "A.*B+C": (sh_fma, [ha, hb, hc], 4), "A+B": (sh_add, [ha, hb], 3),
           "copy": (sh_copy, [ha], 2), "sin(A)": (sh_usin, [ha], 2)}
configs = [dict(RMHIP_EW_UNROLL=u, RMHIP_EW_BLOCK=b) for b in ("256", "512", "1024") for u in ("1", "4")]
def run(sh, ins, cfg, reps=15):
    os.environ.update(cfg)
    prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    prov.timer_begin()
    for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    return prov.timer_end() / reps
for name, (sh, ins, ns) in kernels.items():
    res = {i: [] for i in range(len(configs))}
    for rnd in range(8):
        order = list(range(len(configs)))
        if rnd % 2: order.reverse()
        for i in order: res[i].append(run(sh, ins, configs[i]))
    out = []
    for i, cfg in enumerate(configs):
        v = sorted(res[i]); med = v[len(v) // 2]
        out.append("b%s/u%s:%.0f" % (cfg["RMHIP_EW_BLOCK"], cfg["RMHIP_EW_UNROLL"], ns * 8.0 * n * n / med / 1e6))
    print(f"{name:13s} median GB/s  " + "  ".join(out), flush=True)
