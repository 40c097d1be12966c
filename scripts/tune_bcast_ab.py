"""Developer tool: interleaved A/B of the generated broadcast kernel's block size / elements per thread."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import FusionGroupPlan
PREC = os.environ.get("RMHIP_TUNE_PRECISION", "F64")  # F32: the f32-storage variant (bytes per element halve)
prov = HipProvider(0, precision=PREC)
TY = "f32" if PREC == "F32" else "f64"
EB = 4.0 if PREC == "F32" else 8.0
n = 8192
a = prov.fill_uniform(1, -1, 1, (n, n)); row = prov.fill_uniform(2, -1, 1, (1, n)); col = prov.fill_uniform(3, -1, 1, (n, 1))
p = FusionGroupPlan(); x, r, c = p.input(), p.input(), p.input()
sh_heavy = p.generate_wgsl_for_output(p.primitive("Add", p.primitive("ElemMul", p.builtin("sin", x), r), c), TY)
q = FusionGroupPlan(); x, r = q.input(), q.input()
sh_light = q.generate_wgsl_for_output(q.primitive("Sub", x, r), TY)
cases = {"sin(A).*row+col": (sh_heavy, [a, row, col]), "A-row": (sh_light, [a, row])}
configs = [dict(RMHIP_EW_BCAST_BLOCK=b, RMHIP_EW_BCAST_ELEMS=e) for b in ("256", "512", "1024") for e in ("4", "8")]
def run(sh, ins, cfg, reps=10):
    os.environ.update(cfg)
    prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    prov.timer_begin()
    for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    return prov.timer_end() / reps
for name, (sh, ins) in cases.items():
    res = {i: [] for i in range(len(configs))}
    for rnd in range(6):
        order = list(range(len(configs)))
        if rnd % 2: order.reverse()
        for i in order: res[i].append(run(sh, ins, configs[i]))
    out = []
    for i, cfg in enumerate(configs):
        v = sorted(res[i]); med = v[len(v) // 2]
        out.append("b%s/e%s:%.0f" % (cfg["RMHIP_EW_BCAST_BLOCK"], cfg["RMHIP_EW_BCAST_ELEMS"], 2.0 * EB * n * n / med / 1e6))
    print(f"{name:16s} median GB/s  " + "  ".join(out), flush=True)
