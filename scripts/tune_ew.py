"""Developer tool: sweep the fused-elementwise code generator's tunables on the GPU."""
import itertools, os, sys
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

def bench(sh, ins, bytes_, reps=10):
    for _ in range(2): prov.free(prov.fused_elementwise(sh, ins, (n, n), n*n))
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n*n))
        best = min(best, prov.timer_end()/reps)
    return best, bytes_/best/1e6

kernels = {"sinfma3": (sh_sin, [ha, hb, hc], 4), "fma3": (sh_fma, [ha, hb, hc], 4), "add2": (sh_add, [ha, hb], 3),
           "copy1": (sh_copy, [ha], 2), "sin1": (sh_usin, [ha], 2)}
for name, (sh, ins, nstreams) in kernels.items():
    rows = []
    for unroll, bpc, ntl, nts, chunked in itertools.product((1, 2, 4), (4, 8, 16, 32), (0, 1), (0, 1), (0, 1)):
        os.environ.update(RMHIP_EW_UNROLL=str(unroll), RMHIP_EW_BLOCK="256", RMHIP_EW_BLOCKS_PER_CU=str(bpc),
                          RMHIP_EW_NT_LOAD=str(ntl), RMHIP_EW_NT_STORE=str(nts), RMHIP_EW_CHUNKED=str(chunked))
        ms, gbs = bench(sh, ins, nstreams*8*n*n, reps=8)
        rows.append((gbs, unroll, bpc, ntl, nts, chunked, ms))
    rows.sort(reverse=True)
    print(f"== {name}: top (GB/s, unroll, blocks/CU, nt_load, nt_store, chunked, ms)")
    for r_ in rows[:8]: print("  %.0f u=%d bpc=%d ntl=%d nts=%d ch=%d %.4f ms" % r_)
    print("   worst: %.0f u=%d bpc=%d ntl=%d nts=%d ch=%d" % rows[-1][:6])
    # marginals
    for idx, label in ((1, "unroll"), (2, "bpc"), (3, "ntl"), (4, "nts"), (5, "chunked")):
        vals = sorted(set(r_[idx] for r_ in rows))
        print("   mean by %s: " % label + ", ".join("%s=%.0f" % (v, np.mean([r_[0] for r_ in rows if r_[idx] == v])) for v in vals))
prov.close()
