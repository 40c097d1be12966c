"""Developer tool (GPU box): generated fused reductions (sum(x.^2) along either dimension of 8192^2 and of a ragged shape) beside the
plain sum on the same tensor - us and GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))
from planner_requests import FusionGroupPlan
from runmat_amd import HipProvider, ReductionFlavor

prov = HipProvider(0)


def rate(tag, f, nbytes, reps=12):
    for _ in range(3):
        prov.free(f())
    prov.timer_begin()
    for _ in range(reps):
        prov.free(f())
    ms = prov.timer_end() / reps
    print(f"{tag:44s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:7.0f} GB/s", flush=True)


for shape in ((8192, 8192), (8200, 8192), (8192, 8000)):
    rows, cols = shape
    a = prov.fill_uniform(3, -1.0, 1.0, shape)
    N = rows * cols * 8.0
    for axis in (0, 1):
        p = FusionGroupPlan()
        x = p.input()
        sq = p.primitive("ElemMul", x, x)
        sh = p.generate_reduction_wgsl(sq, "f64", axis=axis)
        red, slices = (rows, cols) if axis == 0 else (cols, rows)
        oshape = (1, cols) if axis == 0 else (rows, 1)
        rate(f"fused sum(x.*x, {axis + 1}) {shape}", lambda: prov.fused_reduction(sh, [a], oshape, red, slices, 256, ReductionFlavor("sum", 1.0)), N)
        rate(f"plain sum(x, {axis + 1}) {shape}", lambda: prov.reduce_sum_dim(a, axis), N)
    prov.free(a)
