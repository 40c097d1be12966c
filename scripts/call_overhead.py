"""Developer tool: host-side cost of one provider call (tiny operands, so the kernel itself is negligible)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan, elementwise_math_plan
prov = HipProvider(0)
def rate(tag, f, n=3000):
    for _ in range(50): prov.free(f())
    prov.synchronize(); t0 = time.perf_counter()
    for _ in range(n): prov.free(f())
    t1 = time.perf_counter(); prov.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: enqueue {1e6*(t1-t0)/n:.2f} us/call, drained {1e6*(t2-t0)/n:.2f} us/call", flush=True)
a = prov.upload(np.ones((8, 8))); b = prov.upload(np.ones((8, 8))); c = prov.upload(np.ones((8, 8)))
plan, out = sin_mul_add_plan(); sh = plan.generate_wgsl_for_output(out, "f64")
plan2, out2 = elementwise_math_plan(); sh2 = plan2.generate_wgsl_for_output(out2, "f64")
rate("unary_sin 8x8", lambda: prov.unary_sin(a))
rate("elem_add 8x8", lambda: prov.elem_add(a, b))
rate(f"fused sin_mul_add 8x8 (shader {len(sh)} chars)", lambda: prov.fused_elementwise(sh, [a, b, c], (8, 8), 64))
x = prov.upload(np.ones((8, 8)))
ins = [x] + [prov.upload(np.array([[v]])) for v in (10.0, 4.0, 0.25, 2.0, 0.1)][: max(0, len(plan2.inputs) - 1)] if hasattr(plan2, "inputs") else [x]
try:
    rate(f"fused chain 8x8 (shader {len(sh2)} chars)", lambda: prov.fused_elementwise(sh2, ins, (8, 8), 64))
except Exception as e:
    print("chain:", str(e)[:100])
rate("matmul 8x8", lambda: prov.matmul(a, b))
rate("reduce_sum 8x8", lambda: prov.reduce_sum(a))
