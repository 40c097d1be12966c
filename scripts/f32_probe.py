"""Developer probe: wall time of the basic precision-32 provider paths (finds pathologically slow steps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
from runmat_amd.fusion import sin_mul_add_plan, elementwise_math_plan

def tick(label, t0):
    print(f"{label:40s} {time.perf_counter() - t0:8.3f} s", flush=True)
    return time.perf_counter()

t = time.perf_counter()
p32 = HipProvider(0, precision="F32"); t = tick("init F32", t)
p64 = HipProvider(0); t = tick("init F64", t)
x = np.linspace(-3, 3, 70001).reshape(-1, 1)
h = p32.upload(x); t = tick("upload 70001", t)
d = p32.download(h); t = tick("download", t)
print("roundtrip ok", np.array_equal(d, x.astype(np.float32).astype(np.float64).reshape(-1)))
for name in ("sin", "abs", "single", "tan", "erf"):
    o = getattr(p32, "unary_" + name)(h); p32.synchronize(); t = tick("unary_" + name, t)
o = p32.elem_add(h, h); p32.synchronize(); t = tick("elem_add", t)
o = p32._binary("mod", h, h); p32.synchronize(); t = tick("mod", t)
o = p32.scalar_mul(h, 0.3); p32.synchronize(); t = tick("scalar_mul", t)
o = p32.reduce_sum(h); print(p32.download(o)); t = tick("reduce_sum", t)
plan, out = sin_mul_add_plan()
sh = plan.generate_wgsl_for_output(out, "f32"); t = tick("emit wgsl", t)
for shape in ((1, 1), (7, 1), (512, 512)):
    n = shape[0] * shape[1]
    A = np.random.default_rng(1).uniform(-3, 3, shape)
    hs = [p32.upload(A) for _ in range(3)]; t = tick(f"upload x3 {shape}", t)
    r = p32.fused_elementwise(sh, hs, shape, n); p32.synchronize(); t = tick(f"fused {shape}", t)
    got = p32.download(r); t = tick("download", t)
    A32 = A.astype(np.float32).astype(np.float64)
    want = (np.sin(A32) * A32 + A32).astype(np.float32).astype(np.float64).reshape(-1, order="F")
    print("max rel err", np.max(np.abs(got - want) / np.maximum(1e-30, np.abs(want))))
plan2, out2 = elementwise_math_plan()
sh2 = plan2.generate_wgsl_for_output(out2, "f32")
r = p32.fused_elementwise(sh2, [p32.upload(np.ones((64, 64)))], (64, 64), 4096); p32.synchronize(); t = tick("fused chain (compile)", t)
a = p32.upload(np.random.default_rng(2).standard_normal((300, 70)))
b = p32.upload(np.random.default_rng(3).standard_normal((1, 70)))
o = p32.elem_add(a, b); p32.synchronize(); t = tick("bcast add", t)
print(np.max(np.abs(p32.download_matrix(o) - (p32.download_matrix(a) + p32.download_matrix(b)))))
c = p32.matmul(a, p32.transpose(a)); p32.synchronize(); t = tick("matmul with view", t)
print("tel", p32.telemetry_snapshot()["kernel_launches"])
