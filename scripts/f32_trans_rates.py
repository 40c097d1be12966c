"""Developer tool (GPU box): f32 matmul rates with plain and transposed A (precision-32 provider)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from runmat_amd import HipProvider
p = HipProvider(0, precision="F32")
n = 8192
a = p.fill_uniform(1, -1, 1, (n, n)); b = p.fill_uniform(2, -1, 1, (n, n)); at = p.transpose(a)
for tag, f in (("A*B", lambda: p.matmul(a, b)), ("A'*B", lambda: p.matmul(at, b)), ("syrk", lambda: p.syrk(a))):
    for _ in range(2): p.free(f())
    p.timer_begin()
    for _ in range(5): p.free(f())
    ms = p.timer_end() / 5
    print(f"f32 {tag}: {ms:.3f} ms {2.0*n**3/ms/1e9:.1f} TFLOP/s", flush=True)
