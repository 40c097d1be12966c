"""Developer tool: interleaved A/B of two librmhip builds (RMHIP_LIBRARY) on the f64 per-op / reduction kernels.
usage: ab_lib.py <old.so> <new.so>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import runmat_amd._lib as L
for k in ("rmhip_set_precision", "rmhip_buffer_bits"):  # absent from builds older than the precision-32 mode
    L.SIGNATURES.pop(k, None)
from runmat_amd import HipProvider
p = HipProvider(0)
N = 8192
a = p.fill_uniform(1, -3.0, 3.0, (N, N)); b = p.fill_uniform(2, -1.0, 1.0, (N, N))
def timed(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    p.timer_begin()
    for _ in range(reps): fn()
    return p.timer_end() / reps
f = lambda h: p.free(h)
out = {"unary_sin": timed(lambda: f(p.unary_sin(a))), "elem_add": timed(lambda: f(p.elem_add(a, b))),
       "dot": timed(lambda: f(p.dot(a, b))), "sum_all": timed(lambda: f(p.reduce_sum(a))),
       "sum_dim0": timed(lambda: f(p.reduce_sum_dim(a, 0))), "scalar_mul": timed(lambda: f(p.scalar_mul(a, 0.5)))}
print(json.dumps(out))
''' % ROOT
res = {"old": [], "new": []}
for rnd in range(4):
    for tag, lib in (("old", sys.argv[1]), ("new", sys.argv[2])):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RMHIP_LIBRARY=os.path.abspath(lib)),
                           capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        res[tag].append(json.loads(r.stdout.strip().splitlines()[-1]))
for k in res["old"][0]:
    med = lambda xs: sorted(xs)[len(xs) // 2]
    print(f"{k:12s} old {med([r[k] for r in res['old']]) * 1e3:8.1f} us   new {med([r[k] for r in res['new']]) * 1e3:8.1f} us")
