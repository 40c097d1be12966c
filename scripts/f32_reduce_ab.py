"""Developer tool: interleaved A/B of two builds on precision-32 reductions (RMHIP_LIBRARY). usage: f32_reduce_ab.py a.so b.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, %r)
from runmat_amd import HipProvider
p = HipProvider(0, precision="F32")
def timed(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    p.timer_begin()
    for _ in range(reps): fn()
    return round(p.timer_end() / reps * 1e3, 1)
f = lambda h: p.free(h)
out = {}
for shape in ((8192, 8192), (12288, 4096), (4096, 8192)):
    a = p.fill_uniform(1, -3.0, 3.0, shape); b = p.fill_uniform(2, -3.0, 3.0, shape)
    k = "%%dx%%d" %% shape
    out["sum0_" + k] = timed(lambda: f(p.reduce_sum_dim(a, 0)))
    out["dot0_" + k] = timed(lambda: f(p.dot(a, b, 0)))
    out["sum1_" + k] = timed(lambda: f(p.reduce_sum_dim(a, 1)))
    out["sumall_" + k] = timed(lambda: f(p.reduce_sum(a)))
    out["dotall_" + k] = timed(lambda: f(p.dot(p.reshape(a, (shape[0] * shape[1], 1)), p.reshape(b, (shape[0] * shape[1], 1)))))
    p.free(a); p.free(b)
print(json.dumps(out))
''' % ROOT
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RMHIP_LIBRARY=os.path.abspath(l)),
                           capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            sys.exit(r.stderr[-1500:])
        res[l].append(json.loads(r.stdout.strip().splitlines()[-1]))
for k in res[libs[0]][0]:
    print(k, {os.path.basename(l): sorted(x[k] for x in res[l])[1] for l in libs})
