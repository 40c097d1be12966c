"""Developer tool: interleaved timing of dgemm build variants (RMHIP_LIBRARY) at 8192^3 and 4096^3, f64."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, %r)
from runmat_amd import HipProvider
p = HipProvider(0)
out = {}
for N in (8192, 4096):
    a = p.fill_uniform(1, -1.0, 1.0, (N, N)); b = p.fill_uniform(2, -1.0, 1.0, (N, N))
    for _ in range(2): p.free(p.matmul(a, b))
    p.timer_begin()
    for _ in range(5): p.free(p.matmul(a, b))
    ms = p.timer_end() / 5
    out[N] = round(2.0 * N ** 3 / ms / 1e9, 2)
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
for l in libs:
    print(os.path.basename(l), res[l])
