"""Developer tool: x = A\b at n = 16384 (and 8192) under a few knob settings, interleaved (one subprocess per setting)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from runmat_amd import HipProvider
prov = HipProvider(0)
out = {}
for n in [int(x) for x in __import__('os').environ.get('LU_SIZES', '16384,8192').split(',')]:
    a = prov.fill_uniform(31, -1, 1, (n, n)); b = prov.fill_uniform(32, -1, 1, (n, 1))
    ts = []
    for rep in range(4):
        prov.synchronize(); t0 = time.perf_counter()
        x = prov.mldivide(a, b)
        prov.synchronize(); ts.append(time.perf_counter() - t0)
        prov.free(x)
    out[n] = round(min(ts[1:]) * 1e3, 1)
    prov.free(a); prov.free(b)
print(json.dumps(out))
''' % ROOT
if os.environ.get("LU_SWEEP") == "rows":
    configs = [{}, dict(RMHIP_LU_PANEL_ROWS="128"), dict(RMHIP_LU_LOOKAHEAD="1"), dict(RMHIP_LU_LOOKAHEAD="1", RMHIP_LU_NB="64"),
               dict(RMHIP_LU_LOOKAHEAD="1", RMHIP_LU_PANEL_ROWS="128")]
elif os.environ.get("LU_SWEEP") == "verify":
    configs = [{}, dict(RMHIP_LU_NB="128"), dict(RMHIP_LU_NB="256"), dict(RMHIP_LU_NB="192")]
elif os.environ.get("LU_SWEEP") == "small":
    configs = [{}, dict(RMHIP_LU_NB="256"), dict(RMHIP_LU_NB="256", RMHIP_LU_LOOKAHEAD="1"), dict(RMHIP_LU_NB="128", RMHIP_LU_LOOKAHEAD="1"),
               dict(RMHIP_LU_NB="512", RMHIP_LU_LOOKAHEAD="1")]
else:
    configs = [{}] + [dict(RMHIP_LU_NB=str(nb)) for nb in (256, 384, 768, 1024)] + \
              [dict(RMHIP_LU_LA_PAD=str(p)) for p in (0, 4096, 20480)] + [dict(RMHIP_LU_LA_TRSM="1"), dict(RMHIP_LU_LA_TRSM="0")]
res = {i: [] for i in range(len(configs))}
for rnd in range(2):
    for i, cfg in enumerate(configs):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **cfg), capture_output=True, text=True, timeout=120)
        res[i].append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"err": r.stderr[-200:]})
for i, cfg in enumerate(configs):
    print(cfg or "default", res[i], flush=True)
