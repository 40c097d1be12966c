"""Developer tool (GPU box): accuracy of the device Box-Muller step against an 80-bit reference built on the oracle's (bit-exact)
uniform stream, next to the oracle's own libm path.  Usage: rng_accuracy.py [n]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import oracle  # noqa: E402  (developer tool, not the product)
from runmat_amd import HipProvider  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n -= n % 2
prov = HipProvider(0)
L = np.longdouble
for seed in (oracle.rng_default_seed(), 0x9E3779B97F4A7C15, 12345):
    u, _ = oracle.rng_uniform(seed, n)
    u1 = u[0::2].astype(L)
    u2 = u[1::2].astype(L)
    u1 = np.where(u1 <= 0, L(2.2250738585072014e-308), u1)
    r = np.sqrt(L(-2) * np.log(u1))
    # sin / cos of 2 pi u2 with the reduction done exactly: t = 2 u2, quarter turns out, then the 80-bit functions
    t = 2 * u2
    q = np.rint(2 * t)
    x = (t - q / 2) * L("3.14159265358979323846264338327950288")
    s0, c0 = np.sin(x), np.cos(x)
    k = q.astype(np.int64) % 4
    sn = np.choose(k, [s0, c0, -s0, -c0])
    cs = np.choose(k, [c0, -s0, -c0, s0])
    z_ref = np.empty(n, dtype=L)
    z_ref[0::2] = r * cs
    z_ref[1::2] = r * sn
    prov.set_rng_state(seed)
    z_dev = prov.download(prov.random_normal((n, 1))).ravel()
    z_cpu, _ = oracle.rng_normal(seed, n)
    rr = np.repeat(r, 2)
    for name, z in (("device", z_dev), ("oracle libm", z_cpu)):
        err = np.abs(z.astype(L) - z_ref)
        print(f"seed {seed:#x} {name:12s} max |err| {float(err.max()):.3e}  max |err|/radius {float((err / rr).max()):.3e}  "
              f"rms/radius {float(np.sqrt(np.mean((err / rr) ** 2))):.3e}")
    print(f"   device vs oracle max |diff| {np.abs(z_dev - z_cpu).max():.3e}")
prov.close()
