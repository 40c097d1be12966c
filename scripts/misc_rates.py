"""Developer tool (GPU box): rates of the round-4 construction / linear-algebra / index hooks (misc_ops.hip, index_ops.hip) at 8192^2 f64 -
ms per call and GB/s on the algorithmic bytes (one read of every operand element, one write of every output element).
Usage: misc_rates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)


def timed(label, fn, nbytes, reps=5):
    def free(r):
        for h in (r if isinstance(r, (list, tuple)) else [r]):
            if hasattr(h, "buffer_id"):
                prov.free(h)
    free(fn())
    prov.synchronize()
    prov.timer_begin()
    for _ in range(reps):
        free(fn())
    t = prov.timer_end() / reps  # ms
    print(f"{label:58s} {t:9.3f} ms  {nbytes/(t*1e-3)/1e9:8.1f} GB/s", flush=True)


n = 8192
e = n * n
h = prov.fill_uniform(5, -1.0, 1.0, (n, n))
g = prov.fill_uniform(6, -1.0, 1.0, (n, n))
for dim in (0, 1):
    timed(f"gradient_dim 8192^2 dim {dim}", lambda: prov.gradient_dim(h, dim, 0.5), 16 * e)
    timed(f"trapz_dim 8192^2 dim {dim} (terms + sum)", lambda: prov.trapz_dim(h, dim, 0.5), 8 * e)
    timed(f"cumtrapz_dim 8192^2 dim {dim} (terms + cumsum)", lambda: prov.cumtrapz_dim(h, dim, 0.5), 16 * e)
timed("round_digits 8192^2 (2 decimals)", lambda: prov.round_digits(h, 2), 16 * e)
timed("pow2_scale 8192^2 (fractional exponents)", lambda: prov.pow2_scale(h, g), 24 * e)
timed("unary_angle 8192^2", lambda: prov.unary_angle(h), 16 * e)
timed("norm 8192^2 fro (sweep + fold on the device)", lambda: prov.norm(h, "fro"), 8 * e)
timed("norm 8192^2 one (|a|, column sums, max)", lambda: prov.norm(h, "one"), 8 * e)
timed("norm 8192^2 inf (|a|, row sums, max)", lambda: prov.norm(h, "inf"), 8 * e)
vv = prov.fill_uniform(15, -1.0, 1.0, (10**8, 1))
timed("norm 1e8 vector, two", lambda: prov.norm(vv, "two"), 8 * 10**8)
timed("norm 1e8 vector, p = 3", lambda: prov.norm(vv, "p", 3.0), 8 * 10**8)
prov.free(vv)
timed("issymmetric 8192^2 (not symmetric)", lambda: [prov.issymmetric(h)] and [], 8 * e)
v = prov.fill_uniform(7, -1.0, 1.0, (n, 1))
timed("scatter_column 8192^2", lambda: prov.scatter_column(h, 100, v), 16 * e)
timed("scatter_row 8192^2", lambda: prov.scatter_row(h, 100, prov.reshape(v, (1, n)) if False else v), 16 * e)
timed("diag_from_vector 8192", lambda: prov.diag_from_vector(v, 0), 8 * e)
a, b = prov.fill_uniform(8, -1.0, 1.0, (128, 64)), prov.fill_uniform(9, -1.0, 1.0, (64, 128))
timed("kron (128 x 64) x (64 x 128) -> 8192^2", lambda: prov.kron(a, b), 8 * e)
c3 = prov.fill_uniform(10, -1.0, 1.0, (3, 2 * 10**7))
d3 = prov.fill_uniform(11, -1.0, 1.0, (3, 2 * 10**7))
timed("cross 3 x 2e7", lambda: prov.cross(c3, d3), 3 * 8 * 3 * 2 * 10**7)
ax = [prov.fill_uniform(12, -1.0, 1.0, (n, 1)), prov.fill_uniform(13, -1.0, 1.0, (n, 1))]
timed("ndgrid 8192 x 8192 (two outputs)", lambda: prov.ndgrid(ax, (n, n), 2), 16 * e)
lin = prov.download(prov.fill_uniform(14, 1.0, float(e), (4 * 10**7, 1)))
hl = prov.upload(np.floor(np.asarray(lin)).reshape(-1, 1))
timed("ind2sub 4e7 indices into 8192^2", lambda: prov.ind2sub((n, n), (1, n), hl, e, 4 * 10**7, (4 * 10**7, 1)), 24 * 4 * 10**7)
for m in (2048, 4096, 8192):
    am = prov.upload(np.random.default_rng(m).standard_normal((m, m)) + np.sqrt(m) * np.eye(m))
    t0 = None
    prov.free(prov.inv(am)); prov.synchronize(); prov.timer_begin(); prov.free(prov.inv(am)); t = prov.timer_end()
    print(f"inv {m}^2: {t:9.3f} ms  {(8.0/3.0)*m**3/(t*1e-3)/1e12:6.1f} TFLOP/s on (8/3) n^3", flush=True)
    prov.free(am)
for m in (2048, 4096, 8192, 16384):
    hm = prov.fill_uniform(20, -1.0, 1.0, (m, m))
    spd = prov.elem_add(prov.syrk(hm), prov.scalar_mul(prov.eye((m, m)), float(m)))
    prov.free(hm)
    prov.free(prov.chol(spd).factor); prov.synchronize(); prov.timer_begin(); prov.free(prov.chol(spd).factor); t = prov.timer_end()
    print(f"chol {m}^2: {t:9.3f} ms  {(1.0/3.0)*m**3/(t*1e-3)/1e12:6.1f} TFLOP/s on n^3 / 3", flush=True)
    prov.free(spd)
