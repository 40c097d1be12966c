"""Developer tool (GPU box): the solve path (panel-restricted pivoting + multiplier check) against the grid-wide rule:
residuals, largest multiplier, fallbacks, wall-clock.  Usage: solvepath_check.py [quick|time|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
prov = HipProvider(0)


def make(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "u":
        return rng.uniform(-1, 1, (n, n))
    if kind == "dd":
        return rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    if kind == "permdd":
        A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
        return A[rng.permutation(n)]
    if kind == "graded":
        q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
        q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (q1 * np.logspace(0, -10, n)) @ q2
    if kind == "wilk":
        A = np.eye(n) - np.tril(np.ones((n, n)), -1)
        A[:, -1] = 1
        return A
    raise ValueError(kind)


def solve(ha, hb, fast):
    os.environ["RMHIP_LU_FAST"] = "1" if fast else "0"
    prov.synchronize()
    t0 = time.perf_counter()
    hx = prov.mldivide(ha, hb)
    prov.synchronize()
    return hx, (time.perf_counter() - t0) * 1e3


def residual(ha, hx, hb):
    r = prov.elem_sub(prov.matmul(ha, hx), hb)
    v = float(np.abs(prov.download(r)).max())
    prov.free(r)
    return v


if mode in ("quick", "all"):
    for kind, n in (("u", 257), ("u", 1000), ("dd", 1000), ("u", 2048), ("u", 4096), ("dd", 4096), ("graded", 2048), ("wilk", 40),
                    ("permdd", 2048), ("u", 5250), ("u", 8192)):
        A = make(kind, n)
        b = A @ np.ones((n, 1))
        ha, hb = prov.upload(A), prov.upload(b)
        st0 = prov.lu_stats()
        hx, ms1 = solve(ha, hb, True)
        st1 = prov.lu_stats()
        hy, ms0 = solve(ha, hb, False)
        x, y = prov.download(hx), prov.download(hy)
        nrm = np.abs(A).max() * n
        print(f"{kind:7s} n={n:5d} fast {ms1:8.2f} ms res {residual(ha, hx, hb)/nrm:.2e} fwd {np.abs(x-1).max():.2e} | gepp {ms0:8.2f} ms res "
              f"{residual(ha, hy, hb)/nrm:.2e} fwd {np.abs(y-1).max():.2e} | max|l| {st1['last_max_multiplier']:.3g} accepted "
              f"{st1['solve_path_factorizations']-st0['solve_path_factorizations']} fallbacks {st1['pivot_growth_fallbacks']-st0['pivot_growth_fallbacks']}",
              flush=True)
        for h in (ha, hb, hx, hy):
            prov.free(h)

if mode in ("time", "all"):
    for n in (4096, 8192, 12288, 16384):
        ha = prov.fill_uniform(31, -1.0, 1.0, (n, n))
        ones = prov.ones((n, 1))
        hb = prov.matmul(ha, ones)
        flops = (2.0 / 3.0) * n ** 3 + 2.0 * n * n
        for rep in range(3):
            for fast in (True, False):
                hx, ms = solve(ha, hb, fast)
                err = float(np.abs(prov.download(hx) - 1).max()) if rep == 0 else float("nan")
                print(f"n={n} rep={rep} {'fast' if fast else 'gepp'} {ms:8.2f} ms {flops/ms/1e9:6.2f} TF/s  max|x-1| {err:.2e}  max|l| "
                      f"{prov.lu_stats()['last_max_multiplier']:.3g}", flush=True)
                prov.free(hx)
        for h in (ha, ones, hb):
            prov.free(h)
print(prov.lu_stats())
print(prov.telemetry_snapshot()["solve_fallbacks"])
