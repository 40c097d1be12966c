"""GPU bring-up probe (developer tool): exercises every C-ABI entry point once against the oracle
and prints errors/timings. Not part of the test-suite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import oracle
from runmat_amd import HipProvider, ReductionFlavor, ProviderError
from planner_requests import sin_mul_add_plan, elementwise_math_plan, FusionGroupPlan

def t(name, f):
    try:
        t0 = time.time(); r = f(); print(f"[ok] {name}: {r} ({time.time()-t0:.2f}s)", flush=True)
    except Exception as e:
        print(f"[FAIL] {name}: {type(e).__name__}: {e}", flush=True)

prov = HipProvider(0)
print(prov.device_info_struct())
rng = np.random.default_rng(1)

def ew():
    m, n = 300, 200
    A = rng.uniform(-3, 3, (m, n)); B = rng.uniform(-1, 1, (m, n)); C = rng.uniform(-1, 1, (m, n))
    p, o = sin_mul_add_plan()
    hd = prov.fused_elementwise(p.generate_wgsl_for_output(o), [prov.upload(A), prov.upload(B), prov.upload(C)], (m, n), m*n)
    return np.max(np.abs(prov.download_matrix(hd) - oracle.sin_mul_add(A, B, C)))
t("fused sin.*B+C", ew)

def ew_bcast():
    A = rng.uniform(-3, 3, (300, 1)); B = rng.uniform(-1, 1, (1, 200)); C = np.array([[0.5]])
    p, o = sin_mul_add_plan()
    hd = prov.fused_elementwise(p.generate_wgsl_for_output(o), [prov.upload(A), prov.upload(B), prov.upload(C)], (300, 200), 60000)
    ref = oracle.binary("add", oracle.binary("mul", oracle.unary("sin", A), B), C)
    return np.max(np.abs(prov.download_matrix(hd) - ref))
t("fused broadcast", ew_bcast)

def chain():
    x = np.linspace(0, 4*np.pi, 1024*64).reshape(1024, 64, order="F")
    p, o = elementwise_math_plan()
    ins = [prov.upload(x)] + [prov.upload(np.array([[v]])) for v in (10.0, 4.0, 0.25, 2.0, 0.1)]
    hd = prov.fused_elementwise(p.generate_wgsl_for_output(o), ins, x.shape, x.size)
    return np.max(np.abs(prov.download_matrix(hd) - oracle.elementwise_math_chain(x)))
t("elementwise-math chain", chain)

for rowmap in ("0", "1"):
    def mm():
        os.environ["RMHIP_MFMA_ROWMAP"] = rowmap
        X = rng.uniform(-1, 1, (256, 64)); Y = rng.uniform(-1, 1, (64, 128))
        Z = prov.download_matrix(prov.matmul(prov.upload(X), prov.upload(Y)))
        return np.max(np.abs(Z - oracle.matmul(X, Y)))
    # rowmap is latched on first use inside the library; only the first iteration is meaningful per process
    t(f"matmul 256x64x128 rowmap(env)={rowmap}", mm)

def mm_edge():
    X = rng.uniform(-1, 1, (97, 45)); Y = rng.uniform(-1, 1, (45, 33))
    Z = prov.download_matrix(prov.matmul(prov.upload(X), prov.upload(Y)))
    return np.max(np.abs(Z - oracle.matmul(X, Y)))
t("matmul edge 97x45x33", mm_edge)

def red(axis):
    X = rng.uniform(-1, 1, (512, 384)); W = rng.uniform(-1, 1, (512, 384))
    p = FusionGroupPlan(); a, b = p.input(), p.input(); m = p.primitive("ElemMul", a, b)
    sh = p.generate_reduction_wgsl(m, "f64", axis=axis)
    if axis == 0:
        h = prov.fused_reduction(sh, [prov.upload(X), prov.upload(W)], (384,), 512, 384, 256, ReductionFlavor.Sum())
        ref = oracle.reduce_sum(X*W, [0]).reshape(-1)
    else:
        h = prov.fused_reduction(sh, [prov.upload(X), prov.upload(W)], (512,), 384, 512, 256, ReductionFlavor.Sum())
        ref = oracle.reduce_sum(X*W, [1]).reshape(-1)
    return np.max(np.abs(prov.download(h) - ref))
t("fused reduction axis0", lambda: red(0))
t("fused reduction axis1", lambda: red(1))

def redall():
    X = rng.uniform(-1, 1, (1000, 1000))
    return [abs(prov.download(prov.reduce_sum(prov.upload(X)))[0] - oracle.reduce_sum(X, "all").reshape(-1)[0]),
            np.max(np.abs(prov.download(prov.reduce_sum_dim(prov.upload(X), 0)) - oracle.reduce_sum(X, [0]).reshape(-1))),
            np.max(np.abs(prov.download(prov.reduce_sum_dim(prov.upload(X), 1)) - oracle.reduce_sum(X, [1]).reshape(-1))),
            abs(prov.download(prov.reduce_mean(prov.upload(X)))[0] - oracle.reduce_sum(X, "all", mean=True).reshape(-1)[0])]
t("reduce_sum all/dim0/dim1/mean", redall)

def unary_binary():
    X = rng.uniform(0.1, 3, (333, 77)); Y = rng.uniform(0.1, 3, (333, 77))
    errs = {}
    hx, hy = prov.upload(X), prov.upload(Y)
    for op in ("sin", "cos", "exp", "log", "sqrt", "abs", "tanh", "floor", "sign", "log1p", "expm1"):
        errs[op] = float(np.max(np.abs(prov.download_matrix(getattr(prov, "unary_"+op)(hx)) - oracle.unary(op, X)) / np.maximum(1, np.abs(oracle.unary(op, X)))))
    for op in ("add", "sub", "mul", "div", "pow", "max", "min", "hypot", "atan2"):
        errs[op] = float(np.max(np.abs(prov.download_matrix(getattr(prov, "elem_"+op)(hx, hy)) - oracle.binary(op, X, Y)) / np.maximum(1, np.abs(oracle.binary(op, X, Y)))))
    return errs
t("unary/binary", unary_binary)

def rngt():
    prov.set_rng_state(oracle.rng_default_seed())
    u = prov.download(prov.random_uniform((1001, 1)))
    ru, s = oracle.rng_uniform(oracle.rng_default_seed(), 1001)
    z = prov.download(prov.random_normal((777, 1)))
    rz, s2 = oracle.rng_normal(s, 777)
    return [float(np.max(np.abs(u - ru))), float(np.max(np.abs(z - rz))), prov.get_rng_state() == s2]
t("rng uniform(bit-exact expect 0)/normal/state", rngt)

def lut():
    A = rng.uniform(-1, 1, (200, 200))
    r = prov.lu(prov.upload(A))
    comb, L, U, P, pv = oracle.lu(A)
    return [float(np.max(np.abs(prov.download_matrix(r.combined) - comb))), bool(np.array_equal(prov.download(r.perm_vector), pv.reshape(-1))),
            float(np.max(np.abs(prov.download_matrix(r.lower) - L))), float(np.max(np.abs(prov.download_matrix(r.upper) - U))),
            bool(np.array_equal(prov.download_matrix(r.perm_matrix), P))]
t("lu 200", lut)

def solve(n):
    A = rng.uniform(-1, 1, (n, n)) + n*np.eye(n); x = np.ones((n, 1)); b = A @ x
    t0 = time.time(); h = prov.mldivide(prov.upload(A), prov.upload(b)); xs = prov.download(h); dt = time.time()-t0
    return [float(np.max(np.abs(xs - 1))), dt]
t("mldivide 300", lambda: solve(300))
t("mldivide 2048", lambda: solve(2048))

def perf():
    n = 8192
    out = {}
    ha = prov.fill_uniform(1, -np.pi, np.pi, (n, n)); hb = prov.fill_uniform(2, -1, 1, (n, n)); hc = prov.fill_uniform(3, -1, 1, (n, n))
    p, o = sin_mul_add_plan(); sh = p.generate_wgsl_for_output(o)
    for _ in range(3): prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n))
    prov.timer_begin()
    for _ in range(10): prov.free(prov.fused_elementwise(sh, [ha, hb, hc], (n, n), n*n))
    ms = prov.timer_end()/10
    out["fused_ew_ms"] = ms; out["fused_ew_GBs"] = 4*8*n*n/ms/1e6
    for _ in range(2): prov.free(prov.matmul(hb, hc))
    prov.timer_begin()
    for _ in range(5): prov.free(prov.matmul(hb, hc))
    ms = prov.timer_end()/5
    out["dgemm_ms"] = ms; out["dgemm_TF"] = 2*n**3/ms/1e9
    prov.timer_begin()
    for _ in range(5): prov.free(prov.reduce_sum(ha))
    out["sum_all_ms"] = prov.timer_end()/5
    prov.timer_begin()
    for _ in range(5): prov.free(prov.reduce_sum_dim(ha, 1))
    out["sum_dim1_ms"] = prov.timer_end()/5
    prov.timer_begin()
    for _ in range(5): prov.free(prov.reduce_sum_dim(ha, 0))
    out["sum_dim0_ms"] = prov.timer_end()/5
    prov.timer_begin()
    for _ in range(3): prov.free(prov.random_normal((100_000_000, 1)))
    out["randn_1e8_ms"] = prov.timer_end()/3
    return out
t("perf", perf)
print(prov.telemetry_snapshot())
prov.close()
