"""GPU parity tests (-m gpu): every hot-path entry point of the C ABI against the CPU oracle on
the same seeded inputs.  Bar: bit-exact for integer / index work and for IEEE-exact arithmetic
(+,-,*,/,sqrt,abs,floor,...,max,min with the CPU NaN/-0 policy, broadcast indexing, the LCG
stream, LU pivot vectors); stated ulp tolerances for libm functions (the reference's CPU path
calls the platform libm through Rust std; ROCm's ocml rounds differently in the last place);
k*eps*sum|a||b| for matmul (MFMA fma chain vs the CPU's separately rounded sum += a*b);
residual / forward-error bounds for A\\b exactly as the reference's own tests state them."""
import math

import os

import numpy as np
import pytest

from workloads import OracleOps, ProviderOps, golden_monte_carlo_cases, lcg_monte_carlo_price

pytestmark = pytest.mark.gpu

EPS = 2.220446049250313e-16


def ulp_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    same_inf = np.isinf(want) & (got == want)
    sp = np.spacing(np.maximum(np.abs(want), 2.2250738585072014e-308))
    with np.errstate(invalid="ignore"):
        e = np.abs(got - want) / sp
    e = np.where(both_nan | same_inf, 0.0, e)
    return float(np.nanmax(np.where(np.isnan(e), np.inf, e))) if e.size else 0.0


def bits_equal(got, want):
    got, want = np.ascontiguousarray(got, dtype=np.float64), np.ascontiguousarray(want, dtype=np.float64)
    a, b = got.view(np.uint64), want.view(np.uint64)
    nan_ok = np.isnan(got) & np.isnan(want)
    return bool(np.all((a == b) | nan_ok))


# ---- memory / handles --------------------------------------------------------------------------
def test_upload_download_roundtrip_and_errors(prov):
    from runmat_amd import GpuTensorHandle, ProviderError

    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    h = prov.upload(a)
    assert h.shape == (3, 4) and h.device_id == prov.device_id()
    assert np.array_equal(prov.download(h), a.reshape(-1, order="F"))  # column-major like HostTensorOwned
    before = prov.telemetry_snapshot()
    r = prov.reshape(h, (4, 3))
    # same buffer, new shape: the trait default (lib.rs:2676-2684) and the wgpu provider return the SAME buffer_id, and
    # callers consume the source handle without freeing it, so exactly one free must release the storage
    assert r.buffer_id == h.buffer_id and r.shape == (4, 3) and prov._handle(r.buffer_id).shape == (4, 3)
    assert np.array_equal(prov.download(r), a.reshape(-1, order="F"))
    prov.free(r)
    after = prov.telemetry_snapshot()
    assert after["bytes_allocated"] == before["bytes_allocated"] and after["bytes_pooled"] >= before["bytes_pooled"] + 96
    with pytest.raises(ProviderError) as e:
        prov.download(h)
    assert e.value.code == 5 and "buffer not found" in str(e.value)
    with pytest.raises(ProviderError):
        prov.free(h)
    with pytest.raises(ProviderError):  # foreign device id (io.rs:269-275)
        prov.free(GpuTensorHandle((1, 1), prov.device_id() + 77, 1))
    z = prov.zeros((5, 2))
    o = prov.ones((5, 2))
    assert np.array_equal(prov.download(z), np.zeros(10)) and np.array_equal(prov.download(o), np.ones(10))
    e0 = prov.upload(np.zeros((0, 3)))
    assert prov.download(e0).size == 0
    for x in (z, o, e0):
        prov.free(x)
    assert prov.precision() == "F64"
    info = prov.device_info_struct()
    assert info["arch"].startswith("gfx950") and info["compute_units"] >= 200


def test_fill_uniform_matches_oracle_bits(prov, oracle):
    h = prov.fill_uniform(42, -np.pi, np.pi, (1001, 3))
    assert bits_equal(prov.download(h), oracle.fill_uniform(42, -np.pi, np.pi, 3003))
    prov.free(h)


# ---- fused elementwise -------------------------------------------------------------------------
def _run_fused(prov, plan, out_ids, arrays, out_shape):
    # keep numpy's exact shape (a rank-1 [n] must stay rank 1 for front-padding, broadcast.rs:108-115)
    hs = [prov.upload(np.asarray(a, dtype=np.float64).reshape(-1, order="F"), np.shape(a)) for a in arrays]
    n = int(np.prod(out_shape))
    if isinstance(out_ids, int):
        sh = plan.generate_wgsl_for_output(out_ids, "f64")
        res = [prov.fused_elementwise(sh, hs, out_shape, n)]
    else:
        sh = plan.generate_wgsl_for_outputs(out_ids, "f64")
        res = prov.fused_elementwise_multi(sh, hs, out_shape, n, len(out_ids))
    outs = [prov.download_matrix(r) for r in res]
    for h in hs + res:
        prov.free(h)
    return outs if not isinstance(out_ids, int) else outs[0]


@pytest.mark.parametrize("shape", [(1024, 1024), (257, 131), (1, 1), (7, 1), (1, 9), (3, 5, 7), (2049,)])
def test_fused_sin_mul_add_vs_oracle(prov, oracle, shape):
    from planner_requests import sin_mul_add_plan

    rng = np.random.default_rng(hash(shape) % 1000)
    A = rng.uniform(-np.pi, np.pi, shape)
    B = rng.uniform(-1, 1, shape)
    C = rng.uniform(-1, 1, shape)
    plan, out = sin_mul_add_plan()
    D = _run_fused(prov, plan, out, [A, B, C], shape)
    ref = oracle.sin_mul_add(A, B, C)
    # sin within 1 ulp of libm, then exact * and + : |err| <= ulp(sin)*|B| + rounding
    assert np.max(np.abs(D - ref)) <= 2 * EPS
    # the arithmetic after sin is exact: feeding the oracle's sin through mul/add must match bit for bit
    s_dev = _run_fused(prov, *_unary_plan("sin"), [A], shape)
    assert bits_equal(D, s_dev * B + C)


def _unary_plan(name):
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    x = p.input()
    return p, p.builtin(name, x)


def test_fused_large_argument_sin(prov, oracle):
    # slow-path (Payne-Hanek) argument reduction: |x| up to 1e6 and a few huge values
    rng = np.random.default_rng(11)
    A = np.concatenate([rng.uniform(-1e6, 1e6, 4000), [1e15, -3e18, 1e300, 0.0, -0.0, np.inf, np.nan]]).reshape(-1, 1)
    got = _run_fused(prov, *_unary_plan("sin"), [A], A.shape)
    want = oracle.unary("sin", A)
    assert ulp_err(got[:-2], want[:-2]) <= 2
    assert math.isnan(got[-1, 0]) and math.isnan(got[-2, 0])
    assert math.copysign(1, got[-3, 0]) < 0  # sin(-0) = -0


def test_fused_broadcast_cases(prov, oracle):
    from planner_requests import FusionGroupPlan

    rng = np.random.default_rng(12)
    cases = [((4, 1), (1, 3), (4, 3)), ((2, 3), (2, 1), (2, 3)), ((300, 200), (1, 1), (300, 200)),
             ((1, 200), (300, 1), (300, 200)), ((5, 1, 7), (1, 6, 1), (5, 6, 7)), ((4,), (2, 3, 4), (2, 3, 4)),
             ((1, 1), (1, 1), (1, 1)), ((6, 5, 4, 3), (6, 1, 4, 1), (6, 5, 4, 3)),
             # a short leading dimension under many outer indices: the flat-thread broadcast kernels
             ((32, 5000), (1, 5000), (32, 5000)), ((32, 1), (32, 5000), (32, 5000)), ((3, 1, 70), (1, 700, 1), (3, 700, 70)),
             ((127, 64), (127, 1), (127, 64)), ((1, 64), (2, 1), (2, 64)), ((5, 7, 11, 13), (5, 1, 11, 1), (5, 7, 11, 13))]
    for sa, sb, so in cases:
        p = FusionGroupPlan()
        a, b = p.input(), p.input()
        out = p.primitive("Sub", p.primitive("ElemMul", a, b), a)
        A, B = rng.standard_normal(sa), rng.standard_normal(sb)
        got = _run_fused(prov, p, out, [A, B], so)
        want = oracle.binary("sub", oracle.binary("mul", A, B), A)
        assert want.shape == so and bits_equal(got, want), (sa, sb)


def test_fused_scalar_inputs_are_broadcast_and_constants_are_inputs(prov, oracle):
    # fusion_exec.rs:305-326: scalars arrive as 1-element tensors shaped [1,1]
    from planner_requests import elementwise_math_plan

    x = np.linspace(0.0, 4.0 * np.pi, 1024 * 1024).reshape(1024, 1024, order="F")  # BASELINE configs[0]
    plan, out = elementwise_math_plan()
    consts = [np.array([[v]]) for v in (10.0, 4.0, 0.25, 2.0, 0.1)]
    got = _run_fused(prov, plan, out, [x] + consts, x.shape)
    want = oracle.elementwise_math_chain(x)
    assert np.max(np.abs(got - want)) <= 8 * EPS  # |y2| < 1.2; a handful of <=1 ulp libm calls
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-3)) <= 1e-12


def test_fused_multi_output(prov, oracle):
    from planner_requests import FusionGroupPlan

    rng = np.random.default_rng(13)
    A, B = rng.standard_normal((129, 65)), rng.standard_normal((129, 65))
    p = FusionGroupPlan()
    a, b = p.input(), p.input()
    s = p.primitive("Add", a, b)
    m = p.primitive("ElemMul", s, b)
    d = p.primitive("ElemDiv", m, a)
    o_m, o_s, o_d = _run_fused(prov, p, [m, s, d], [A, B], A.shape)
    assert bits_equal(o_s, A + B) and bits_equal(o_m, (A + B) * B) and bits_equal(o_d, ((A + B) * B) / A)


def test_fused_exact_ops_and_policies_bitwise(prov, oracle):
    from planner_requests import FusionGroupPlan

    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.5, -1.5, 2.5, -2.5, np.inf, -np.inf, np.nan, 1e-310, 3.0, -7.25])
    X, Y = np.meshgrid(special, special, indexing="ij")
    for name in ("abs", "floor", "ceil", "round", "trunc", "fix", "sign", "sqrt", "heaviside", "isnan", "isinf", "isfinite"):
        got = _run_fused(prov, *_unary_plan(name), [X], X.shape)
        oname = {"trunc": "fix"}.get(name, name)
        assert bits_equal(got, oracle.unary(oname, X)), name
    for name in ("max", "min", "mod", "rem"):
        p = FusionGroupPlan()
        a, b = p.input(), p.input()
        out = p.builtin(name, a, b)
        got = _run_fused(prov, p, out, [X, Y], X.shape)
        assert bits_equal(got, oracle.binary(name, X, Y)), name
    for op, oname in (("Add", "add"), ("Sub", "sub"), ("ElemMul", "mul"), ("ElemDiv", "div")):
        p = FusionGroupPlan()
        a, b = p.input(), p.input()
        out = p.primitive(op, a, b)
        assert bits_equal(_run_fused(prov, p, out, [X, Y], X.shape), oracle.binary(oname, X, Y)), op


LIBM_ULP = {"sin": 2, "cos": 2, "tan": 3, "asin": 2, "acos": 2, "atan": 2, "sinh": 3, "cosh": 3, "tanh": 3,
            "asinh": 3, "acosh": 3, "atanh": 3, "exp": 2, "expm1": 2, "log": 2, "log2": 2, "log10": 2, "log1p": 2,
            "exp2": 2}


@pytest.mark.parametrize("name", sorted(LIBM_ULP))
def test_fused_libm_functions_ulp(prov, oracle, name):
    rng = np.random.default_rng(14)
    lo, hi = {"asin": (-1, 1), "acos": (-1, 1), "atanh": (-0.999, 0.999), "acosh": (1.0, 50.0), "log": (1e-6, 1e6),
              "log2": (1e-6, 1e6), "log10": (1e-6, 1e6), "log1p": (-0.999, 50.0), "exp": (-50, 50), "exp2": (-50, 50),
              "expm1": (-20, 20), "sinh": (-20, 20), "cosh": (-20, 20)}.get(name, (-10.0, 10.0))
    X = rng.uniform(lo, hi, (4096, 1))
    fused_name = {"exp2": "exp2"}.get(name, name)
    got = _run_fused(prov, *_unary_plan(fused_name), [X], X.shape)
    assert ulp_err(got, oracle.unary(name, X)) <= LIBM_ULP[name], name


def test_fused_pow_hypot_atan2(prov, oracle):
    from planner_requests import FusionGroupPlan

    rng = np.random.default_rng(15)
    X, Y = rng.uniform(0.01, 20, (4096, 1)), rng.uniform(-5, 5, (4096, 1))
    for name, prim in (("pow", "ElemPow"), ("hypot", None), ("atan2", None)):
        p = FusionGroupPlan()
        a, b = p.input(), p.input()
        out = p.primitive(prim, a, b) if prim else p.builtin(name, a, b)
        got = _run_fused(prov, p, out, [X, Y], X.shape)
        assert ulp_err(got, oracle.binary(name, X, Y)) <= 2, name


def test_fused_errors(prov):
    from runmat_amd import ProviderError
    from planner_requests import sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    sh = plan.generate_wgsl_for_output(out)
    a = prov.upload(np.ones((4, 3)))
    b = prov.upload(np.ones((5, 3)))
    with pytest.raises(ProviderError) as e:  # wrong input count
        prov.fused_elementwise(sh, [a, a], (4, 3), 12)
    assert e.value.code == 1
    with pytest.raises(ProviderError) as e:  # non-broadcastable
        prov.fused_elementwise(sh, [a, b, a], (4, 3), 12)
    assert e.value.code == 3
    with pytest.raises(ProviderError) as e:  # len mismatch
        prov.fused_elementwise(sh, [a, a, a], (4, 3), 13)
    assert e.value.code == 3
    with pytest.raises(ProviderError) as e:  # zero-length: fusion_exec.rs:273
        z = prov.upload(np.zeros((0, 3)))
        prov.fused_elementwise(sh, [z, z, z], (0, 3), 0)
    assert e.value.code == 2
    with pytest.raises(ProviderError) as e:
        prov.fused_elementwise("not wgsl", [a], (4, 3), 12)
    assert e.value.code == 6
    t = prov.telemetry_snapshot()
    assert t["fused_elementwise_count"] > 0 and t["kernel_launches"] > 0


# ---- fused reductions --------------------------------------------------------------------------
def _fused_reduce(prov, plan, data_vid, arrays, axis, reduce_len, num_slices, flavor=None, omitnan=False, is_mean=False):
    from runmat_amd import ReductionFlavor

    sh = plan.generate_reduction_wgsl(data_vid, "f64", axis=axis, omitnan=omitnan, is_mean=is_mean)
    hs = [prov.upload(a) for a in arrays]
    fl = flavor or (ReductionFlavor.Mean() if is_mean else ReductionFlavor.Sum())
    h = prov.fused_reduction(sh, hs, (num_slices,), reduce_len, num_slices, 256, fl)
    out = prov.download(h)
    for x in hs + [h]:
        prov.free(x)
    return out


def test_fused_reduction_sum_mul_kat(prov):
    # crates/runmat-accelerate/tests/fused_reduction_sum_mul.rs:40-137 (X[r,c]=r+1, W[r,c]=c+1), tol 1e-6;
    # integers => exact here
    from planner_requests import FusionGroupPlan

    rows, cols = 1000, 37
    X = np.fromfunction(lambda r, c: r + 1.0, (rows, cols))
    W = np.fromfunction(lambda r, c: c + 1.0, (rows, cols))
    p = FusionGroupPlan()
    a, b = p.input(), p.input()
    m = p.primitive("ElemMul", a, b)
    got = _fused_reduce(prov, p, m, [X, W], 0, rows, cols)
    assert np.array_equal(got, [(c + 1) * rows * (rows + 1) / 2 for c in range(cols)])


@pytest.mark.parametrize("rows,cols", [(512, 512), (1024, 1024), (33, 7), (1, 300), (300, 1), (100000, 3), (3, 100000)])
def test_fused_reduction_rows_of_sin_x_times_x(prov, oracle, rows, cols):
    # crates/runmat-vm/tests/fusion_gpu.rs:1429-1449: Y = sin(X).*X + 2; S = sum(Y, 2); also sum(Y, 1) and 'all'
    from planner_requests import FusionGroupPlan

    X = np.fromfunction(lambda r, c: ((c % 97) + 1) * 10.0 + (r % 1013) + 1, (rows, cols))
    Y = oracle.binary("add", oracle.binary("mul", oracle.unary("sin", X), X), np.array([[2.0]]))
    absY = np.abs(Y)

    def plan():
        p = FusionGroupPlan()
        x = p.input()
        two = p.constant(2.0)
        return p, p.primitive("Add", p.primitive("ElemMul", p.builtin("sin", x), x), two)

    # tolerance: summation-order bound n*eps*sum|y| plus 2 ulp per term from sin
    p, v = plan()
    s2 = _fused_reduce(prov, p, v, [X], 1, cols, rows)
    assert np.all(np.abs(s2 - oracle.reduce_sum(Y, [1]).reshape(-1)) <= (cols + 4) * EPS * absY.sum(axis=1) + 1e-300)
    p, v = plan()
    s1 = _fused_reduce(prov, p, v, [X], 0, rows, cols)
    assert np.all(np.abs(s1 - oracle.reduce_sum(Y, [0]).reshape(-1)) <= (rows + 4) * EPS * absY.sum(axis=0) + 1e-300)
    p, v = plan()
    sa = _fused_reduce(prov, p, v, [X], 0, rows * cols, 1)
    assert abs(sa[0] - oracle.reduce_sum(Y, "all")[0, 0]) <= 64 * math.sqrt(rows * cols) * EPS * absY.sum()
    p, v = plan()
    assert bits_equal(sa, _fused_reduce(prov, p, v, [X], 0, rows * cols, 1))  # deterministic


def test_fused_reduction_nan_policy_mean_and_scale(prov, oracle):
    from runmat_amd import ReductionFlavor
    from planner_requests import FusionGroupPlan

    rng = np.random.default_rng(16)
    X = rng.standard_normal((400, 9))
    X[5, 2] = np.nan
    X[:, 4] = np.nan
    p = FusionGroupPlan()
    x = p.input()
    v = p.primitive("ElemMul", x, x)
    Y = X * X
    inc = _fused_reduce(prov, p, v, [X], 0, 400, 9)
    ref = oracle.reduce_sum(Y, [0]).reshape(-1)
    assert np.isnan(inc[2]) and np.isnan(inc[4]) and np.allclose(inc[[0, 1, 3]], ref[[0, 1, 3]], rtol=1e-13)
    assert inc.view(np.uint64)[2] == 0x7FF8000000000000  # canonical quiet NaN (fusion.rs:2013-2021)
    om = _fused_reduce(prov, p, v, [X], 0, 400, 9, omitnan=True)
    refo = oracle.reduce_sum(Y, [0], omitnan=True).reshape(-1)
    assert om[4] == 0.0 and np.allclose(om, refo, rtol=1e-13)
    # mean divides by the count (CPU mean.rs:1134-1151), also for counts that f32 cannot hold
    mean = _fused_reduce(prov, p, v, [X[:, :2]], 0, 400, 2, is_mean=True)
    assert np.allclose(mean, oracle.reduce_sum(Y[:, :2], [0], mean=True).reshape(-1), rtol=1e-13)
    sc = _fused_reduce(prov, p, v, [X[:, :2]], 1, 2, 400, flavor=ReductionFlavor.CustomScale(0.125))
    assert np.allclose(sc, 0.125 * oracle.reduce_sum(Y[:, :2], [1]).reshape(-1), rtol=1e-13)
    # scalar operand uploaded as a 1-element tensor (fusion_exec.rs:522-543)
    q = FusionGroupPlan()
    a, s = q.input(), q.input()
    w = q.primitive("ElemMul", a, s)
    got = _fused_reduce(prov, q, w, [X[:, :2], np.array([[3.0]])], 0, 400, 2)
    assert np.allclose(got, 3.0 * oracle.reduce_sum(X[:, :2], [0]).reshape(-1), rtol=1e-13)


# ---- per-op kernels ----------------------------------------------------------------------------
def test_unary_binary_scalar_ops_vs_oracle(prov, oracle):
    rng = np.random.default_rng(17)
    X = rng.uniform(0.05, 3.0, (333, 77))
    Y = rng.uniform(0.05, 3.0, (333, 77))
    hx, hy = prov.upload(X), prov.upload(Y)
    exact_unary = ("sqrt", "abs", "sign", "floor", "ceil", "round", "fix", "neg", "heaviside", "isnan", "isinf", "isfinite", "uplus",
                   "single", "double")
    for op in exact_unary:
        h = getattr(prov, "unary_" + op)(hx)
        assert bits_equal(prov.download_matrix(h), oracle.unary(op, X)), op
        prov.free(h)
    for op, tol in LIBM_ULP.items():
        Xa = np.clip(X / 3.1, 0.02, 0.98) if op in ("asin", "acos", "atanh") else (X + 1.0 if op == "acosh" else X)
        ha = prov.upload(Xa)
        h = getattr(prov, "unary_" + op)(ha)
        assert ulp_err(prov.download_matrix(h), oracle.unary(op, Xa)) <= tol, op
        prov.free(h)
        prov.free(ha)
    for op in ("add", "sub", "mul", "div", "max", "min"):
        h = getattr(prov, "elem_" + op)(hx, hy)
        assert bits_equal(prov.download_matrix(h), oracle.binary(op, X, Y)), op
        prov.free(h)
    for op in ("pow", "hypot", "atan2"):
        h = getattr(prov, "elem_" + op)(hx, hy)
        assert ulp_err(prov.download_matrix(h), oracle.binary(op, X, Y)) <= 2, op
        prov.free(h)
    s = 1.75
    for op, ref in (("add", X + s), ("sub", X - s), ("mul", X * s), ("div", X / s), ("rsub", s - X), ("rdiv", s / X),
                    ("max", np.maximum(X, s)), ("min", np.minimum(X, s))):
        h = getattr(prov, "scalar_" + op)(hx, s)
        assert bits_equal(prov.download_matrix(h), ref), op
        prov.free(h)
    prov.free(hx)
    prov.free(hy)


def test_binary_broadcast_and_mismatch(prov, oracle):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(18)
    for sa, sb in [((4, 1), (1, 3)), ((2, 3), (2, 1)), ((1, 1), (40, 30)), ((7, 1, 5), (1, 6, 1)), ((1000, 1), (1, 1000)),
                   ((32, 5000), (1, 5000)), ((32, 1), (32, 5000)), ((3, 1, 70), (1, 700, 1)), ((127, 64), (127, 1)), ((2, 1), (1, 64)),
                   ((5, 7, 11, 13), (5, 1, 11, 1))]:  # the last six: short leading dimension, flat-thread kernel
        A, B = rng.standard_normal(sa), rng.standard_normal(sb)
        ha, hb = prov.upload(A), prov.upload(B)
        h = prov.elem_mul(ha, hb)
        want = oracle.binary("mul", A, B)
        assert h.shape == want.shape and bits_equal(prov.download_matrix(h), want)
        for x in (ha, hb, h):
            prov.free(x)
    ha, hb = prov.upload(np.ones((2, 3))), prov.upload(np.ones((3, 2)))
    with pytest.raises(ProviderError) as e:
        prov.elem_add(ha, hb)
    assert e.value.code == 3 and "size mismatch" in str(e.value)
    # reference KATs: crates/runmat-runtime-integration-tests/tests/gpu.rs:28-60
    a, b = prov.upload(np.array([1, 2, 3, 4.0]), (2, 2)), prov.upload(np.array([5, 6, 7, 8.0]), (2, 2))
    assert list(prov.download(prov.elem_add(a, b))) == [6, 8, 10, 12]
    assert list(prov.download(prov.elem_mul(a, b))) == [5, 12, 21, 32]


# ---- reductions --------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1000, 1000), (1, 1), (5, 1), (1, 5), (3, 70000), (70000, 3), (257, 129),
                                   (513, 37), (1001, 9), (4097, 5), (2049, 7), (4099, 3), (8191, 33)])  # odd extents >= 512 / 2048: the unaligned-pair kernels
def test_reduce_sum_mean_shapes_and_values(prov, oracle, shape):
    rng = np.random.default_rng(19)
    X = rng.uniform(-1, 1, shape)
    h = prov.upload(X)
    n = X.size
    sabs = np.abs(X).sum()
    r = prov.reduce_sum(h)
    assert r.shape == (1, 1)  # simple_provider.rs:6743
    assert abs(prov.download(r)[0] - oracle.reduce_sum(X, "all")[0, 0]) <= 64 * math.sqrt(n) * EPS * sabs
    r0 = prov.reduce_sum_dim(h, 0)
    r1 = prov.reduce_sum_dim(h, 1)
    assert r0.shape == (1, shape[1]) and r1.shape == (shape[0], 1)  # :6765, :6777
    assert np.all(np.abs(prov.download(r0) - oracle.reduce_sum(X, [0]).reshape(-1)) <= shape[0] * EPS * np.abs(X).sum(axis=0) + 1e-300)
    assert np.all(np.abs(prov.download(r1) - oracle.reduce_sum(X, [1]).reshape(-1)) <= shape[1] * EPS * np.abs(X).sum(axis=1) + 1e-300)
    m = prov.reduce_mean(h)
    assert abs(prov.download(m)[0] - oracle.reduce_sum(X, "all", mean=True)[0, 0]) <= 64 * math.sqrt(n) * EPS * sabs / n
    m1 = prov.reduce_mean_dim(h, 1)
    assert np.allclose(prov.download(m1), oracle.reduce_sum(X, [1], mean=True).reshape(-1), rtol=1e-12, atol=1e-15)
    assert prov.download(prov.reduce_min(h))[0] == X.min() and prov.download(prov.reduce_max(h))[0] == X.max()
    assert np.array_equal(prov.download(prov.reduce_max_dim(h, 0).values), X.max(axis=0))
    assert np.array_equal(prov.download(prov.reduce_min_dim(h, 1).values), X.min(axis=1))
    # determinism
    assert bits_equal(prov.download(prov.reduce_sum(h)), prov.download(r))


def test_reduce_sum_dim1_sequential_order_is_bit_exact(prov, oracle):
    # threads walk columns in ascending order (skel_reduce.h kernel B, nsplit == 1): identical to the
    # CPU's ascending accumulation (sum.rs:1031-1053) when the slice count already fills the chip
    rng = np.random.default_rng(20)
    X = rng.standard_normal((600000, 12))
    h = prov.upload(X)
    got = prov.download(prov.reduce_sum_dim(h, 1))
    assert bits_equal(got, oracle.reduce_sum(X, [1]).reshape(-1))


def test_reduce_nan_and_integers_exact(prov, oracle):
    X = np.arange(1, 100001, dtype=np.float64).reshape(1000, 100, order="F")
    h = prov.upload(X)
    assert prov.download(prov.reduce_sum(h))[0] == 100000 * 100001 / 2
    assert np.array_equal(prov.download(prov.reduce_sum_dim(h, 0)), X.sum(axis=0))
    assert np.array_equal(prov.download(prov.reduce_sum_dim(h, 1)), X.sum(axis=1))
    X[7, 3] = np.nan
    h = prov.upload(X)
    assert math.isnan(prov.download(prov.reduce_sum(h))[0])
    c = prov.download(prov.reduce_sum_dim(h, 0))
    assert math.isnan(c[3]) and np.array_equal(np.delete(c, 3), np.delete(np.nansum(X, axis=0), 3))
    assert prov.download(prov._reduce("sum", h, -1, omitnan=True))[0] == np.nansum(X)


# ---- matmul ------------------------------------------------------------------------------------
def test_matmul_reference_kats_exact(prov):
    from runmat_amd import ProviderError

    # mtimes.rs:495-503
    a = prov.upload(np.array([[1, 2, 3], [4, 5, 6.0]]))
    b = prov.upload(np.array([[7, 8], [9, 10], [11, 12.0]]))
    c = prov.matmul(a, b)
    assert c.shape == (2, 2) and np.array_equal(prov.download_matrix(c), [[58, 64], [139, 154]])
    # mtimes.rs:688-706 (column-major data)
    ha, hb = prov.upload(np.array([1, 2, 3, 4.0]), (2, 2)), prov.upload(np.array([5, 7, 6, 8.0]), (2, 2))
    assert list(prov.download(prov.matmul(ha, hb))) == [26.0, 38.0, 30.0, 44.0]
    with pytest.raises(ProviderError) as e:  # mtimes.rs:628-637 / simple_provider.rs:7709-7711
        prov.matmul(a, a)
    assert e.value.code == 3 and "inner dims must agree" in str(e.value)
    with pytest.raises(ProviderError):  # simple_provider.rs:7704-7706
        prov.matmul(prov.upload(np.ones((2, 2, 2))), b)
    # crates/runmat-accelerate/tests/matmul_small_k.rs:52-96, tol 1e-9 (small integers + quarters: exact)
    m, n, k = 64, 32, 4
    A = np.fromfunction(lambda r, c: (r + 1) + 0.25 * c, (m, k))
    B = np.fromfunction(lambda r, c: (r + 2 * c) % 7, (k, n))
    assert np.array_equal(prov.download_matrix(prov.matmul(prov.upload(A), prov.upload(B))), A @ B)


@pytest.mark.parametrize("m,k,n", [(2, 2, 2), (130, 66, 258), (258, 130, 70), (64, 1030, 200), (384, 2050, 130), (1000, 1000, 1000),
                                   (6, 18, 3), (640, 48, 1290), (256, 9000, 128), (1, 1, 1), (3, 5, 7), (131, 67, 259), (257, 1031, 129),
                                   (999, 1001, 997), (129, 17, 1), (1, 4097, 255),
                                   # A fills whole 2 MiB pages and m is a partial tile: a clamped thread reading a PAIR from the last row
                                   # would touch the page behind the buffer (found by scripts/gemm_grid.py at 32 x 8192 x 32)
                                   (32, 8192, 32), (64, 4096, 48), (96, 8192, 17), (160, 16384, 40)])
def test_matmul_ragged_shapes(prov, oracle, m, k, n):
    """Shapes that are not whole tiles - odd m, n, k and leading dimensions included - run the tile kernels with clamped operand
    loads, scalar loads for the pairs that straddle the matrix edge, a zeroed k tail and checked stores (dgemm.hip, GUARD): the
    64 x 64 kernel for few tiles and k <= 1024, the eight-wave kernel otherwise - plain, A' * B, C <- C - A * B on a view
    (preloaded C) and the epilogue store.  Neighbouring memory must stay untouched."""
    rng = np.random.default_rng(m * 7 + k * 3 + n)
    A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
    want = A @ B if m * n * k > 3e7 else oracle.matmul(A, B)
    tol = _gemm_tol(A, B, k)
    hA, hB = prov.upload(A), prov.upload(B)
    got = prov.download_matrix(prov.matmul(hA, hB))
    assert got.shape == (m, n) and np.max(np.abs(got - want)) <= tol
    gt = prov.download_matrix(prov.matmul(prov.transpose(prov.upload(A.T.copy())), hB))  # A' view read in place
    assert np.max(np.abs(gt - want)) <= tol
    # the update form on a view inside a larger buffer: a frame of sentinels around C must survive
    if m >= 4 and n >= 2:
        big = np.full((m + 4, n + 3), 7.25)
        C0 = rng.uniform(-1, 1, (m, n))
        big[2:2 + m, 1:1 + n] = C0
        hC = prov.upload(big)
        prov.blk_gemm(-1.0, (hA, 0, 0, m, k), (hB, 0, 0, k, n), 1.0, (hC, 2, 1, m, n))
        out = prov.download_matrix(hC)
        assert np.max(np.abs(out[2:2 + m, 1:1 + n] - (C0 - want))) <= tol + 4 * EPS
        frame = out.copy()
        frame[2:2 + m, 1:1 + n] = 7.25
        assert np.all(frame == 7.25)
    rs = rng.uniform(0.5, 2.0, (m, 1))
    for kw in (dict(alpha=0.5, beta=0.25, row_scale=rs, clamp_min=-0.3), dict(clamp_min=0.0, pow_exponent=1.5)):
        we, _ = oracle.matmul_epilogue(A, B, **kw) if m * n * k <= 3e7 else (None, None)
        if we is None:
            continue
        gk = {kk: (prov.upload(rs) if kk == "row_scale" else v) for kk, v in kw.items()}
        ge = prov.download_matrix(prov.matmul_epilogue(hA, hB, **gk))
        assert np.all(np.abs(ge - we) <= 3.0 * (k + 4) * EPS * (np.abs(A) @ np.abs(B)) + 1e-13), kw


def test_matmul_random_shapes_fuzz(prov):
    """120 random shapes (1 .. 700 in every dimension, biased towards tile boundaries) through matmul, A' * B, syrk and the update
    form on a view: every dispatch decision of dgemm.hip (whole tiles, guarded tiles, 64 x 64 kernel, split-K) against numpy."""
    rng = np.random.default_rng(20260927)
    edges = np.array([1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 511, 512, 513])

    def dim():
        return int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 700))

    for _ in range(120):
        m, k, n = dim(), dim(), dim()
        A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
        want = A @ B
        tol = _gemm_tol(A, B, k)
        hA, hB = prov.upload(A), prov.upload(B)
        got = prov.download_matrix(prov.matmul(hA, hB))
        assert got.shape == (m, n) and np.max(np.abs(got - want)) <= tol, (m, k, n)
        hAt = prov.upload(np.ascontiguousarray(A.T))
        gt = prov.download_matrix(prov.matmul(prov.transpose(hAt), hB))
        assert np.max(np.abs(gt - want)) <= tol, ("A'", m, k, n)
        if k <= 300:
            sy = prov.download_matrix(prov.syrk(hA))
            assert np.max(np.abs(sy - A.T @ A)) <= _gemm_tol(A.T, A, m), ("syrk", m, k)
        big = np.full((m + 3, n + 2), -3.5)
        C0 = rng.uniform(-1, 1, (m, n))
        big[1:1 + m, 2:2 + n] = C0
        hC = prov.upload(big)
        prov.blk_gemm(-1.0, (hA, 0, 0, m, k), (hB, 0, 0, k, n), 1.0, (hC, 1, 2, m, n))
        out = prov.download_matrix(hC)
        assert np.max(np.abs(out[1:1 + m, 2:2 + n] - (C0 - want))) <= tol + 4 * EPS, ("update", m, k, n)
        out[1:1 + m, 2:2 + n] = -3.5
        assert np.all(out == -3.5), ("frame", m, k, n)
        for h in (hA, hB, hAt, hC):
            prov.free(h)


def test_matmul_identity_with_asymmetric_b_detects_transposes(prov):
    n = 160
    B = np.fromfunction(lambda i, j: 3.0 * i - 7.0 * j + 0.5, (n, n))
    I = np.eye(n)
    assert np.array_equal(prov.download_matrix(prov.matmul(prov.upload(I), prov.upload(B))), B)
    assert np.array_equal(prov.download_matrix(prov.matmul(prov.upload(B), prov.upload(I))), B)


@pytest.mark.parametrize("m,k,n", [(128, 16, 128), (256, 64, 128), (384, 1024, 256), (97, 45, 33), (1, 1, 1), (1, 77, 1),
                                   (130, 17, 129), (5, 300, 7), (128, 20, 128), (127, 16, 128), (1024, 1024, 1024), (64, 0, 8)])
def test_matmul_vs_oracle_bound(prov, oracle, m, k, n):
    rng = np.random.default_rng(m * 7 + k * 3 + n)
    A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
    got = prov.download_matrix(prov.matmul(prov.upload(A), prov.upload(B)))
    want = oracle.matmul(A, B)
    # both sides are k-ordered sums; they differ only in per-term rounding: |d| <= (k+2)*eps*sum|a||b|
    bound = (k + 2) * EPS * (np.abs(A) @ np.abs(B)) + 1e-300
    assert got.shape == (m, n) and np.all(np.abs(got - want) <= bound)
    if k and max(m, n) <= 512:
        assert np.max(np.abs(got - want)) < 1e-9  # the reference's own tolerance (matmul_small_k.rs:84-96)


def test_matmul_profile_generator_shapes(prov, oracle):
    # crates/runmat-accelerate/src/bin/wgpu_profile.rs:1219-1229, abs tol 1e-5 / rel 1e-4 there
    def gen(rows, cols, base, delta):
        idx = np.arange(rows * cols)
        return (base + delta * ((idx % 128) / 127.0)).reshape((rows, cols), order="F")

    for (m, k, n) in [(256, 256, 256), (128, 2048, 128)]:
        A, B = gen(m, k, 0.5, 1.5), gen(k, n, -1.0, 2.0)
        got = prov.download_matrix(prov.matmul(prov.upload(A), prov.upload(B)))
        want = oracle.matmul(A, B)
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-12


# ---- LU / mldivide -----------------------------------------------------------------------------
def _check_lu(prov, oracle, A, tol=1e-11):
    r = prov.lu(prov.upload(A))
    comb, L, U, P, piv = oracle.lu(A)
    g_piv = prov.download(r.perm_vector)
    assert r.combined.shape == A.shape and r.lower.shape == (A.shape[0], A.shape[0]) and r.upper.shape == A.shape
    assert r.perm_matrix.shape == (A.shape[0], A.shape[0]) and r.perm_vector.shape == (A.shape[0], 1)
    assert np.array_equal(g_piv, piv.reshape(-1)), "pivot vector must be identical (integer work)"
    assert np.array_equal(prov.download_matrix(r.perm_matrix), P)
    scale = max(1.0, np.abs(comb).max())
    assert np.max(np.abs(prov.download_matrix(r.combined) - comb)) <= tol * scale
    gl, gu = prov.download_matrix(r.lower), prov.download_matrix(r.upper)
    assert np.max(np.abs(gl - L)) <= tol * scale and np.max(np.abs(gu - U)) <= tol * scale
    assert np.array_equal(np.diag(gl), np.ones(A.shape[0])) and np.array_equal(np.triu(gl, 1), np.zeros_like(gl))
    assert np.array_equal(np.tril(gu, -1), np.zeros_like(gu))
    assert np.max(np.abs(P @ A - gl @ gu)) <= 1e-12 * max(1.0, np.abs(A).max()) * A.shape[0]


@pytest.mark.parametrize("shape", [(4, 4), (200, 200), (257, 257), (64, 40), (40, 64), (1, 1), (17, 1), (1, 17), (512, 512)])
def test_lu_vs_oracle(prov, oracle, shape):
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    _check_lu(prov, oracle, rng.uniform(-1, 1, shape))


def test_lu_tie_break_singular_and_structured(prov, oracle):
    # host_lu.rs:38-47 (first max on ties), :54-59 (cut-off 1e-12)
    _check_lu(prov, oracle, np.array([[1.0, 2.0], [-1.0, 5.0]]))
    _check_lu(prov, oracle, np.array([[1e-13, 1.0], [5e-13, 2.0]]))
    _check_lu(prov, oracle, np.zeros((5, 5)))
    A = np.fromfunction(lambda i, j: ((i * 7 + j * 3) % 5) - 2.0, (48, 48))  # many exact ties / repeated values
    _check_lu(prov, oracle, A + np.eye(48) * 0.5)
    B = np.ones((20, 20))  # rank one: pivots hit the cut-off after the first column
    _check_lu(prov, oracle, B)


def test_mldivide_reference_tests(prov, oracle):
    from runmat_amd import ProviderError

    # mldivide.rs:662-680: A=[1 2;3 4], b=[5;6], residual < 1e-12
    A, b = np.array([[1.0, 2.0], [3.0, 4.0]]), np.array([[5.0], [6.0]])
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(b)))
    assert x.shape == (2, 1) and np.linalg.norm(A @ x - b) < 1e-12
    assert np.max(np.abs(x - oracle.mldivide_svd(A, b))) < 1e-12
    # scalar lhs: mldivide.rs:321-325
    s = prov.download_matrix(prov.mldivide(prov.upload(np.array([[4.0]])), prov.upload(np.array([[2.0, 8.0]]))))
    assert np.array_equal(s, [[0.5, 2.0]])
    # rank-deficient inputs: the LU / Gram paths refuse them - soft errors RMHIP_NO_SVD_PATH=1 still shows (and every system beyond 4096
    # columns gets); up to 4096 columns the Jacobi-SVD path answers with the reference's minimum-norm solution (tests/test_gpu_svdpath.py)
    for Ax, bx, code in ((np.ones((3, 2)), np.ones((3, 1)), 2), (np.array([[1.0, 2.0], [2.0, 4.0]]), np.ones((2, 1)), 7)):
        os.environ["RMHIP_NO_SVD_PATH"] = "1"
        try:
            with pytest.raises(ProviderError) as e:
                prov.mldivide(prov.upload(Ax), prov.upload(bx))
            assert e.value.code == code
        finally:
            del os.environ["RMHIP_NO_SVD_PATH"]
        xs = prov.download_matrix(prov.mldivide(prov.upload(Ax), prov.upload(bx)))
        assert np.max(np.abs(xs - oracle.mldivide_svd(Ax, bx))) < 1e-12
    with pytest.raises(ProviderError) as e:
        prov.mldivide(prov.upload(np.eye(3)), prov.upload(np.ones((2, 1))))
    assert e.value.code == 3


@pytest.mark.parametrize("m,n,nrhs", [(300, 40, 3), (1500, 200, 1), (4096, 128, 2), (40, 300, 2), (128, 1000, 1), (700, 690, 1),
                                      (5000, 7, 1), (100003, 17, 3), (20000, 30, 2), (4096, 1, 1), (70000, 24, 8)])  # regression shapes: the Gram matrix of [A | b] on the VALU kernel
def test_mldivide_rectangular_full_rank_vs_svd_oracle(prov, oracle, m, n, nrhs):
    """Rectangular A\\b on the device for full-rank A (round 2): least squares (rows > cols) / minimum norm (rows < cols)
    through the Gram matrix's LU and one refinement step; the reference answers with the SVD's pseudo-inverse solve
    (mldivide.rs:380-404), which the oracle restates.  Same solution to 1e-9 relative for cond(A) ~ 1e1..1e2."""
    rng = np.random.default_rng(m * 7 + n)
    A = rng.uniform(-1.0, 1.0, (m, n))
    B = rng.uniform(-1.0, 1.0, (m, nrhs))
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
    assert x.shape == (n, nrhs)
    want = oracle.mldivide_svd(A, B) if max(m, n) <= 1500 else np.linalg.lstsq(A, B, rcond=None)[0]
    assert np.max(np.abs(x - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))
    if m > n:  # least squares: the residual is orthogonal to the columns of A
        assert np.max(np.abs(A.T @ (A @ x - B))) <= 1e-9 * np.linalg.norm(A) * np.linalg.norm(B)
    else:  # minimum norm: exact solution that lies in the row space
        assert np.max(np.abs(A @ x - B)) <= 1e-10 * np.linalg.norm(A) * max(1.0, np.linalg.norm(x))


def test_mldivide_rectangular_reference_kat_and_guards(prov, oracle):
    from runmat_amd import ProviderError

    # mldivide.rs:681-697 `solves_least_squares`: residual norm < 1e-10
    A = np.array([1.0, 3.0, 5.0, 2.0, 4.0, 6.0]).reshape(3, 2, order="F")
    b = np.array([[7.0], [8.0], [9.0]])
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(b)))
    assert x.shape == (2, 1) and np.linalg.norm(A @ x - b) < 1e-10
    assert np.max(np.abs(x - oracle.mldivide_svd(A, b))) < 1e-10
    # B / A with a rectangular divisor goes the same way (mrdivide.rs:379-388)
    rng = np.random.default_rng(44)
    Ar, Br = rng.uniform(-1, 1, (30, 200)), rng.uniform(-1, 1, (5, 200))
    X = prov.download_matrix(prov.mrdivide(prov.upload(Br), prov.upload(Ar)))
    assert X.shape == (5, 30) and np.max(np.abs(X - np.linalg.lstsq(Ar.T, Br.T, rcond=None)[0].T)) < 1e-9
    # rank deficient or badly conditioned: the caller's CPU SVD path (soft error, counted as a fallback)
    bad = rng.uniform(-1, 1, (200, 20))
    bad[:, 7] = bad[:, 3] * 2.0
    U, _ = np.linalg.qr(rng.standard_normal((300, 30)))
    V, _ = np.linalg.qr(rng.standard_normal((30, 30)))
    ill = U @ np.diag(np.logspace(0, -8, 30)) @ V.T  # cond 1e8
    os.environ["RMHIP_NO_SVD_PATH"] = "1"  # the Gram route's own guards (with the SVD path on, both are answered: test_gpu_svdpath.py)
    try:
        with pytest.raises(ProviderError) as e:
            prov.mldivide(prov.upload(bad), prov.upload(np.ones((200, 1))))
        assert e.value.code == 2
        with pytest.raises(ProviderError) as e:
            prov.mldivide(prov.upload(ill), prov.upload(np.ones((300, 1))))
        assert e.value.code == 2
    finally:
        del os.environ["RMHIP_NO_SVD_PATH"]
    xb = prov.download_matrix(prov.mldivide(prov.upload(bad), prov.upload(np.ones((200, 1)))))
    assert np.max(np.abs(xb - oracle.mldivide_svd(bad, np.ones((200, 1))))) <= 1e-10 * np.max(np.abs(xb))
    # linsolve without hints on a rectangular system: the same solve, rcond = NaN (linsolve.rs:933-970 is the SVD solve);
    # with TRANSA the transposed system
    At, bt = rng.uniform(-1, 1, (120, 25)), rng.uniform(-1, 1, (120, 2))
    rl = prov.linsolve(prov.upload(At), prov.upload(bt))
    assert np.isnan(rl.reciprocal_condition)
    assert np.max(np.abs(prov.download_matrix(rl.solution) - np.linalg.lstsq(At, bt, rcond=None)[0])) < 1e-10
    from runmat_amd import ProviderLinsolveOptions as Opt
    rt = prov.linsolve(prov.upload(At.T.copy()), prov.upload(bt), Opt(transposed=True))
    assert np.max(np.abs(prov.download_matrix(rt.solution) - np.linalg.lstsq(At, bt, rcond=None)[0])) < 1e-10
    ok = U @ np.diag(np.logspace(0, -4, 30)) @ V.T  # cond 1e4: accepted
    rhs = rng.uniform(-1, 1, (300, 1))
    xo = prov.download_matrix(prov.mldivide(prov.upload(ok), prov.upload(rhs)))
    want = np.linalg.lstsq(ok, rhs, rcond=None)[0]
    assert np.max(np.abs(xo - want)) <= 1e-7 * np.max(np.abs(want))


@pytest.mark.parametrize("n,nrhs", [(3, 1), (64, 1), (300, 3), (1000, 1), (2048, 2), (513, 8), (640, 4), (700, 9), (129, 5)])
def test_mldivide_well_conditioned_vs_oracle(prov, oracle, n, nrhs):
    # SURVEY.md 8(d) config 5 generator: A = U(-1,1) + n*I, b = A*1
    rng = np.random.default_rng(n)
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    X = np.ones((n, nrhs)) * np.arange(1, nrhs + 1)
    B = A @ X
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
    assert np.max(np.abs(x - X)) <= 1e-9
    assert np.linalg.norm(A @ x - B) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-12 * n
    if n <= 300:
        assert np.max(np.abs(x - oracle.mldivide_svd(A, B))) <= 1e-11  # SVD solve (reference CPU algorithm)
    assert np.max(np.abs(x - oracle.mldivide_lu(A, B))) <= 1e-11 if n <= 1000 else True


def test_mldivide_general_matrix(prov, oracle):
    rng = np.random.default_rng(77)
    n = 384
    A = rng.standard_normal((n, n))
    b = rng.standard_normal((n, 1))
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(b)))
    ref = oracle.mldivide_lu(A, b)
    cond = np.linalg.cond(A)
    assert np.linalg.norm(A @ x - b) <= 1e-13 * n * np.linalg.norm(A) * np.linalg.norm(x)
    assert np.max(np.abs(x - ref)) <= 1e-14 * cond * max(1.0, np.abs(ref).max())


# ---- RNG ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 7, 1001, 65536, 1000003])
def test_rng_uniform_stream_is_bit_exact(prov, oracle, n):
    prov.set_rng_state(oracle.rng_default_seed())
    got = prov.download(prov.random_uniform((n, 1)))
    want, state = oracle.rng_uniform(oracle.rng_default_seed(), n)
    assert bits_equal(got, want)
    assert prov.get_rng_state() == state
    more = prov.download(prov.random_uniform((5, 1)))  # the stream continues where the CPU's would
    want2, _ = oracle.rng_uniform(state, 5)
    assert bits_equal(more, want2)


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 100001])
def test_rng_normal_stream(prov, oracle, n):
    prov.rng_seed(0)  # rng(0) == default seed (random.rs:128-131)
    assert prov.get_rng_state() == oracle.rng_default_seed()
    got = prov.download(prov.random_normal((n, 1)))
    want, state = oracle.rng_normal(oracle.rng_default_seed(), n)
    # same uniforms bit for bit; log/sqrt/cos/sin differ from libm by <= ~2 ulp each
    assert np.max(np.abs(got - want)) <= 1e-14 * 8.0
    assert prov.get_rng_state() == state  # odd n consumes a whole pair (random.rs:536-540)
    prov.rng_seed(12345)
    assert prov.get_rng_state() == oracle.rng_mix_seed(12345)


def _normals_80bit(oracle, state, n):
    """Box-Muller on the oracle's (bit-exact) uniform stream in 80-bit arithmetic, the angle reduced exactly."""
    L = np.longdouble
    u, _ = oracle.rng_uniform(state, n + (n & 1))
    u1, u2 = u[0::2].astype(L), u[1::2].astype(L)
    u1 = np.where(u1 <= 0, L(2.2250738585072014e-308), u1)
    r = np.sqrt(L(-2) * np.log(u1))
    t = 2 * u2
    q = np.rint(2 * t)
    x = (t - q / 2) * L("3.14159265358979323846264338327950288")
    s0, c0 = np.sin(x), np.cos(x)
    k = q.astype(np.int64) % 4
    z = np.empty(2 * len(r), dtype=L)
    z[0::2] = r * np.choose(k, [c0, -s0, -c0, s0])
    z[1::2] = r * np.choose(k, [s0, c0, -s0, -c0])
    return z[:n], np.repeat(r, 2)[:n]


def test_rng_normal_accuracy_vs_80bit_reference(prov, oracle):
    """The device's table-based log / rsq / sincos (rng.hip) against exact arithmetic on the same uniforms: every sample within
    6e-16 of its radius - tighter than the CPU's own libm chain manages (7e-16 here), so the 8e-14 stream tolerance above is
    all libm's."""
    n = 400000
    for seed in (oracle.rng_default_seed(), 0x9E3779B97F4A7C15):
        prov.set_rng_state(seed)
        got = prov.download(prov.random_normal((n, 1))).ravel()
        ref, radius = _normals_80bit(oracle, seed, n)
        err = np.abs(got.astype(np.longdouble) - ref)
        assert float((err / radius).max()) <= 6e-16


def test_rng_normal_edge_uniforms(prov, oracle):
    """States chosen so that the first pair draws the extreme uniforms: u1 = 0 (replaced by f64::MIN_POSITIVE, random.rs:281-283),
    u1 = 2^-53, u1 = 1 - 2^-53 (radius 1.5e-8: the logarithm must keep its RELATIVE accuracy there), u1 at the table's split
    and cell edges, u2 = 0 and u2 = 1 - 2^-53."""
    a, mask = 6364136223846793005, (1 << 64) - 1
    ainv = pow(a, -1, 1 << 64)
    before = lambda x: ((x - 1) * ainv) & mask  # noqa: E731  the state whose successor is x
    firsts = [0, 5, 1 << 11, ((1 << 53) - 1) << 11, (1 << 63), (1 << 63) | (1 << 62), ((1 << 52) | (54 << 45)) << 11,
              (((1 << 52) | (54 << 45)) << 11) - (1 << 11), ((1 << 52) | (1 << 44)) << 11, (((1 << 52) | (1 << 44)) << 11) - (1 << 11)]
    seconds = [0, 7, ((1 << 53) - 1) << 11, 1 << 63, (1 << 62), (1 << 62) - (1 << 11), ((1 << 44) - 1) << 11, (1 << 44) << 11]
    states = [before(x1) for x1 in firsts] + [before(before(x2)) for x2 in seconds]
    for st in states:
        prov.set_rng_state(st)
        got = prov.download(prov.random_normal((2, 1))).ravel()
        want, st2 = oracle.rng_normal(st, 2)
        assert prov.get_rng_state() == st2
        assert np.max(np.abs(got - want)) <= 8e-14
        ref, radius = _normals_80bit(oracle, st, 2)
        assert float(np.max(np.abs(got.astype(np.longdouble) - ref) / radius)) <= 6e-16


def test_rng_moments(prov):
    # crates/runmat-runtime/tests/rng.rs:19-63
    prov.rng_seed(0)
    z = prov.download(prov.random_normal((50000, 1)))
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02
    u = prov.download(prov.random_uniform((50000, 1)))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01


# ---- workloads / golden ------------------------------------------------------------------------
@pytest.mark.parametrize("case", golden_monte_carlo_cases()[:2], ids=lambda c: f"M{c['M']}_T{c['T']}")
def test_lcg_monte_carlo_vs_oracle_and_reference_golden(prov, oracle, case):
    g = lcg_monte_carlo_price(ProviderOps(prov), case["M"], case["T"])
    o = lcg_monte_carlo_price(OracleOps(oracle), case["M"], case["T"])
    assert abs(g - o) <= 1e-10 * max(1.0, abs(o))          # rel 1e-10
    assert abs(g - case["price"]) <= 2e-4 * max(1.0, abs(case["price"]))  # the reference script's own output (f32 pipeline)


def test_device_pipelines_vs_the_f64_runs_of_the_reference_scripts(prov):
    """tests/golden/reference_f64.json: the reference's numpy comparators run in f64 (make_golden.py main_f64).  The device's per-op
    kernels - the unfused path the planner takes without a fusion plan - reproduce the LCG Monte-Carlo price to 1e-10 (SURVEY.md 8(d)
    config 4) and the fused 14-op chain the script's y2 samples to a few ulp."""
    from planner_requests import elementwise_math_plan
    from workloads import golden_reference_f64

    ref = golden_reference_f64()
    for case in ref["monte_carlo_lcg"]["cases"][:2]:
        g = lcg_monte_carlo_price(ProviderOps(prov), case["M"], case["T"], f32_constants=False)
        assert abs(g - case["price"]) <= 1e-10 * max(1.0, abs(case["price"])), (g, case)
    plan, out_id = elementwise_math_plan()
    shader = plan.generate_wgsl_for_output(out_id, "f64")
    consts = [prov.upload(np.array([v]), (1, 1)) for v in (10.0, 4.0, 0.25, 2.0, 0.1)]
    for case in ref["elementwise_math"]:
        n = case["points"]
        hx = prov.upload(np.linspace(0.0, 4.0 * np.pi, n), (n, 1))
        hy = prov.fused_elementwise(shader, [hx] + consts, (n, 1), n)
        y2 = prov.download(hy)
        assert np.max(np.abs(y2[case["indices"]] - np.array(case["y2"]))) <= 2e-14
        prov.free(hx)
        prov.free(hy)
    for h in consts:
        prov.free(h)


def test_rng_monte_carlo_price_vs_oracle(prov, oracle):
    # benchmarks/monte-carlo-analysis/runmat_rng.m in f64 on the CPU-parity randn stream
    M, T = 200000, 4
    S0, mu, sigma, dt, K = 100.0, 0.05, 0.2, 1.0 / 252.0, 100.0
    drift, scale = (mu - 0.5 * sigma * sigma) * dt, sigma * math.sqrt(dt)
    prov.rng_seed(0)
    S = prov.fill((M, 1), S0)
    for _ in range(T):
        Z = prov.random_normal((M, 1))
        S = prov.elem_mul(S, prov.unary_exp(prov.scalar_add(prov.scalar_mul(Z, scale), drift)))
    price = prov.download(prov.reduce_mean(prov.scalar_max(prov.scalar_sub(S, K), 0.0)))[0] * math.exp(-mu * T * dt)
    want, state = oracle.monte_carlo_price(oracle.rng_default_seed(), M, T)
    assert abs(price - want) <= 1e-10 * want
    assert prov.get_rng_state() == state


def test_sharding_paths_on_one_gpu(prov, oracle):
    """runmat_amd.sharding with a single-rank group through the real provider: the per-op and the
    fused (planner-shaped) Monte-Carlo agree with the CPU generator's price and final RNG state."""
    from runmat_amd import sharding as sh

    g = sh.Group()
    M, T = 100001, 3
    want, want_state = oracle.monte_carlo_price(oracle.rng_default_seed(), M, T)
    p1, s1 = sh.monte_carlo_price_sharded(prov, g, M, T, rng_state=oracle.rng_default_seed())
    from planner_requests import monte_carlo_shaders

    shaders = monte_carlo_shaders(100.0)
    p2, s2 = sh.monte_carlo_price_fused(prov, g, M, T, shaders, rng_state=oracle.rng_default_seed())
    p3, s3 = sh.monte_carlo_price_evolved(prov, g, M, T, rng_state=oracle.rng_default_seed(), payoff_shader=shaders[1])  # one-call time loop
    assert s1 == want_state and s2 == want_state and s3 == want_state
    assert abs(p1 - want) <= 1e-10 * want and abs(p2 - want) <= 1e-10 * want and abs(p3 - want) <= 1e-10 * want
    # row-block matmul + device-side gather are identities at world 1
    A = np.arange(12.0).reshape(3, 4)
    h, keep = sh.gather_row_blocks_device(g, prov, prov.upload(A), 3)
    assert np.array_equal(prov.download_matrix(h), A) and keep is None
    x = prov.upload(np.linspace(0.0, 1.0, 1001))
    assert abs(sh.sum_all_sharded(prov, g, x) - oracle.reduce_sum(np.linspace(0.0, 1.0, 1001).reshape(-1, 1), "all")[0, 0]) < 1e-12


def test_host_executors_mixed_operands(prov, oracle):
    """execute_elementwise / execute_reduction (fusion_exec.rs:196-628): resident handles, host
    tensors and scalars in one request; temporaries freed; shapes from runtime broadcast."""
    from planner_requests import FusionGroupPlan
    from planner_exec import execute_elementwise, execute_reduction

    rng = np.random.default_rng(21)
    A = rng.standard_normal((64, 1))
    B = rng.standard_normal((1, 48))
    p = FusionGroupPlan()
    a, b, c = p.input(), p.input(), p.input()
    out = p.primitive("Add", p.primitive("ElemMul", a, b), c)
    before = prov.telemetry_snapshot()["bytes_pooled"]
    ha = prov.upload(A)
    (h,) = execute_elementwise(prov, p, [out], [ha, B, 0.5])
    assert h.shape == (64, 48)
    assert bits_equal(prov.download_matrix(h), A * B + 0.5)
    prov.free(h)
    prov.free(ha)
    # the two temporaries (B and the scalar, uploaded by the executor) went back to the pool; so did the operand and the result just
    # freed - whether their blocks had come out of the pool or from a fresh allocation
    assert prov.telemetry_snapshot()["bytes_pooled"] >= before
    q = FusionGroupPlan()
    x, s = q.input(), q.input()
    v = q.primitive("ElemMul", x, s)
    X = rng.standard_normal((100, 7))
    r = execute_reduction(prov, q, v, [X, 2.0], 100, 7, axis=0)
    assert r.shape == (7,) and np.allclose(prov.download(r), 2.0 * X.sum(axis=0), rtol=1e-13)


def test_dot(prov, oracle):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(22)
    a, b = rng.standard_normal((5000, 1)), rng.standard_normal((5000, 1))
    d = prov.dot(prov.upload(a), prov.upload(b))
    assert d.shape == (1, 1)
    want = oracle.reduce_sum(oracle.binary("mul", a, b), "all")[0, 0]
    assert abs(prov.download(d)[0] - want) <= 5000 * EPS * np.abs(a * b).sum()
    A, B = rng.standard_normal((300, 40)), rng.standard_normal((300, 40))
    ha, hb = prov.upload(A), prov.upload(B)
    c0 = prov.dot(ha, hb)          # first non-singleton dim = rows
    c1 = prov.dot(ha, hb, 1)
    assert c0.shape == (1, 40) and c1.shape == (300, 1)
    assert np.allclose(prov.download(c0), (A * B).sum(axis=0), rtol=1e-12, atol=1e-13)
    assert np.allclose(prov.download(c1), (A * B).sum(axis=1), rtol=1e-12, atol=1e-13)
    with pytest.raises(ProviderError) as e:
        prov.dot(ha, prov.upload(np.ones((40, 300))))
    assert e.value.code == 3


@pytest.mark.parametrize("m,k,n", [(200, 96, 136), (256, 64, 128), (384, 160, 256)], ids=["edge-tiles", "eight-wave-tile", "eight-wave-3x2"])
def test_matmul_epilogue_vs_oracle(prov, oracle, m, k, n):
    """MatmulEpilogue folded into the dgemm store (lib.rs:3498-3560; order simple_provider.rs:7800-7836).  Shapes that are whole
    128 x 128 x 16 tiles take the eight-wave kernel: the short epilogue inlined, requests with an exponent out of line (dgemm.hip)."""
    from runmat_amd import ProviderError

    rng = np.random.default_rng(23)
    A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
    rs, cs = rng.uniform(0.5, 2.0, (m, 1)), rng.uniform(0.5, 2.0, (1, n))
    ha, hb, hrs, hcs = prov.upload(A), prov.upload(B), prov.upload(rs), prov.upload(cs)
    base = (k + 4) * EPS * (np.abs(A) @ np.abs(B))
    cases = [dict(alpha=2.5, beta=-0.75), dict(row_scale=rs, row_op="divide"), dict(col_scale=cs),
             dict(alpha=0.5, beta=0.1, row_scale=rs, col_scale=cs, col_op="divide", clamp_min=-0.2, clamp_max=0.4),
             dict(clamp_min=0.0, pow_exponent=1.5), dict(alpha=1.0, beta=0.0)]
    for kw in cases:
        want, _ = oracle.matmul_epilogue(A, B, **kw)
        gk = {kk: (hrs if kk == "row_scale" else hcs if kk == "col_scale" else v) for kk, v in kw.items()}
        got = prov.download_matrix(prov.matmul_epilogue(ha, hb, **gk))
        scale = abs(kw.get("alpha", 1.0)) * 4.0 + 1.0  # scales <= 2 each way
        assert np.all(np.abs(got - want) <= scale * base + 1e-13), kw
    diag = prov.zeros((min(m, n), 1))
    want, wd = oracle.matmul_epilogue(A, B, alpha=3.0, diag=True)
    got = prov.download_matrix(prov.matmul_epilogue(ha, hb, alpha=3.0, diag_output=diag))
    assert np.all(np.abs(got - want) <= 4 * base + 1e-13)
    assert np.array_equal(prov.download(diag), np.diag(got)[: min(m, n)])
    with pytest.raises(ProviderError) as e:
        prov.matmul_epilogue(ha, hb, diag_output=prov.zeros((5, 1)))
    assert e.value.code == 3 and "diag_output length" in str(e.value)


def test_empty_and_degenerate_shapes(prov, oracle):
    """Empty tensors flow through every per-op entry point like the CPU builtins treat them."""
    e = prov.upload(np.zeros((0, 3)))
    assert prov.unary_sin(e).shape == (0, 3) and prov.download(prov.unary_sin(e)).size == 0
    assert prov.elem_add(e, e).shape == (0, 3)
    assert prov.elem_mul(e, prov.upload(np.ones((1, 3)))).shape == (0, 3)
    assert prov.scalar_mul(e, 2.0).shape == (0, 3)
    s0 = prov.reduce_sum_dim(e, 0)   # sum over an empty dim -> zeros (sum.rs: saw_value false => 0)
    assert s0.shape == (1, 3) and np.array_equal(prov.download(s0), np.zeros(3))
    s1 = prov.reduce_sum_dim(e, 1)
    assert s1.shape == (0, 1) and prov.download(s1).size == 0
    assert prov.download(prov.reduce_sum(e))[0] == 0.0
    assert math.isnan(prov.download(prov.reduce_mean(e))[0])  # mean of nothing is NaN (mean.rs:1130-1133)
    a = prov.upload(np.ones((4, 0)))
    b = prov.upload(np.ones((0, 5)))
    z = prov.matmul(a, b)             # k == 0: all-zero product
    assert z.shape == (4, 5) and np.array_equal(prov.download(z), np.zeros(20))
    assert prov.matmul(prov.upload(np.ones((0, 4))), prov.upload(np.ones((4, 2)))).shape == (0, 2)
    one = prov.upload(np.array([[3.0]]))
    assert prov.download(prov.matmul(one, one))[0] == 9.0
    assert prov.download(prov.reduce_sum_dim(one, 0))[0] == 3.0 and prov.download(prov.reduce_max(one))[0] == 3.0
    n0 = prov.random_normal((0, 1))
    assert n0.shape == (0, 1)
    same = prov.upload(np.arange(6.0).reshape(2, 3))
    assert np.array_equal(prov.download(prov.elem_mul(same, same)), (np.arange(6.0).reshape(2, 3) ** 2).reshape(-1, order="F"))


def test_block_ops_and_block_cyclic_solve_on_gpu(prov, oracle):
    """The blk_* building blocks through the C ABI, and the distributed solver's driver with a
    single-rank group (same code path as multi-GPU minus the broadcasts) against the oracle."""
    from runmat_amd import ProviderError
    from runmat_amd import sharding as sh

    rng = np.random.default_rng(24)
    M = rng.standard_normal((50, 40))
    h = prov.upload(M)
    sub = prov.blk_copy((h, 5, 7, 20, 11))
    assert np.array_equal(prov.download_matrix(sub), M[5:25, 7:18])
    prov.blk_assign((h, 0, 0, 20, 11), sub)
    M2 = M.copy()
    M2[0:20, 0:11] = M[5:25, 7:18]
    assert np.array_equal(prov.download_matrix(h), M2)
    a, b, c = prov.upload(rng.standard_normal((30, 20))), prov.upload(rng.standard_normal((20, 10))), prov.upload(np.ones((30, 10)))
    prov.blk_gemm(-1.0, (a, 2, 3, 25, 16), (b, 4, 1, 16, 8), 1.0, (c, 5, 2, 25, 8))
    want = np.ones((30, 10))
    want[5:30, 2:10] -= prov.download_matrix(a)[2:27, 3:19] @ prov.download_matrix(b)[4:20, 1:9]
    assert np.max(np.abs(prov.download_matrix(c) - want)) < 1e-13
    with pytest.raises(ProviderError) as e:
        prov.blk_copy((h, 45, 0, 10, 5))
    assert e.value.code == 3
    for n, nb, nrhs in [(300, 64, 1), (1000, 256, 2), (130, 512, 1)]:
        A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
        X = np.ones((n, nrhs)) * np.arange(1, nrhs + 1)
        B = A @ X
        y = sh.mldivide_block_cyclic(prov, sh.Group(), prov.upload(A), n, prov.upload(B), nb=nb)
        got = prov.download_matrix(y)
        assert np.max(np.abs(got - X)) <= 1e-9
        assert np.max(np.abs(got - oracle.mldivide_lu(A, B))) <= 1e-11
    Ag = rng.standard_normal((257, 257))  # general matrix: real pivoting, ragged last block
    bg = rng.standard_normal((257, 1))
    yg = prov.download_matrix(sh.mldivide_block_cyclic(prov, sh.Group(), prov.upload(Ag), 257, prov.upload(bg), nb=64))
    ref = prov.download_matrix(prov.mldivide(prov.upload(Ag), prov.upload(bg)))
    assert np.max(np.abs(yg - ref)) <= 1e-10 * max(1.0, np.abs(ref).max())
    with pytest.raises(ProviderError) as e:
        sh.mldivide_block_cyclic(prov, sh.Group(), prov.upload(np.ones((64, 64))), 64, prov.upload(np.ones((64, 1))), nb=32)
    assert e.value.code == 7


def test_unary_erf_sinc_single(prov, oracle):
    x = np.concatenate([np.linspace(-4, 4, 2001), [0.0, -0.0, 1.0, -3.0, 1e300, np.inf, -np.inf, np.nan, 0.1, 1e-310]]).reshape(-1, 1)
    h = prov.upload(x)
    assert ulp_err(prov.download_matrix(prov.unary_erf(h)), oracle.unary("erf", x)) <= 2
    got, want = prov.download_matrix(prov.unary_sinc(h)), oracle.unary("sinc", x)
    fin = np.isfinite(want)
    assert np.max(np.abs(got[fin] - want[fin])) <= 4 * EPS and np.array_equal(np.isnan(got), np.isnan(want))
    assert got[2001, 0] == 1.0 and got[2003, 0] == 0.0 and got[2004, 0] == 0.0  # sinc(0)=1, sinc(integer)=0
    with np.errstate(over="ignore"):
        want32 = x.astype(np.float32).astype(np.float64)
    assert bits_equal(prov.download_matrix(prov.unary_single(h)), want32)


# ---- linsolve / transpose --------------------------------------------------------------------------
def test_linsolve_reference_tests(prov, oracle):
    from runmat_amd import ProviderError, ProviderLinsolveOptions as Opt
    # linsolve.rs:1224-1247: LT hint
    a = np.array([3.0, -1.0, 4.0, 0.0, 2.0, 1.0, 0.0, 0.0, 5.0]).reshape(3, 3, order="F")
    r = prov.linsolve(prov.upload(a), prov.upload(np.array([[9.0], [1.0], [19.0]])), Opt(lower=True, need_rcond=True))
    assert np.max(np.abs(prov.download_matrix(r.solution)[:, 0] - [3.0, 2.0, 1.0])) < 1e-12
    assert r.reciprocal_condition == 2.0 / 5.0
    # linsolve.rs:1249-1285: LT + TRANSA=T (upper solve with A')
    a2 = np.array([3.0, 1.0, 0.0, 0.0, 4.0, 2.0, 0.0, 0.0, 5.0]).reshape(3, 3, order="F")
    b2 = np.array([[5.0], [14.0], [23.0]])
    r2 = prov.linsolve(prov.upload(a2), prov.upload(b2), Opt(lower=True, transposed=True))
    want, _ = oracle.linsolve(a2, b2, lower=True, transposed=True)
    assert np.max(np.abs(prov.download_matrix(r2.solution) - want)) < 1e-12
    # linsolve.rs:1209-1222: general square
    r3 = prov.linsolve(prov.upload(np.array([[2.0, 1.0], [1.0, 2.0]])), prov.upload(np.array([[4.0], [5.0]])))
    assert np.max(np.abs(prov.download_matrix(r3.solution)[:, 0] - [1.0, 2.0])) < 1e-12 and np.isnan(r3.reciprocal_condition)
    # soft errors: zero diagonal, RCOND threshold, rcond of a general matrix, rectangular, row mismatch
    with pytest.raises(ProviderError):
        prov.linsolve(prov.upload(np.array([[1.0, 0.0], [1.0, 0.0]])), prov.upload(np.ones((2, 1))), Opt(lower=True))
    with pytest.raises(ProviderError):
        prov.linsolve(prov.upload(np.diag([1.0, 1e-9])), prov.upload(np.ones((2, 1))), Opt(upper=True, rcond=1e-6))
    with pytest.raises(ProviderError):
        prov.linsolve(prov.upload(np.eye(2) * 2), prov.upload(np.ones((2, 1))), Opt(need_rcond=True))
    os.environ["RMHIP_NO_SVD_PATH"] = "1"
    try:
        with pytest.raises(ProviderError):
            prov.linsolve(prov.upload(np.ones((3, 2))), prov.upload(np.ones((3, 1))))
    finally:
        del os.environ["RMHIP_NO_SVD_PATH"]
    with pytest.raises(ProviderError):
        prov.linsolve(prov.upload(np.eye(3)), prov.upload(np.ones((2, 1))), Opt(lower=True))
    with pytest.raises(ProviderError):  # scalar operands stay on the host (linsolve.rs:408-412)
        prov.linsolve(prov.upload(np.array([[2.0]])), prov.upload(np.ones((1, 1))), Opt(lower=True))


@pytest.mark.parametrize("n,nrhs", [(2, 1), (31, 3), (32, 1), (33, 5), (257, 2), (1000, 7)])
@pytest.mark.parametrize("kind", ["lower", "upper", "lower_t", "upper_t"])
def test_linsolve_triangular_vs_oracle(prov, oracle, n, nrhs, kind):
    from runmat_amd import ProviderLinsolveOptions as Opt
    rng = np.random.default_rng(n * 7 + nrhs)
    full = rng.uniform(-1, 1, (n, n)) / max(n, 1) + np.diag(rng.uniform(1.0, 2.0, n) * rng.choice([-1.0, 1.0], n))
    b = rng.uniform(-1, 1, (n, nrhs))
    lower, trans = kind.startswith("lower"), kind.endswith("_t")
    # the unused triangle holds garbage on purpose: the solve must not read it
    opts = Opt(lower=lower, upper=not lower, transposed=trans, need_rcond=True)
    r = prov.linsolve(prov.upload(full), prov.upload(b), opts)
    want, rc = oracle.linsolve(full, b, lower=lower, upper=not lower, transposed=trans)
    got = prov.download_matrix(r.solution)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 64 * EPS * max(1.0, np.max(np.abs(want))) * n  # tolerance: summation order differs
    assert r.reciprocal_condition == rc


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (64, 64), (65, 127), (1000, 3), (1, 777), (513, 1025)])
def test_transpose_bit_exact(prov, oracle, shape):
    a = np.random.default_rng(shape[0]).uniform(-1, 1, shape)
    got = prov.download_matrix(prov.transpose(prov.upload(a)))
    assert got.shape == (shape[1], shape[0]) and bits_equal(got, oracle.transpose(a))


def test_lu_conservative_retry(oracle):
    """If the persistent panel kernels report non-co-resident workgroups the solve refactors a fresh copy with the
    one-launch-per-column panels (RMHIP_LU_TEST_RETRY forces that report once per context)."""
    import os
    from runmat_amd import HipProvider
    p2 = HipProvider(0)
    rng = np.random.default_rng(5)
    n = 300
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    B = rng.uniform(-1, 1, (n, 2))
    os.environ["RMHIP_LU_TEST_RETRY"] = "1"
    try:
        x = p2.download_matrix(p2.mldivide(p2.upload(A), p2.upload(B)))
    finally:
        del os.environ["RMHIP_LU_TEST_RETRY"]
    assert np.max(np.abs(x - oracle.mldivide_lu(A, B))) <= 1e-12
    x2 = p2.download_matrix(p2.mldivide(p2.upload(A), p2.upload(B)))  # the context stays conservative and keeps working
    assert bits_equal(x, x2)


@pytest.mark.parametrize("n,steps", [(2, 3), (3, 4), (1, 1), (1001, 7), (4096, 16), (65537, 2)])
def test_stochastic_evolution_vs_oracle(prov, oracle, n, steps):
    seed = 0x1234567 + n
    x = np.linspace(0.5, 2.0, n).reshape(n, 1)
    prov.set_rng_state(seed)
    got = prov.download_matrix(prov.stochastic_evolution(prov.upload(x), 0.001, 0.02, steps))
    want, st = oracle.stochastic_evolution(seed, x, 0.001, 0.02, steps)
    assert np.max(np.abs(got - want) / np.abs(want)) <= 64 * steps * EPS  # libm rounding of log/sqrt/sincos/exp per step
    assert prov.get_rng_state() == st  # integer stream position: exact
    # the stream continues exactly where the CPU's would
    nxt = prov.download(prov.random_uniform((4, 1)))
    assert np.array_equal(nxt, oracle.rng_uniform(st, 4)[0])


def test_stochastic_evolution_reference_tests(prov, oracle):
    # accelerate/tests/stochastic_evolution.rs:17-47: zero scale, tolerance 1e-9
    out = prov.download(prov.stochastic_evolution(prov.upload(np.array([[1.0], [2.0], [3.0]])), 0.05, 0.0, 4))
    assert np.max(np.abs(out - np.array([1.0, 2.0, 3.0]) * np.exp(0.2))) < 1e-9
    h = prov.upload(np.array([[1.0, 2.0]]))
    st = prov.get_rng_state()
    same = prov.download(prov.stochastic_evolution(h, 0.1, 0.3, 0))  # steps == 0: unchanged, nothing drawn
    assert np.array_equal(same, [1.0, 2.0]) and prov.get_rng_state() == st


@pytest.mark.parametrize("shape,dims", [((6, 5, 4), [1, 2]), ((6, 5, 4), [0, 2]), ((3, 64, 48), [1, 2]), ((7, 9), [0, 1]),
                                        ((8, 16, 16), [2, 1, 1, 7])])
def test_reduce_mean_nd_vs_cpu_order(prov, oracle, shape, dims):
    """mean(x, vecdim): the CPU takes the dims one after the other in ascending order (mean.rs:1107-1116)."""
    x = np.random.default_rng(sum(shape)).uniform(-1, 1, shape)
    got = prov.download(prov.reduce_mean_nd(prov.upload(x), dims))
    want = x
    for d in sorted({d for d in dims if d < x.ndim}):
        want = oracle.reduce_sum(want, [d], mean=True)
    assert got.shape[0] == want.size
    assert np.max(np.abs(got - want.reshape(-1, order="F"))) <= 8 * EPS
    with pytest.raises(Exception):
        prov.reduce_mean_nd(prov.upload(x), [17])  # nd.rs:69-72: no valid dims


# ---- transpose views, A'*B / A*B', syrk ---------------------------------------------------------------
def _gemm_tol(a, b, k):
    return (k + 2) * EPS * np.max(np.abs(a) @ np.abs(b))


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 512), (5, 7, 3), (130, 257, 75), (1, 9, 33), (64, 1, 17), (300, 2, 1000)])
def test_matmul_with_transpose_views(prov, oracle, m, n, k):
    rng = np.random.default_rng(m * 31 + n * 7 + k)
    At, B = rng.uniform(-1, 1, (k, m)), rng.uniform(-1, 1, (k, n))   # op(A) = At'
    A, Bt = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (n, k))   # op(B) = Bt'
    hAt, hB, hA, hBt = prov.upload(At), prov.upload(B), prov.upload(A), prov.upload(Bt)
    vA = prov.transpose(hAt)            # m x k view, no data moved
    vB = prov.transpose(hBt)            # k x n view
    assert vA.shape == (m, k) and vB.shape == (k, n)
    tn = prov.download_matrix(prov.matmul(vA, hB))
    assert np.max(np.abs(tn - oracle.matmul(At.T.copy(), B))) <= _gemm_tol(At.T, B, k)
    nt = prov.download_matrix(prov.matmul(hA, vB))
    assert np.max(np.abs(nt - oracle.matmul(A, Bt.T.copy()))) <= _gemm_tol(A, Bt.T, k)
    tt = prov.download_matrix(prov.matmul(vA, prov.transpose(prov.upload(rng.uniform(-1, 1, (n, k))))))
    assert tt.shape == (m, n)
    # views read back as the transposed matrix, and a view of a view is the base again
    assert bits_equal(prov.download_matrix(vA), At.T) and bits_equal(prov.download_matrix(prov.transpose(prov.transpose(hB))), B)
    # other consumers see a materialised copy
    assert bits_equal(prov.download_matrix(prov.unary_neg(prov.transpose(hBt))), -Bt.T)
    assert bits_equal(prov.download_matrix(prov.elem_add(prov.transpose(hAt), hA)), At.T + A)


@pytest.mark.parametrize("rows,cols", [(16, 5), (1000, 3), (257, 129), (128, 256), (4096, 64)])
def test_syrk_vs_oracle(prov, oracle, rows, cols):
    if (rows, cols) == (16, 5):  # accelerate/tests/syrk.rs:55-100 (tolerance there: 1e-9)
        a = np.array([[r + 1 + 3 * c for c in range(cols)] for r in range(rows)], dtype=np.float64)
    else:
        a = np.random.default_rng(rows + cols).uniform(-1, 1, (rows, cols))
    got = prov.download_matrix(prov.syrk(prov.upload(a)))
    want = oracle.syrk(a)
    assert got.shape == (cols, cols)
    assert np.max(np.abs(got - want)) <= max(1e-9 if rows == 16 else 0.0, _gemm_tol(a.T, a, rows))
    assert np.array_equal(got, got.T)  # commutative products, identical k order: exactly symmetric


@pytest.mark.parametrize("m,n,k", [(130, 70, 20000), (128, 128, 16384), (1, 1, 100000), (256, 128, 9000)])
def test_matmul_split_k(prov, oracle, m, n, k):
    """Few output tiles and a long k run split over k with an ordered reduction of the partial products."""
    rng = np.random.default_rng(k + m)
    A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
    hA, hB = prov.upload(A), prov.upload(B)
    got = prov.download_matrix(prov.matmul(hA, hB))
    want = A @ B if m * n * k > 5e7 else oracle.matmul(A, B)   # the naive oracle loop is slow for the larger cases
    assert np.max(np.abs(got - want)) <= _gemm_tol(A, B, k)
    assert bits_equal(got, prov.download_matrix(prov.matmul(hA, hB)))  # fixed split count and order: deterministic
    gt = prov.download_matrix(prov.matmul(prov.transpose(prov.upload(A.T.copy())), hB))  # A' view, split over k
    assert np.max(np.abs(gt - want)) <= _gemm_tol(A, B, k)
    if m == 130:
        s = prov.download_matrix(prov.syrk(prov.upload(B)))       # 20000 x 70 -> 70 x 70
        assert np.max(np.abs(s - B.T @ B)) <= _gemm_tol(B.T, B, k) and np.array_equal(s, s.T)


# ---- special fusion-pattern hooks: image_normalize, matmul_power_step -----------------------------------
@pytest.mark.parametrize("shape", [(3, 4, 5), (1, 7, 9), (16, 64, 48), (5, 33, 17), (256, 8, 8), (7, 1, 1),
                                   (1, 64, 50), (3, 31, 20), (5, 7, 9), (1, 300, 201), (15, 16, 9),  # 16-byte vectors over an odd batch extent / none for an odd count
                                   (300, 16, 16), (1001, 9, 7), (4096, 8, 8), (257, 3, 3), (70000, 2, 1)])  # more than 256 planes
@pytest.mark.parametrize("opts", [dict(gain=1.05, bias=-0.02, gamma=1.8, clamp_zero=True), dict(clamp_zero=False), dict(gain=2.0)])
def test_image_normalize_vs_oracle(prov, oracle, shape, opts):
    if shape == (3, 4, 5):  # accelerate/tests/image_normalize.rs:74-92
        b, h, w = np.meshgrid(np.arange(3), np.arange(4), np.arange(5), indexing="ij")
        x = b + 0.1 * h + 0.01 * w
    else:
        x = np.random.default_rng(sum(shape)).uniform(0.0, 1.0, shape)
    got = prov.download(prov.image_normalize(prov.upload(x), *shape, 1e-6, **opts)).reshape(shape, order="F")
    want = oracle.image_normalize(x, 1e-6, **opts)
    # tolerance: the plane sums are tree-ordered here, sequential on the CPU; pow amplifies by gamma
    assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, np.max(np.abs(want)))


def test_image_normalize_errors_and_degenerate(prov):
    from runmat_amd import ProviderError
    h = prov.upload(np.ones((2, 3, 3)))
    assert np.array_equal(prov.download(prov.image_normalize(h, 2, 3, 3, 0.0, clamp_zero=False)), np.zeros(18))  # sigma == 0
    for bad in (dict(epsilon=float("nan")), dict(epsilon=-1.0)):
        with pytest.raises(ProviderError):
            prov.image_normalize(h, 2, 3, 3, bad["epsilon"])
    with pytest.raises(ProviderError):
        prov.image_normalize(h, 3, 3, 2, 1e-6)          # descriptor dims do not match
    with pytest.raises(ProviderError):
        prov.image_normalize(prov.upload(np.ones((4, 4))), 4, 4, 1, 1e-6)  # not 3-D


@pytest.mark.parametrize("m,k,n", [(2, 2, 2), (64, 32, 8), (1000, 64, 5), (257, 129, 33)])
def test_matmul_power_step_vs_oracle(prov, oracle, m, k, n):
    rng = np.random.default_rng(m + k + n)
    A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
    got = prov.download_matrix(prov.matmul_power_step(prov.upload(A), prov.upload(B), 1e-12))
    want = oracle.matmul_power_step(A, B, 1e-12)
    assert np.max(np.abs(got - want)) <= 64 * (k + m) * EPS
    assert np.max(np.abs((got * got).sum(axis=0) - 1.0)) < 1e-9


@pytest.mark.parametrize("rows,cols", [(4, 3), (1000, 7), (257, 129), (20000, 64), (1, 5), (2, 2),
                                       (5000, 1), (4096, 8), (100003, 17), (70001, 32), (9000, 24), (300000, 3), (40000, 33)])  # many samples of <= 32 variables: the VALU Gram kernel
@pytest.mark.parametrize("biased", [False, True])
def test_covariance_vs_oracle(prov, oracle, rows, cols, biased):
    x = np.random.default_rng(rows * 3 + cols).uniform(-1, 1, (rows, cols)) + np.arange(cols)
    got = prov.download_matrix(prov.covariance(prov.upload(x), biased=biased))
    want = oracle.covariance(x, biased)
    assert got.shape == (cols, cols) and np.array_equal(np.isnan(got), np.isnan(want))
    fin = np.isfinite(want)
    if fin.any():  # tolerance: tree-ordered column means and MFMA-ordered products vs the sequential CPU loops
        assert np.max(np.abs(got[fin] - want[fin])) <= 64 * EPS * (1.0 + np.max(np.abs(want[fin]))) * np.sqrt(rows)
        assert np.array_equal(got, got.T)


def test_covariance_nonfinite_and_unsupported(prov, oracle):
    from runmat_amd import ProviderError
    x = np.random.default_rng(9).uniform(-1, 1, (50, 4))
    x[7, 2] = np.inf
    got = prov.download_matrix(prov.covariance(prov.upload(x)))
    want = oracle.covariance(x)
    assert np.array_equal(np.isnan(got), np.isnan(want))  # the poisoned column's pairs are NaN, the others finite
    y = np.random.default_rng(10).uniform(-1, 1, (30000, 11))  # the same through the tall-skinny kernel, and its syrk form
    y[12345, 9] = np.inf
    y[77, 3] = np.nan
    got, want = prov.download_matrix(prov.covariance(prov.upload(y))), oracle.covariance(y)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.allclose(got[np.isfinite(want)], want[np.isfinite(want)], rtol=1e-11, atol=1e-13)
    z = np.round(np.random.default_rng(11).uniform(-8, 8, (50001, 19)) * 4) / 4  # quarters: the Gram matrix is exact
    g = prov.download_matrix(prov.syrk(prov.upload(z)))
    assert np.array_equal(g, z.T @ z)
    with pytest.raises(ProviderError):
        prov.covariance(prov.upload(x), weights=prov.upload(np.ones((50, 1))))


def test_diag_extract(prov):
    from runmat_amd import ProviderError
    a = np.arange(20.0).reshape(4, 5)
    for off in (0, 1, 3, 4, 5, -1, -3, -4):
        got = prov.download(prov.diag_extract(prov.upload(a), off))
        assert np.array_equal(got, np.diagonal(a, off))
    with pytest.raises(ProviderError):
        prov.diag_extract(prov.upload(np.ones((5, 1))), 0)  # "diag: matrix input required"


def test_image_normalize_golden_from_reference_script(prov):
    """The device pipeline against the MSE printed by the reference's own benchmark comparator
    (tests/golden/image_normalize_lcg.json, generated by tests/golden/make_golden.py)."""
    from workloads import golden_image_cases, lcg_image_field
    g = golden_image_cases()
    p = g["params"]
    f32 = lambda v: float(np.float32(v))
    for case in g["cases"]:
        imgs = lcg_image_field(case["B"], case["H"], case["W"], p["seed"])
        h = prov.upload(imgs.reshape(-1, order="F"), imgs.shape)
        out = prov.download(prov.image_normalize(h, case["B"], case["H"], case["W"], f32(p["eps0"]), gain=f32(p["gain"]),
                                                 bias=f32(p["bias"]), gamma=f32(p["gamma"]), clamp_zero=True))
        mse = float(np.mean((out - imgs.reshape(-1, order="F")) ** 2))
        assert abs(mse - case["mse"]) <= 2e-5 * case["mse"], (mse, case["mse"])


def test_comparisons_and_logicals_bit_exact(prov, oracle):
    """elem_eq/ne/lt/le/gt/ge, logical_and/or/xor/not (lib.rs:1939-2068): IEEE comparisons, non-zero tests, broadcast."""
    special = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 2.5])
    a = special.reshape(-1, 1)
    b = special.reshape(1, -1)            # 8 x 1 against 1 x 8: every pair, through the broadcast kernel
    big = np.random.default_rng(3).integers(-2, 3, (257, 129)).astype(np.float64)
    big2 = np.random.default_rng(4).integers(-2, 3, (257, 129)).astype(np.float64)
    for name in ("eq", "ne", "lt", "le", "gt", "ge", "and", "or", "xor"):
        f = getattr(prov, ("elem_" if name in ("eq", "ne", "lt", "le", "gt", "ge") else "logical_") + name)
        assert bits_equal(prov.download_matrix(f(prov.upload(a), prov.upload(b))), oracle.binary(name, a, b)), name
        assert bits_equal(prov.download_matrix(f(prov.upload(big), prov.upload(big2))), oracle.binary(name, big, big2)), name
    assert bits_equal(prov.download_matrix(prov.logical_not(prov.upload(a))), oracle.unary("not", a))
    nan_row = prov.download_matrix(prov.elem_eq(prov.upload(a), prov.upload(b)))[4]
    assert np.array_equal(nan_row, np.zeros(8))           # NaN == anything is false ...
    assert prov.download_matrix(prov.elem_ne(prov.upload(a), prov.upload(b)))[4].all()   # ... and != is true
    assert prov.download_matrix(prov.logical_and(prov.upload(np.array([[np.nan]])), prov.upload(np.array([[1.0]]))))[0, 0] == 1.0


@pytest.mark.parametrize("shape,dims", [((6, 5, 4), [0, 2]), ((300, 40), [0]), ((300, 40), [0, 1]), ((17, 1, 9), [2]), ((64, 64, 3), [0, 1])])
def test_reduce_moments_nd_vs_oracle(prov, oracle, shape, dims):
    """`reduce_moments_nd` (lib.rs:2770-2778): E[x] and E[x^2] == the CPU's mean(x, dims) and mean(x.^2, dims)."""
    from runmat_amd import ProviderError

    rng = np.random.default_rng(len(shape) * 100 + sum(dims))
    X = rng.standard_normal(shape)
    mean, ex2 = prov.reduce_moments_nd(prov.upload(X), dims)
    want_shape = tuple(1 if d in dims else s for d, s in enumerate(shape))
    assert mean.shape == want_shape and ex2.shape == want_shape
    wm, w2 = X, oracle.binary("mul", X, X)
    for d in sorted(dims):  # mean of means in ascending dim order (mean.rs:1107-1116)
        wm, w2 = oracle.reduce_sum(wm, [d], mean=True), oracle.reduce_sum(w2, [d], mean=True)
    assert np.allclose(prov.download_matrix(mean), wm, rtol=1e-13, atol=1e-15)
    assert np.allclose(prov.download_matrix(ex2), w2, rtol=1e-13, atol=1e-15)
    var = prov.download_matrix(ex2) - prov.download_matrix(mean) ** 2
    assert np.all(var > -1e-12)
    Xn = X.copy()
    Xn.reshape(-1)[0] = np.nan
    mn, e2 = prov.reduce_moments_nd(prov.upload(Xn), dims)
    assert np.isnan(prov.download(mn)[0]) and np.isnan(prov.download(e2)[0])
    with pytest.raises(ProviderError):
        prov.reduce_moments_nd(prov.upload(np.zeros((0, 3))), [0])


# ---- special-function unary hooks (lib.rs:2089-2118, 2319) ---------------------------------------------------------
@pytest.mark.parametrize("op", ["gamma", "gammaln", "factorial", "nextpow2", "erfcinv"])
def test_special_unary_vs_oracle(prov, oracle, op):
    rng = np.random.default_rng(11)
    if op == "gamma":
        x = np.concatenate([rng.uniform(-20.5, 30.0, 4000), [5.0, 0.5, -0.5, 0.0, -3.0, -1e-10, 1.0, 170.5, 171.7, 180.0, -170.3,
                                                             np.nan, np.inf, -np.inf, 1e-300, -2.0 + 1e-13, 3.0 + 1e-9]])
    elif op == "gammaln":
        x = np.concatenate([rng.uniform(0.0, 300.0, 4000), 10.0 ** rng.uniform(-310, 10, 500), [0.0, 0.5, 1.0, 2.0, 171.0, np.inf, np.nan, 1e-306, 0.49999]])
    elif op == "factorial":
        x = np.concatenate([np.arange(0.0, 175.0), [2.5, -1.0, np.inf, -np.inf, np.nan, 3.0 + 4e-16, 3.0 + 1e-12, 1e6]])
    elif op == "nextpow2":
        x = np.concatenate([rng.uniform(-1e6, 1e6, 2000), 2.0 ** rng.integers(-1000, 1000, 200), [0.0, -0.0, 1.0, 9.0, -3.0, np.inf, np.nan, 5e-324]])
    else:
        x = np.concatenate([rng.uniform(0.0, 2.0, 3000), 10.0 ** rng.uniform(-320, 0, 500), [0.0, 1.0, 2.0, 0.3, 0.5, 1.5, 1e-100, -0.1, 2.1, np.nan,
                                                                                               0.999999999999, 1.000000000001, 5e-324]])
    x = x.reshape(-1, 1)
    got = prov.download(getattr(prov, "unary_" + op)(prov.upload(x)))
    want = oracle.unary(op, x).reshape(-1)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan) and np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    assert np.array_equal(np.sign(got[fin]), np.sign(want[fin]))
    if op in ("factorial", "nextpow2"):
        assert np.array_equal(got[fin], want[fin])  # products / ceil(log2): exact
        assert np.array_equal(got[~fin & ~nan], want[~fin & ~nan])
    else:
        # same formulas, different libm (ocml vs glibc): pow / exp / log / sin / erfc differ by an ulp or two, amplified by
        # the exponent of t^(z-1/2) (gamma), by cancellation near the zeros of gammaln at 1 and 2, and by the slope of
        # erfc at the bisection's fixed point
        rel = np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-300)
        if op == "gamma":
            assert np.max(rel) <= 2e-13
        elif op == "gammaln":
            assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0)) <= 1e-13
        else:
            # The bisection stops somewhere inside the interval on which erfc rounds to the target, and that interval is
            # wide where erfc is flat in ulps of its value (near 1: |dy| ~ 1e-16 absolute; subnormal targets: a few 1e-4),
            # so values are compared where the problem is well conditioned and everywhere through the forward map, as the
            # reference's own test does (erfcinv.rs `scalar_values_match_reference_points`: 2e-16 absolute near 1).
            from scipy.special import erfc

            xs = x.reshape(-1)[fin]
            well = (xs > 1e-300) & (np.abs(xs - 1.0) > 1e-3)
            assert np.max(np.abs(got[fin][well] - want[fin][well]) / np.abs(want[fin][well])) <= 1e-13
            assert np.max(np.abs(got[fin] - want[fin])[np.abs(xs - 1.0) <= 1e-3]) <= 4e-16
            back = erfc(got[fin])  # d(erfc)/erfc = -2y dy/y * y: an ulp of y is up to 2 y^2 (~1500 at y = 27) ulps of erfc(y)
            normal = xs >= 2.3e-308
            bad = np.abs(back - xs) > np.maximum(3e-10 * xs, 2.0 * np.spacing(xs))
            assert not np.any(bad & normal), (xs[bad & normal][:5], got[fin][bad & normal][:5], want[fin][bad & normal][:5])
            # subnormal targets (a handful of significant bits; erfcinv.rs `tiny_tail_inputs_remain_ordered_and_finite`
            # asks for finite, ordered, below 32): the two erfc implementations round differently down there
            sub = ~normal
            assert np.all(np.isfinite(got[fin][sub])) and np.all(got[fin][sub] < 32.0) and np.all(got[fin][sub] > 26.5)
            assert np.max(np.abs(got[fin][sub] - want[fin][sub])) <= 2e-3


def test_mrdivide_and_solve_telemetry(prov, oracle):
    """`mrdivide` (lib.rs:2484) with the reference's KATs (mrdivide.rs `solves_square_system`, `divides_matrix_by_scalar`,
    `reports_dimension_mismatch`), a larger system vs the oracle, and the solve counters / fallback reasons /
    kernel-launch log of `ProviderTelemetry` (lib.rs:1337-1357)."""
    from runmat_amd import ProviderError

    prov.reset_telemetry()
    a = np.array([1.0, 3.0, 2.0, 4.0]).reshape(2, 2, order="F")
    b = np.array([5.0, 7.0, 6.0, 8.0]).reshape(2, 2, order="F")
    x = prov.download_matrix(prov.mrdivide(prov.upload(a), prov.upload(b)))
    assert np.max(np.abs(x.reshape(-1, order="F") - [3.0, 2.0, -2.0, -1.0])) < 1e-12
    s = prov.download_matrix(prov.mrdivide(prov.upload(np.array([[2.0, 4.0, 6.0]])), prov.upload(np.array([[2.0]]))))
    assert np.array_equal(s, [[1.0, 2.0, 3.0]])
    with pytest.raises(ProviderError) as e:  # column counts must agree (mrdivide.rs:327)
        prov.mrdivide(prov.upload(np.ones((1, 2))), prov.upload(np.ones((3, 1))))
    assert e.value.code == 3
    rng = np.random.default_rng(9)
    n, m = 300, 17
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    B = rng.uniform(-1, 1, (m, n))
    X = prov.download_matrix(prov.mrdivide(prov.upload(B), prov.upload(A)))
    assert X.shape == (m, n) and np.max(np.abs(X - oracle.mrdivide(B, A))) <= 1e-12
    assert np.linalg.norm(X @ A - B) <= 1e-12 * n * np.linalg.norm(A) * np.linalg.norm(X)
    # soft failures are counted by reason (telemetry.rs:95-99).  Singular / rank-deficient systems up to 4096 columns are answered on
    # the device by the Jacobi-SVD path (tests/test_gpu_svdpath.py); RMHIP_NO_SVD_PATH=1 shows the hand-back every larger system gets
    os.environ["RMHIP_NO_SVD_PATH"] = "1"
    try:
        with pytest.raises(ProviderError):
            prov.mrdivide(prov.upload(np.ones((2, 3))), prov.upload(np.ones((4, 3))))  # rectangular divisor: CPU least squares
        with pytest.raises(ProviderError):
            prov.mldivide(prov.upload(np.array([[1.0, 2.0], [2.0, 4.0]])), prov.upload(np.ones((2, 1))))
        with pytest.raises(ProviderError):
            prov.mldivide(prov.upload(np.ones((3, 2))), prov.upload(np.ones((3, 1))))
    finally:
        del os.environ["RMHIP_NO_SVD_PATH"]
    prov.linsolve(prov.upload(A), prov.upload(B.T.copy()))
    t = prov.telemetry_snapshot()
    assert t["mrdivide_count"] == 5 and t["mldivide_count"] == 2 and t["linsolve_count"] == 1 and t["mrdivide_ns"] > 0
    fb = dict(t["solve_fallbacks"])
    assert fb == {"mrdivide:unsupported": 1, "mldivide:singular": 1, "mldivide:unsupported": 1}
    # kernel-launch log: names and attribute keys of the reference's wgpu provider (ops/telemetry.rs:26-34, 140-146; helpers.rs:36-50)
    from planner_requests import FusionGroupPlan, sin_mul_add_plan
    from runmat_amd.provider import ReductionFlavor

    prov.reset_telemetry()
    plan, out_id = sin_mul_add_plan()
    hs = [prov.upload(rng.uniform(-1, 1, (33, 5))) for _ in range(3)]
    prov.fused_elementwise(plan.generate_wgsl_for_output(out_id, "f64"), hs, (33, 5), 165)
    red = FusionGroupPlan()
    v = red.primitive("ElemMul", red.input(), red.input())
    prov.fused_reduction(red.generate_reduction_wgsl(v, "f64", axis=0), hs[:2], (5,), 33, 5, 256, ReductionFlavor.Mean())
    prov.matmul(hs[0], prov.transpose(hs[1]))
    log = prov.telemetry_snapshot()["kernel_launches_log"]
    assert [r["kernel"] for r in log] == ["fused_elementwise", "fused_reduction", "matmul"]
    assert log[0]["precision"] == "f64" and log[0]["shape"] == {"len": 165, "inputs": 3, "rank": 2} and "wg" in log[0]["tuning"]
    assert log[1]["shape"] == {"reduce_len": 33, "slices": 5, "rank": 1} and log[1]["tuning"]["flavor"] == 1
    assert log[2]["shape"] == {"m": 33, "n": 33, "k": 5} and log[2]["tuning"]["tb"] == 1
    for _ in range(70):  # bounded log: newest last, oldest dropped
        prov.free(prov.matmul(hs[0], prov.transpose(hs[1])))
    log = prov.telemetry_snapshot()["kernel_launches_log"]
    assert len(log) == 64 and all(r["kernel"] == "matmul" for r in log)
