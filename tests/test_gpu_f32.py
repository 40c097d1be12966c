"""`ProviderPrecision::F32` (crates/runmat-accelerate-api/src/lib.rs:815-818) through the C ABI.

Contract under test (include/rmhip.h, rmhip_set_precision): the host boundary stays f64 (`HostTensorView`,
lib.rs:3362-3372); tensors live in HBM as f32; arithmetic is f64 in registers and rounded once on store -- what the
CPU path does for `single` arrays (f64 storage pre-rounded through f32 after every builtin,
builtins/math/elementwise/times.rs:750-760).  Two checks per op:

  * bit-exact against the F64 provider: the same f64 kernels run on the f32-rounded inputs, rounded once
    (`f32r(F64(f32r(x)))`), so any f32-storage indexing / vector-tail / conversion bug shows up as a mismatch
    (generated f32 kernels evaluate sin / cos with skel_common.h's shorter form: same <= 1 ulp-of-f64 error, i.e. the
    same f32 value except within 2^-28 of an f32 rounding boundary - `test_f32_generated_sin_cos_whole_range`);
  * against the oracle's CPU semantics with the tolerance the reference's own F32 provider tests use
    (1e-5 absolute, e.g. trigonometry/sin.rs:886-893), tightened to a few f32 ulps where one op is involved.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ULP32 = float(np.finfo(np.float32).eps)  # 1.19e-7


def f32r(x):
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def same_bits(a, b):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64)).reshape(-1)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def same_values(a, b):
    """Equal including NaN positions and signed zeros (NaN payloads may differ after an f32 round trip)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    return np.array_equal(a[m], b[m]) and np.array_equal(np.signbit(a[m]), np.signbit(b[m]))


def close32(got, want, ulps=2.0, atol=0.0):
    got, want = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(want, dtype=np.float64).reshape(-1)
    if got.shape != want.shape or not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    m = np.isfinite(want)
    if not np.array_equal(got[~m & ~np.isnan(want)], want[~m & ~np.isnan(want)]):
        return False
    return bool(np.all(np.abs(got[m] - want[m]) <= ulps * ULP32 * np.abs(want[m]) + atol))


@pytest.fixture(scope="module")
def prov32(built):
    import os
    from runmat_amd import HipProvider

    p = HipProvider(int(os.environ.get("RMHIP_TEST_DEVICE", "0")), precision="F32")
    yield p
    p.close()


def test_f32_boundary_is_f64_and_rounds_once(prov32, prov):
    from runmat_amd import ProviderError

    assert prov32.precision() == "F32" and prov32.scalar_ty() == "f32" and prov.precision() == "F64"
    assert prov32.device_info_struct()["precision_bits"] == 32 and prov.device_info_struct()["precision_bits"] == 64
    x = np.array([0.1, -0.1, 1.0 / 3.0, 1e-45, 1e-50, 3.4028234663852886e38, 3.5e38, -1e300, np.inf, -np.inf, np.nan, 0.0,
                  -0.0, 16777217.0, 1.0000000596046448, 2.0 ** -126, 2.0 ** -149])
    for n in (1, 2, 3, 4, 5, 17):
        h = prov32.upload(x[:n])
        assert prov32.buffer_bits(h) == 32 and h.shape == (n, 1)
        assert same_values(prov32.download(h), f32r(x[:n]))
        prov32.free(h)
    rng = np.random.default_rng(5)
    for shape in ((1, 1), (3, 5), (129, 7), (1, 4097), (64, 64, 3)):
        A = rng.standard_normal(shape) * 10.0 ** rng.integers(-20, 20, size=shape)
        h = prov32.upload(A)
        assert h.shape == shape and same_bits(prov32.download_matrix(h), f32r(A))
        r = prov32.reshape(h, (int(np.prod(shape)), 1))  # same buffer id, new shape
        assert r.buffer_id == h.buffer_id and prov32.buffer_bits(r) == 32 and same_bits(prov32.download(r), f32r(A).reshape(-1, order="F"))
        prov32.free(r)
    f = prov32.fill((5, 3), 0.1)
    assert prov32.buffer_bits(f) == 32 and same_bits(prov32.download(f), np.full(15, f32r(0.1)))
    B = rng.standard_normal((37, 19))
    t = prov32.transpose(prov32.upload(B))  # a view of f32 storage
    assert t.shape == (19, 37) and prov32.buffer_bits(t) == 32 and same_bits(prov32.download_matrix(t), f32r(B).T)
    tt = prov32.reshape(t, (19 * 37, 1))  # reshape of a view materialises, still f32
    assert prov32.buffer_bits(tt) == 32 and same_bits(prov32.download(tt), f32r(B).T.reshape(-1, order="F"))
    # precision is a property of the provider: it cannot change once buffers exist
    with pytest.raises(ProviderError) as e:
        prov32._check(prov32._lib.rmhip_set_precision(prov32._ctx, 64))
    assert e.value.code == 1 and "property of the provider" in str(e.value)
    with pytest.raises(ProviderError):
        prov32._check(prov32._lib.rmhip_set_precision(prov32._ctx, 16))


UNARY = ["sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "exp", "expm1", "log", "log2", "log10",
         "log1p", "sqrt", "abs", "sign", "floor", "ceil", "round", "fix", "neg", "exp2", "heaviside", "isnan", "isinf",
         "isfinite", "single", "erf", "sinc", "not"]
BINARY = ["add", "sub", "mul", "div", "pow", "max", "min", "hypot", "atan2", "mod", "rem", "eq", "ne", "lt", "le", "gt", "ge",
          "and", "or", "xor"]
SCALAR = ["add", "sub", "mul", "div", "rsub", "rdiv", "max", "min"]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 1023, 4099, 70001])
def test_f32_per_op_kernels_match_f64_kernels_rounded(prov32, prov, oracle, n):
    rng = np.random.default_rng(100 + n)
    x = f32r(rng.uniform(-3.0, 3.0, (n, 1)))
    y = f32r(rng.uniform(-3.0, 3.0, (n, 1)))
    x[::7] = np.round(x[::7])
    if n > 8:
        x[3], y[5], x[6] = np.nan, np.inf, 0.0
    h32, g32, h64, g64 = prov32.upload(x), prov32.upload(y), prov.upload(x), prov.upload(y)
    names = UNARY if n in (5, 1023, 70001) else ["sin", "abs", "single"]
    for name in names:
        got = prov32.download(getattr(prov32, "unary_" + name)(h32) if name != "not" else prov32.logical_not(h32))
        ref = prov.download(getattr(prov, "unary_" + name)(h64) if name != "not" else prov.logical_not(h64))
        assert same_values(got, f32r(ref)), name
        with np.errstate(all="ignore"):
            assert close32(got, f32r(oracle.unary(name, x)), ulps=1.0), name  # f64 libm ulps can flip one f32 rounding
    for name in (BINARY if n in (5, 1023, 70001) else ["add", "div", "max"]):
        got = prov32.download(prov32._binary(name, h32, g32))
        ref = prov.download(prov._binary(name, h64, g64))
        assert same_values(got, f32r(ref)), name
        with np.errstate(all="ignore"):
            assert close32(got, f32r(oracle.binary(name, x, y)), ulps=1.0), name
    for name in SCALAR:
        got = prov32.download(getattr(prov32, "scalar_" + name)(h32, 0.3))
        ref = prov.download(getattr(prov, "scalar_" + name)(h64, 0.3))
        assert same_values(got, f32r(ref)), name  # the scalar itself stays f64, like a double scalar on the CPU path
    for h in (h32, g32):
        assert prov32.buffer_bits(h) == 32
        prov32.free(h)
    prov.free(h64)
    prov.free(g64)


def test_f32_binary_broadcast(prov32, prov, oracle):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(7)
    cases = [((300, 1), (1, 70)), ((300, 70), (1, 70)), ((300, 70), (300, 1)), ((1, 1), (33, 5)), ((5, 1, 7), (1, 6, 1)),
             ((4, 3, 2), (4, 1, 2)), ((1025, 3), (1025, 3)), ((32, 700), (1, 700)), ((32, 1), (32, 700)), ((3, 1, 70), (1, 90, 1))]
    for sa, sb in cases:
        A, B = f32r(rng.standard_normal(sa)), f32r(rng.standard_normal(sb))
        for name in ("add", "mul", "atan2", "lt"):
            o32 = getattr(prov32, "elem_" + name)(prov32.upload(A), prov32.upload(B))
            o64 = getattr(prov, "elem_" + name)(prov.upload(A), prov.upload(B))
            assert o32.shape == o64.shape and prov32.buffer_bits(o32) == 32
            assert same_values(prov32.download(o32), f32r(prov.download(o64))), (sa, sb, name)
            assert close32(prov32.download_matrix(o32), f32r(oracle.binary(name, A, B)), ulps=1.0), (sa, sb, name)
    with pytest.raises(ProviderError) as e:
        prov32.elem_add(prov32.upload(np.ones((3, 2))), prov32.upload(np.ones((2, 3))))
    assert e.value.code == 3


def _chain_cpu_single(oracle, x):
    """benchmarks/elementwise-math on a `single` array, CPU semantics: every builtin rounds its result to f32."""
    s = lambda v: f32r(v)  # noqa: E731
    y0 = s(s(oracle.unary("sin", x)) * s(oracle.unary("exp", s(-x / 10.0))))
    y1 = s(s(y0 * s(oracle.unary("cos", s(x / 4.0)))) + s(0.25 * s(y0 * y0)))
    return s(s(oracle.unary("tanh", y1)) + s(0.1 * y1))


@pytest.mark.parametrize("shape", [(1, 1), (3, 1), (7, 1), (5, 3), (127, 9), (1024, 33), (4093, 1), (512, 512)])
def test_f32_fused_elementwise_fast_path(prov32, prov, oracle, shape):
    from planner_requests import elementwise_math_plan, sin_mul_add_plan

    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    A, B, C = (f32r(rng.uniform(-np.pi, np.pi, shape)) for _ in range(3))
    plan, out = sin_mul_add_plan()
    n = A.size
    h32 = prov32.fused_elementwise(plan.generate_wgsl_for_output(out, "f32"), [prov32.upload(M) for M in (A, B, C)], shape, n)
    h64 = prov.fused_elementwise(plan.generate_wgsl_for_output(out, "f64"), [prov.upload(M) for M in (A, B, C)], shape, n)
    assert prov32.buffer_bits(h32) == 32 and h32.shape == shape
    assert same_bits(prov32.download(h32), f32r(prov.download(h64)))
    # CPU semantics: single(single(sin(A)) .* B) + C, each step rounded
    want = f32r(f32r(f32r(oracle.unary("sin", A)) * B) + C)
    assert close32(prov32.download_matrix(h32), want, ulps=2.0, atol=2 * ULP32)
    # the 14-op benchmark chain; its constants arrive as 1-element inputs (stored as f32 like every tensor of this provider)
    from planner_exec import execute_elementwise

    plan2, out2 = elementwise_math_plan()
    x = f32r(np.linspace(0.0, 4.0 * np.pi, n).reshape(shape, order="F"))
    consts = [float(f32r(v)) for v in (10.0, 4.0, 0.25, 2.0, 0.1)]
    (g32,) = execute_elementwise(prov32, plan2, [out2], [prov32.upload(x)] + consts)
    (g64,) = execute_elementwise(prov, plan2, [out2], [prov.upload(x)] + consts)
    assert prov32.buffer_bits(g32) == 32 and same_bits(prov32.download(g32), f32r(prov.download(g64)))
    assert np.max(np.abs(prov32.download_matrix(g32) - _chain_cpu_single(oracle, x))) < 1e-5  # reference F32 tolerance



def test_f32_generated_sin_cos_whole_range(prov32, oracle):
    """Generated kernels of a precision-32 provider use rm_sincos_r32 (skel_common.h): one-step Cody-Waite + the plain
    minimax polynomials below 2^20, the library path above.  Against the oracle's f64 sin / cos rounded once: never more
    than 1 ulp of f32 away, and (because the f64 error is ~1 ulp of f64) the same f32 value everywhere on this sample."""
    from planner_requests import FusionGroupPlan

    rng = np.random.default_rng(77)
    k = np.arange(-4000, 4001, dtype=np.float64)
    parts = [
        np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, 1e-30, -1e-30, 1e-8, 0.5, -0.5, 1.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi,
                  1048575.9375, -1048575.9375, 1048576.0, -1048576.0, 1048576.125, 3e6, -7e9, 1e22, 3.4028234663852886e38,
                  np.inf, -np.inf, np.nan]),
        k * (np.pi / 2), np.nextafter(f32r(k * (np.pi / 2)).astype(np.float32), np.float32(np.inf)).astype(np.float64),
        rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-1e3, 1e3, 200000), rng.uniform(-1048576.0, 1048576.0, 400000),
        np.ldexp(rng.uniform(0.5, 1.0, 100000), rng.integers(-126, 127, 100000)) * rng.choice([-1.0, 1.0], 100000),
    ]
    x = f32r(np.concatenate(parts)).reshape(-1, 1)
    hx = prov32.upload(x)
    for fn in ("sin", "cos"):
        p = FusionGroupPlan()
        a = p.input()
        out = p.builtin(fn, a)
        h = prov32.fused_elementwise(p.generate_wgsl_for_output(out, "f32"), [hx], x.shape, x.size)
        got = prov32.download_matrix(h)
        with np.errstate(invalid="ignore"):
            want = f32r(oracle.unary(fn, x))
        assert np.array_equal(np.isnan(got), np.isnan(want)), fn
        m = ~np.isnan(want)
        assert np.all(np.abs(got[m] - want[m]) <= ULP32 * np.abs(want[m])), fn
        assert np.array_equal(np.signbit(got[m]), np.signbit(want[m])) or fn == "cos", fn  # sin(-0) = -0
        same = np.count_nonzero(got[m] == want[m])
        assert same == np.count_nonzero(m), (fn, np.count_nonzero(m) - same)
        prov32.free(h)

def test_f32_fused_scalars_broadcast_and_multi_output(prov32, prov, oracle):
    from planner_requests import FusionGroupPlan
    from planner_exec import execute_elementwise, execute_reduction

    rng = np.random.default_rng(9)
    p = FusionGroupPlan()
    a, b, c, quarter = p.input(), p.input(), p.input(), p.input()  # elementwise constants are inputs (fusion.rs:1026-1027)
    t = p.primitive("ElemMul", p.builtin("exp", a), b)
    o1 = p.primitive("Add", t, c)
    o2 = p.primitive("Sub", t, quarter)
    for sa, sb, sc in (((257, 1), (1, 65), 0.5), ((300, 70), (300, 70), (1, 70)), ((129, 3), 2.0, (129, 3)), ((6, 5, 4), (6, 1, 4), -1.5)):
        vals = []
        for s in (sa, sb, sc):
            vals.append(float(f32r(s)) if not isinstance(s, tuple) else f32r(rng.uniform(-1, 1, s)))
        vals.append(0.25)
        out32 = execute_elementwise(prov32, p, [o1, o2], [v if not isinstance(v, np.ndarray) else prov32.upload(v) for v in vals])
        out64 = execute_elementwise(prov, p, [o1, o2], [v if not isinstance(v, np.ndarray) else prov.upload(v) for v in vals])
        for h32, h64 in zip(out32, out64):
            assert h32.shape == h64.shape and prov32.buffer_bits(h32) == 32
            assert same_bits(prov32.download(h32), f32r(prov.download(h64))), (sa, sb, sc)
    # fused reductions: f32 operands read in place, f64 accumulation, one rounding of the result
    q = FusionGroupPlan()
    x, w = q.input(), q.input()
    v = q.primitive("Add", q.primitive("ElemMul", q.builtin("sin", x), w), q.constant(2.0))
    for rows, cols in ((100, 7), (4096, 5), (2, 3000), (8193, 33), (1, 1)):
        X, W = f32r(rng.standard_normal((rows, cols))), f32r(rng.standard_normal((rows, cols)))
        for axis, (rl, ns) in ((0, (rows, cols)), (1, (cols, rows))):
            r32 = execute_reduction(prov32, q, v, [prov32.upload(X), prov32.upload(W)], rl, ns, axis=axis)
            r64 = execute_reduction(prov, q, v, [prov.upload(X), prov.upload(W)], rl, ns, axis=axis)
            assert prov32.buffer_bits(r32) == 32 and r32.shape == (ns,)
            assert close32(prov32.download(r32), f32r(prov.download(r64)), ulps=1.0), (rows, cols, axis)
            want = (np.sin(X) * W + 2.0).sum(axis=axis)
            assert np.allclose(prov32.download(r32), want, rtol=4 * ULP32, atol=1e-5)
    r32 = execute_reduction(prov32, q, v, [prov32.upload(f32r(np.ones((50, 4)))), 3.0], 50, 4, axis=0)  # scalar operand
    assert np.allclose(prov32.download(r32), f32r(50 * (np.sin(1.0) * 3.0 + 2.0)), rtol=2 * ULP32)


def test_f32_shader_precision_must_match_provider(prov32, prov):
    from runmat_amd import ProviderError
    from planner_requests import sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    A = np.ones((4, 4))
    with pytest.raises(ProviderError) as e:
        prov32.fused_elementwise(plan.generate_wgsl_for_output(out, "f64"), [prov32.upload(A)] * 3, (4, 4), 16)
    assert e.value.code == 6 and "F32 provider" in str(e.value)
    with pytest.raises(ProviderError) as e:
        prov.fused_elementwise(plan.generate_wgsl_for_output(out, "f32"), [prov.upload(A)] * 3, (4, 4), 16)
    assert e.value.code == 6 and "F64 provider" in str(e.value)


@pytest.mark.parametrize("shape", [(1, 1), (7, 1), (1, 9), (300, 40), (4096, 3), (3, 5000), (2048, 2), (6, 5, 4), (100001, 1)])
def test_f32_reductions_and_dot(prov32, prov, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    X, Y = f32r(rng.standard_normal(shape)), f32r(rng.standard_normal(shape))
    if X.size > 10:
        X.reshape(-1)[::5] = np.round(X.reshape(-1)[::5] * 4)
    h32, h64 = prov32.upload(X), prov.upload(X)
    for op in ("sum", "mean", "min", "max", "prod"):
        # min / max are exact; sums may group differently (the block size follows the slice's BYTES, so an f32 slice of
        # 8192..16383 elements runs 256-thread blocks where the f64 one runs 1024): equal up to one f32 rounding flip
        same = same_values if op in ("min", "max") else (lambda g, r: close32(g, r, ulps=1.0, atol=1e-30))
        got = getattr(prov32, "reduce_" + op)(h32)
        ref = getattr(prov, "reduce_" + op)(h64)
        assert got.shape == (1, 1) and prov32.buffer_bits(got) == 32
        assert same(prov32.download(got), f32r(prov.download(ref))), op
        for d in range(len(shape)):
            got = getattr(prov32, f"reduce_{op}_dim")(h32, d)
            ref = getattr(prov, f"reduce_{op}_dim")(h64, d)
            if op in ("min", "max"):  # ReduceDimResult: the values round-trip through f32 exactly, the 1-based positions are identical
                assert np.array_equal(prov32.download(got.indices), prov.download(ref.indices)), (op, d)
                got, ref = got.values, ref.values
            assert got.shape == ref.shape and same(prov32.download(got), f32r(prov.download(ref))), (op, d)
    # CPU semantics for `sum`/`mean` of a single array: f64 accumulation of the f32 values, result rounded
    assert np.allclose(prov32.download(prov32.reduce_sum(h32))[0], f32r(oracle.reduce_sum(X.reshape(X.shape[0], -1), "all")[0, 0]),
                       rtol=2 * ULP32, atol=1e-6)
    g32, g64 = prov32.upload(Y), prov.upload(Y)
    dims = [None] + list(range(min(2, len(shape))))
    for d in dims:
        got, ref = prov32.dot(h32, g32, d), prov.dot(h64, g64, d)
        assert got.shape == ref.shape and close32(prov32.download(got), f32r(prov.download(ref)), ulps=1.0, atol=1e-12), d
    Xn = X.copy()
    Xn.reshape(-1)[0] = np.nan
    hn = prov32.upload(Xn)
    assert np.isnan(prov32.download(prov32.reduce_sum(hn))[0])
    got = prov32.download(prov32._reduce("sum", hn, -1, omitnan=True))[0]
    assert np.isclose(got, np.nansum(Xn), rtol=4 * ULP32, atol=1e-5)
    if len(shape) == 3:
        got = prov32.reduce_mean_nd(h32, [0, 2])  # mean of means, each step rounded like the CPU's builtin-by-builtin path
        want = f32r(f32r(X.mean(axis=0, keepdims=True)).mean(axis=2, keepdims=True))
        assert got.shape == (1, shape[1], 1) and close32(prov32.download_matrix(got), want, ulps=2.0)


@pytest.fixture
def exact_f32_matmul(monkeypatch):
    """RMHIP_F32_MATMUL=f64: widen -> dgemm -> round once (the CPU's `single` result) instead of the f32 matrix cores."""
    monkeypatch.setenv("RMHIP_F32_MATMUL", "f64")


def test_f32_matmul_solve_and_friends_use_f64_kernels_on_widened_operands(prov32, prov, oracle, exact_f32_matmul):
    rng = np.random.default_rng(33)
    A, B = f32r(rng.standard_normal((150, 70))), f32r(rng.standard_normal((70, 90)))
    c32, c64 = prov32.matmul(prov32.upload(A), prov32.upload(B)), prov.matmul(prov.upload(A), prov.upload(B))
    assert prov32.buffer_bits(c32) == 32 and same_bits(prov32.download(c32), f32r(prov.download(c64)))
    assert np.max(np.abs(prov32.download_matrix(c32) - f32r(oracle.matmul(A, B)))) <= 4 * ULP32 * np.max(np.abs(A) @ np.abs(B))
    # transpose views of f32 storage feed the transposed-operand dgemm after widening
    At, Bt = f32r(rng.standard_normal((70, 150))), f32r(rng.standard_normal((90, 70)))
    v32 = prov32.matmul(prov32.transpose(prov32.upload(At)), prov32.transpose(prov32.upload(Bt)))
    v64 = prov.matmul(prov.transpose(prov.upload(At)), prov.transpose(prov.upload(Bt)))
    assert same_bits(prov32.download(v32), f32r(prov.download(v64)))
    assert np.allclose(prov32.download_matrix(v32), At.T @ Bt.T, rtol=1e-5, atol=1e-4)
    s32, s64 = prov32.syrk(prov32.upload(A)), prov.syrk(prov.upload(A))
    assert same_bits(prov32.download(s32), f32r(prov.download(s64)))
    # x = A\b: solved in f64 from the f32 operands, rounded once
    n = 300
    M = f32r(rng.uniform(-1, 1, (n, n)) + n * np.eye(n))
    rhs = f32r(M @ np.ones((n, 2)))
    x32, x64 = prov32.mldivide(prov32.upload(M), prov32.upload(rhs)), prov.mldivide(prov.upload(M), prov.upload(rhs))
    assert prov32.buffer_bits(x32) == 32 and same_bits(prov32.download(x32), f32r(prov.download(x64)))
    assert np.max(np.abs(prov32.download(x32) - 1.0)) < 1e-5
    lu32, lu64 = prov32.lu(prov32.upload(M)), prov.lu(prov.upload(M))
    for k in ("combined", "lower", "upper", "perm_matrix", "perm_vector"):
        assert same_bits(prov32.download(getattr(lu32, k)), f32r(prov.download(getattr(lu64, k)))), k
    # the remaining hooks (composites round after every inner step, like the CPU's builtin-by-builtin path)
    Xc = f32r(rng.standard_normal((200, 12)))
    cov32 = prov32.covariance(prov32.upload(Xc))
    assert prov32.buffer_bits(cov32) == 32 and np.allclose(prov32.download_matrix(cov32), np.cov(Xc, rowvar=False), rtol=1e-5, atol=1e-5)
    Xt = f32r(rng.standard_normal((30000, 11)) + np.arange(11))  # many samples of a few variables: f32 storage read in place by the VALU Gram kernel
    covt = prov32.covariance(prov32.upload(Xt))
    assert prov32.buffer_bits(covt) == 32 and close32(prov32.download_matrix(covt), f32r(np.cov(Xt, rowvar=False)), ulps=2.0, atol=1e-7)
    xn = Xc.copy()
    xn[7, 2] = np.inf  # the poisoned column's pairs are NaN, the others finite (cov.rs:916-953), also through f32 storage
    covn = prov32.download_matrix(prov32.covariance(prov32.upload(xn)))
    wantn = oracle.covariance(xn)
    assert np.array_equal(np.isnan(covn), np.isnan(wantn)) and np.allclose(covn[~np.isnan(wantn)], wantn[~np.isnan(wantn)], rtol=1e-5, atol=1e-5)
    img = f32r(rng.uniform(0, 1, (3, 16, 20)))
    i32 = prov32.image_normalize(prov32.upload(img), 3, 16, 20, 1e-6, gain=1.5, bias=0.1, gamma=1.8, clamp_zero=True)
    i64 = prov.image_normalize(prov.upload(img), 3, 16, 20, 1e-6, gain=1.5, bias=0.1, gamma=1.8, clamp_zero=True)
    assert close32(prov32.download(i32), f32r(prov.download(i64)), ulps=1.0, atol=1e-7)  # (four elements per access against two: other merge order)
    for bshape in ((4, 33, 20), (16, 64, 48), (2, 40, 30), (8, 7, 5), (1, 30, 22), (3, 5, 7), (300, 6, 5), (1000, 4, 4)):  # vector widths differ between f32 (4) and f64 (2); > 256 planes: widened
        img = f32r(rng.uniform(0, 1, bshape))
        j32 = prov32.image_normalize(prov32.upload(img), bshape[0], bshape[1], bshape[2], 1e-6, gain=1.5, bias=0.1)
        j64 = prov.image_normalize(prov.upload(img), bshape[0], bshape[1], bshape[2], 1e-6, gain=1.5, bias=0.1)
        assert prov32.buffer_bits(j32) == 32 and close32(prov32.download(j32), f32r(prov.download(j64)), ulps=1.0, atol=1e-7), bshape
        assert np.allclose(prov32.download_matrix(j32), oracle.image_normalize(img, 1e-6, gain=1.5, bias=0.1), rtol=1e-5, atol=1e-5), bshape
    p32 = prov32.matmul_power_step(prov32.upload(A), prov32.upload(B), 1e-12)
    assert np.allclose(prov32.download_matrix(p32), oracle.matmul_power_step(A, B, 1e-12), rtol=1e-5, atol=1e-6)
    # matmul_epilogue: the in-place diag_output buffer must receive its values although its storage is f32
    Ae, Be = f32r(rng.standard_normal((40, 30))), f32r(rng.standard_normal((30, 40)))
    rsc = f32r(rng.uniform(0.5, 2.0, (40, 1)))
    dg32, dg64 = prov32.zeros((40, 1)), prov.zeros((40, 1))
    e32 = prov32.matmul_epilogue(prov32.upload(Ae), prov32.upload(Be), alpha=0.5, beta=0.25, row_scale=prov32.upload(rsc), clamp_min=-1.0,
                                 diag_output=dg32)
    e64 = prov.matmul_epilogue(prov.upload(Ae), prov.upload(Be), alpha=0.5, beta=0.25, row_scale=prov.upload(rsc), clamp_min=-1.0,
                               diag_output=dg64)
    assert same_bits(prov32.download(e32), f32r(prov.download(e64)))
    assert prov32.buffer_bits(dg32) == 32 and same_bits(prov32.download(dg32), f32r(prov.download(dg64)))
    assert np.any(prov32.download(dg32) != 0.0)
    d32 = prov32.diag_extract(prov32.upload(f32r(rng.standard_normal((9, 7)))), 1)
    assert prov32.buffer_bits(d32) == 32 and d32.shape[0] == 6


@pytest.mark.parametrize("m,k,n", [(128, 16, 128), (256, 128, 384), (150, 70, 90), (1, 33, 1), (129, 1, 127), (5, 1000, 7),
                                   (640, 515, 130), (768, 1024, 512), (3, 2, 4),
                                   (128, 16384, 128), (200, 9000, 70), (64, 20000, 64),  # these three split along k
                                   (64, 8192, 48), (32, 16384, 20)])  # A fills whole 2 MiB pages, partial tile of rows: no pair read behind the buffer
def test_f32_matmul_on_the_f32_matrix_cores(prov32, oracle, m, k, n, monkeypatch):
    """Default precision-32 matmul: v_mfma_f32_16x16x4_f32 with f32 accumulation (what the reference's F32 backend
    does; its checks allow 1e-4 relative, wgpu_profile.rs:20-21).  Bound: k * eps32 * sum|a||b| per element, the
    forward error of any f32 dot product; the f64 path (RMHIP_F32_MATMUL=f64) is the comparison."""
    monkeypatch.delenv("RMHIP_F32_MATMUL", raising=False)
    rng = np.random.default_rng(m * 7 + k * 3 + n)
    A, B = f32r(rng.standard_normal((m, k))), f32r(rng.standard_normal((k, n)))
    want = oracle.matmul(A, B)
    bound = (k + 2) * ULP32 * (np.abs(A) @ np.abs(B)) + 1e-30
    ha, hb = prov32.upload(A), prov32.upload(B)
    c = prov32.matmul(ha, hb)
    assert c.shape == (m, n) and prov32.buffer_bits(c) == 32
    got = prov32.download_matrix(c)
    assert np.all(np.abs(got - want) <= bound), float(np.max(np.abs(got - want) / bound))
    # transposed operands in place: A' * B and A * B' (views of f32 storage), both transposed (B is materialised)
    hat, hbt = prov32.transpose(prov32.upload(np.ascontiguousarray(A.T))), prov32.transpose(prov32.upload(np.ascontiguousarray(B.T)))
    for x, y in ((hat, hb), (ha, hbt), (hat, hbt)):
        g2 = prov32.download_matrix(prov32.matmul(x, y))
        assert np.all(np.abs(g2 - want) <= bound)
    if k <= 2048:  # A' * A (k x k) on the same kernel, transposed-A variant; the CPU check is O(m k^2)
        g3 = prov32.syrk(ha)
        assert g3.shape == (k, k) and prov32.buffer_bits(g3) == 32
        assert np.all(np.abs(prov32.download_matrix(g3) - oracle.matmul(A.T, A)) <= (m + 2) * ULP32 * (np.abs(A).T @ np.abs(A)) + 1e-30)
    else:  # tall operand: B' * B is n x n with the long k -- the split-K shape of a Gram matrix
        g3 = prov32.syrk(hb)
        assert g3.shape == (n, n) and np.all(np.abs(prov32.download_matrix(g3) - oracle.matmul(B.T, B)) <= (k + 2) * ULP32 * (np.abs(B).T @ np.abs(B)) + 1e-30)
    monkeypatch.setenv("RMHIP_F32_MATMUL", "f64")
    exact = prov32.download_matrix(prov32.matmul(ha, hb))
    assert same_bits(exact, f32r(want)) or np.max(np.abs(exact - f32r(want))) <= ULP32 * np.max(np.abs(want))


def test_f32_matmul_random_shapes_fuzz(prov32, monkeypatch):
    """80 random shapes through the f32 matrix-core product and its A' * B form (whole and guarded eight-wave tiles, the element-checking
    kernel for A * B', split-K) against the f64 product of the same f32 operands."""
    monkeypatch.delenv("RMHIP_F32_MATMUL", raising=False)
    rng = np.random.default_rng(927)
    edges = np.array([1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513])

    def dim():
        return int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 600))

    for _ in range(80):
        m, k, n = dim(), dim(), dim()
        A, B = f32r(rng.standard_normal((m, k))), f32r(rng.standard_normal((k, n)))
        want = A @ B
        bound = (k + 2) * ULP32 * (np.abs(A) @ np.abs(B)) + 1e-30
        ha, hb = prov32.upload(A), prov32.upload(B)
        got = prov32.download_matrix(prov32.matmul(ha, hb))
        assert got.shape == (m, n) and np.all(np.abs(got - want) <= bound), (m, k, n)
        hat = prov32.upload(np.ascontiguousarray(A.T))
        g2 = prov32.download_matrix(prov32.matmul(prov32.transpose(hat), hb))
        assert np.all(np.abs(g2 - want) <= bound), ("A'", m, k, n)
        for h in (ha, hb, hat):
            prov32.free(h)


def test_f32_matmul_exactness_cases_and_errors(prov32, monkeypatch):
    from runmat_amd import ProviderError

    monkeypatch.delenv("RMHIP_F32_MATMUL", raising=False)
    # small integers are exact in f32 accumulation: the reference's matmul KATs (mtimes.rs:495-503,688-706)
    a = prov32.upload(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]))
    b = prov32.upload(np.array([[7.0, 8.0], [9.0, 10.0], [11.0, 12.0]]))
    assert np.array_equal(prov32.download_matrix(prov32.matmul(a, b)), np.array([[58.0, 64.0], [139.0, 154.0]]))
    eye = prov32.upload(np.eye(200))
    rngm = f32r(np.random.default_rng(4).standard_normal((200, 330)))
    assert same_bits(prov32.download_matrix(prov32.matmul(eye, prov32.upload(rngm))), rngm)  # detects any transposition slip
    with pytest.raises(ProviderError) as e:
        prov32.matmul(a, a)
    assert e.value.code == 3
    z = prov32.matmul(prov32.upload(np.zeros((4, 0))), prov32.upload(np.zeros((0, 5))))  # k == 0: zeros
    assert z.shape == (4, 5) and np.array_equal(prov32.download(z), np.zeros(20))
    x = np.array([[np.inf, 1.0], [np.nan, 2.0]])
    got = prov32.download_matrix(prov32.matmul(prov32.upload(x), prov32.upload(np.ones((2, 2)))))
    assert np.isinf(got[0, 0]) and np.isnan(got[1, 0])


def test_f32_rng_streams_are_the_f64_streams_rounded(prov32, prov, oracle):
    for seed in (0, 7):
        prov32.rng_seed(seed)
        prov.rng_seed(seed)
        u32, u64 = prov32.random_uniform((1001, 3)), prov.random_uniform((1001, 3))
        z32, z64 = prov32.random_normal((777,)), prov.random_normal((777,))
        assert prov32.buffer_bits(z32) == 32
        assert same_bits(prov32.download(u32), f32r(prov.download(u64)))  # randn(...,'single') = the f64 stream rounded
        assert same_bits(prov32.download(z32), f32r(prov.download(z64)))
        assert prov32.get_rng_state() == prov.get_rng_state()
    S = f32r(np.full((5000, 1), 100.0))
    prov32.rng_seed(0)
    prov.rng_seed(0)
    e32 = prov32.stochastic_evolution(prov32.upload(S), 0.0002, 0.0126, 8)
    e64 = prov.stochastic_evolution(prov.upload(S), 0.0002, 0.0126, 8)
    assert same_bits(prov32.download(e32), f32r(prov.download(e64)))


def test_f32_context_with_external_f64_memory_and_block_views(prov32, prov):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(3)
    A, B = rng.standard_normal((33, 9)), f32r(rng.standard_normal((33, 9)))
    owner = prov.upload(A)  # f64 storage owned by the other context on the same device
    prov.synchronize()
    ext = prov32.wrap_external(prov.device_ptr(owner), (33, 9))
    assert prov32.buffer_bits(ext) == 64
    hb = prov32.upload(B)
    s = prov32.elem_add(ext, hb)  # mixed operands: widened f64 path, result stored as f32
    assert prov32.buffer_bits(s) == 32 and same_bits(prov32.download(s), f32r(A + B).reshape(-1, order="F"))
    u = prov32.unary_sin(ext)
    assert prov32.buffer_bits(u) == 32 and same_bits(prov32.download(u), f32r(prov.download(prov.unary_sin(owner))))
    r = prov32.reduce_sum_dim(ext, 0)
    assert same_bits(prov32.download(r), f32r(prov.download(prov.reduce_sum_dim(owner, 0))))
    with pytest.raises(ProviderError) as e:  # in-place block updates need f64 storage
        prov32.blk_copy((hb, 0, 0, 4, 4))
    assert e.value.code == 2 and "f32" in str(e.value)
    prov32.free(ext)
    assert same_bits(prov.download(owner), A.reshape(-1, order="F"))  # untouched and never freed by the wrapper


def test_f32_large_fused_and_traffic_halves(prov32, prov):
    """8192 x 2048 keeps the test quick: the f32 kernel must agree with the f64 kernel everywhere and run measurably
    faster (half the bytes); the strict bandwidth numbers live in bench.py / DESIGN.md."""
    from planner_requests import sin_mul_add_plan

    shape = (8192, 2048)
    n = shape[0] * shape[1]
    plan, out = sin_mul_add_plan()
    sh32, sh64 = plan.generate_wgsl_for_output(out, "f32"), plan.generate_wgsl_for_output(out, "f64")
    in32 = [prov32.fill_uniform(s, lo, hi, shape) for s, lo, hi in ((1, -np.pi, np.pi), (2, -1, 1), (3, -1, 1))]
    # the same f32-rounded values on the F64 side
    in64 = [prov.upload(prov32.download(h), shape) for h in in32]
    h32, h64 = prov32.fused_elementwise(sh32, in32, shape, n), prov.fused_elementwise(sh64, in64, shape, n)
    assert same_bits(prov32.download(h32), f32r(prov.download(h64)))

    def time(p, shader, ins):
        for _ in range(3):
            p.free(p.fused_elementwise(shader, ins, shape, n))
        p.timer_begin()
        for _ in range(20):
            p.free(p.fused_elementwise(shader, ins, shape, n))
        return p.timer_end() / 20

    t32, t64 = time(prov32, sh32, in32), time(prov, sh64, in64)
    print(f"fused sin(A).*B+C {shape}: f32 storage {t32 * 1e3:.1f} us, f64 storage {t64 * 1e3:.1f} us")
    assert t32 < t64


def test_f32_provider_against_the_reference_scripts_float32_outputs(prov32):
    """The reference's own benchmark comparators run in float32 (tests/golden/*.json, generated by tests/golden/
    make_golden.py from /root/reference/benchmarks); a precision-32 provider is the like-for-like configuration."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from workloads import golden_elementwise_math, golden_image_cases, lcg_image_field
    from planner_requests import elementwise_math_plan
    from planner_exec import execute_elementwise

    g = golden_image_cases()
    p = g["params"]
    for case in g["cases"]:  # 4k-image-processing/python_numpy_lcg.py: MSE of the normalised, gamma-corrected frames
        imgs = lcg_image_field(case["B"], case["H"], case["W"], p["seed"])
        h = prov32.upload(imgs.reshape(-1, order="F"), imgs.shape)
        out = prov32.download(prov32.image_normalize(h, case["B"], case["H"], case["W"], p["eps0"], gain=p["gain"], bias=p["bias"],
                                                     gamma=p["gamma"], clamp_zero=True))
        mse = float(np.mean((out - imgs.reshape(-1, order="F")) ** 2))
        assert abs(mse - case["mse"]) <= 2e-5 * case["mse"], (mse, case["mse"])
    plan, out_id = elementwise_math_plan()
    for case in golden_elementwise_math()["cases"]:  # elementwise-math/python_numpy.py: y2 at 17 sample points
        n = case["points"]
        x = np.linspace(0.0, 4.0 * np.pi, n, dtype=np.float32).astype(np.float64).reshape(n, 1)
        (y,) = execute_elementwise(prov32, plan, [out_id], [prov32.upload(x), 10.0, 4.0, 0.25, 2.0, 0.1])
        got = prov32.download(y)[case["indices"]]
        assert np.max(np.abs(got - np.array(case["y2"]))) <= 2e-6, np.max(np.abs(got - np.array(case["y2"])))
