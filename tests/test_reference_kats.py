"""Known-answer tests the reference holds for the hot path, restated on their own generators:

  crates/runmat-accelerate/tests/matmul_epilogue.rs:24-210          (alpha/beta + row/col scale, column divide, clamp + pow)
  crates/runmat-accelerate/tests/fused_square_mean_all_parity.rs:18-170   (mean(x.*x,'all'), x_i = sin(0.001 i), n = 1024)
  crates/runmat-accelerate/tests/fused_reduction_dim_n.rs:21-238    (sum(X.*W, dim) for dim = 1 and 2 on 6 x 5)
  crates/runmat-accelerate/tests/fused_reduction_sum_square.rs:20-121     (sum(x.*x) over rows on 5 x 4)
  crates/runmat-accelerate/tests/reduction_mean_all.rs:21-185       (mean(e.*e,'all') of a 64-vector: code generation + shape)
  crates/runmat-accelerate/tests/reduction_broadcast.rs:20-159      (reduce_sum_dim shapes/values, P x C times 1 x C, per-column dot)
  crates/runmat-accelerate/tests/transpose.rs:28-140                (transpose round trip 37 x 29, A' * B with a transposed view)
  crates/runmat-accelerate/tests/matmul_pca_regression.rs:14-161    (#[ignore]d there and fed from dump files that are not in
                                                                    the tree; its shape - 1024 x 1024 times 1024 x 8, repeated,
                                                                    product never collapsing to zero - is kept on a seeded input)
  crates/runmat-accelerate/src/fusion.rs:3736-3875                  (unit tests of the shader generator: `builds_plan_and_template`,
                                                                    `builtin_expr_supports_extended_set`) against tests/planner_requests.py,
                                                                    the request emitter the tests and bench.py drive the ABI with.

Each case runs twice: on the oracle (CPU, `-m "not gpu"`: pins the restatement to the reference's expected values with
the reference's own tolerance) and through the C ABI on the device (`-m gpu`: HIP output vs the oracle on the same data)."""
import numpy as np
import pytest

EPS = 2.220446049250313e-16


# ---- generators (verbatim formulas of the reference tests) -------------------------------------------------------------
def _epilogue_case_1():
    a = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]).reshape(3, 2, order="F")
    b = np.array([1.0, 3.0, 5.0, 7.0, 2.0, 4.0, 6.0, 8.0]).reshape(2, 4, order="F")
    row, col = np.array([2.0, 0.5, 1.0]), np.array([1.0, 2.0, 0.25, 1.5])
    alpha, beta = 1.25, -0.5
    base = np.zeros((3, 4))
    for j in range(4):
        for i in range(3):
            s = 0.0
            for k in range(2):
                s += a[i, k] * b[k, j]
            base[i, j] = s
    want = (base * alpha + beta) * row[:, None] * col[None, :]
    return a, b, row, col, alpha, beta, want


def _fused_plan_mul(inputs=2):
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    ids = [p.input() for _ in range(inputs)]
    v = p.primitive("ElemMul", ids[0], ids[-1])
    return p, v


# ---- CPU: the oracle against the reference's expectations ------------------------------------------------------------------
def test_oracle_matmul_epilogue_kats(oracle):
    a, b, row, col, alpha, beta, want = _epilogue_case_1()
    got, _ = oracle.matmul_epilogue(a, b, alpha=alpha, beta=beta, row_scale=row, col_scale=col)
    assert np.max(np.abs(got - want)) < 1e-9  # matmul_epilogue.rs:97-109
    a2 = np.array([1.0, 2.0, 3.0, 4.0]).reshape(2, 2, order="F")
    got, _ = oracle.matmul_epilogue(a2, np.eye(2), col_scale=np.array([2.0, 4.0]), col_op="divide")
    assert np.max(np.abs(got.reshape(-1, order="F") - [0.5, 1.0, 0.75, 1.0])) < 1e-9  # :149-154
    b3 = np.array([2.0, 0.0, 0.0, 2.0]).reshape(2, 2, order="F")
    got, _ = oracle.matmul_epilogue(a2, b3, clamp_min=4.0, clamp_max=10.0, pow_exponent=2.0)
    want3 = np.clip(a2 @ b3, 4.0, 10.0) ** 2.0
    assert np.max(np.abs(got - want3)) < 5e-5 and np.array_equal(got, want3)  # :190-209 (exact here: small integers)


def test_oracle_reduction_kats(oracle):
    x = np.sin(np.arange(1024) * 0.001).reshape(-1, 1)
    want = float(np.sum(x.reshape(-1) ** 2) / 1024.0)  # fused_square_mean_all_parity.rs:162
    got = oracle.reduce_sum(oracle.binary("mul", x, x), "all", mean=True).reshape(-1)[0]
    assert abs(got - want) < 1e-6 and abs(got - want) <= 1024 * EPS * want
    rows, cols = 6, 5
    X = np.fromfunction(lambda r, c: r + 1.0, (rows, cols))
    W = np.fromfunction(lambda r, c: c + 2.0, (rows, cols))
    prod = oracle.binary("mul", X, W)
    assert np.array_equal(oracle.reduce_sum(prod, (0,)).reshape(-1), [(c + 2.0) * 21.0 for c in range(cols)])  # dim = 1
    assert np.array_equal(oracle.reduce_sum(prod, (1,)).reshape(-1), [(r + 1.0) * 20.0 for r in range(rows)])  # dim = 2
    x2 = np.fromfunction(lambda r, c: r + c + 1.0, (5, 4))
    want2 = [sum((r + c + 1.0) ** 2 for r in range(5)) for c in range(4)]  # fused_reduction_sum_square.rs:88-101
    assert np.array_equal(oracle.reduce_sum(oracle.binary("mul", x2, x2), (0,)).reshape(-1), want2)
    host = np.fromfunction(lambda r, c: r + 10.0 * c, (4, 3))  # reduction_broadcast.rs:24-72
    d0, d1 = oracle.reduce_sum(host, (0,)), oracle.reduce_sum(host, (1,))
    assert d0.shape == (1, 3) and d1.shape == (4, 1)
    assert np.array_equal(d0.reshape(-1), [6.0, 46.0, 86.0]) and np.array_equal(d1.reshape(-1), [30.0, 33.0, 36.0, 39.0])
    xb = np.fromfunction(lambda r, c: r + 1.0, (4, 3))
    sb = np.array([[0.0, 2.0, 4.0]])
    assert np.array_equal(oracle.binary("mul", xb, sb), xb * sb)  # :75-110


def test_oracle_transpose_kats(oracle):
    rows, cols = 37, 29
    data = np.fromfunction(lambda r, c: ((r * 13 + c * 7) % 101) * 0.03125 + 0.5, (rows, cols))
    t = oracle.transpose(data)
    assert t.shape == (cols, rows) and np.array_equal(t, data.T) and np.array_equal(oracle.transpose(t), data)
    m, k, n = 48, 32, 27
    a = np.fromfunction(lambda r, c: ((r * 5 + c * 3) % 17) * 0.125 - 0.75, (m, k))
    b = np.fromfunction(lambda r, c: ((r + c * 11) % 23) * 0.0625 + 0.25, (m, n))
    got = oracle.matmul(oracle.transpose(a), b)
    assert got.shape == (k, n) and np.max(np.abs(got - a.T @ b)) <= 1e-9  # transpose.rs:113-122


def test_request_emitter_matches_the_generators_unit_tests():
    """fusion.rs unit tests on tests/planner_requests.py: `builds_plan_and_template` (:3736-3747), `builtin_expr_supports_
    extended_set` (:3821-3875), and the module prologue build_wgsl_shader writes (:1536-1610)."""
    from planner_requests import FusionGroupPlan, builtin_expr, sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    for ty in ("f32", "f64"):
        wgsl = plan.generate_wgsl_for_output(out, ty)
        assert "@compute" in wgsl  # :3745
        lines = wgsl.splitlines()
        assert lines[0] == "const MAX_RANK: u32 = 128u;"
        assert lines[1] == "struct PackedValue { value: u32, _pad0: u32, _pad1: u32, _pad2: u32 };"
        assert lines[2] == "alias PackedArray = array<PackedValue, MAX_RANK>;"
        assert f"struct Tensor {{ data: array<{ty}>, }};" in lines
        for i in range(3):
            assert f"    in{i}_shape: PackedArray," in lines and f"    in{i}_stride: PackedArray," in lines
            assert f"@group(0) @binding({i}) var<storage, read> input{i}: Tensor;" in lines
        assert "@group(0) @binding(4) var<uniform> params: Params;" in lines
        assert "@compute @workgroup_size(@WG@)" in lines and "fn main(@builtin(global_invocation_id) gid: vec3<u32>) {" in lines
        assert "    if (idx >= params.len) { return; }" in lines and "    let g = idx + params.offset;" in lines
        assert f"fn hypot(a: {ty}, b: {ty}) -> {ty} {{" in lines and f"fn isNan(x: {ty}) -> bool" in wgsl
        assert f"    let tmp0: {ty} = sin(input0.data[i0]);" in lines and "    output.data[g] = tmp2;" in lines
    v = ["v0", "v1"]
    assert builtin_expr("log1p", v[:1], "f32") is not None
    assert "log" in builtin_expr("log10", v[:1], "f64") and "exp" in builtin_expr("expm1", v[:1], "f32")
    for name, want in (("floor", "floor(v0)"), ("asinh", "asinh(v0)"), ("acosh", "acosh(v0)"), ("atanh", "atanh(v0)"),
                       ("sign", "sign(v0)"), ("fix", "trunc(v0)"), ("pow2", "exp2(v0)")):
        assert builtin_expr(name, v[:1], "f32") == want, name
    assert builtin_expr("atan2", v, "f32") == "atan2(v0, v1)" and builtin_expr("hypot", v, "f32") == "hypot(v0, v1)"
    hv = builtin_expr("heaviside", v[:1], "f32")
    assert "0.5" in hv and "isNan(v0)" in hv
    assert all(s in builtin_expr("mod", v, "f32") for s in ("floor", "isInf"))
    assert all(s in builtin_expr("rem", v, "f32") for s in ("trunc", "isInf"))
    # reduction_mean_all.rs:167-171: mean(all) of a square generates a reduction shader
    red = FusionGroupPlan()
    e = red.input()
    assert "@compute" in red.generate_reduction_wgsl(red.primitive("ElemMul", e, e), "f32", axis=0, is_mean=True)


# ---- GPU: the HIP path against the oracle on the same generators ----------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_matmul_epilogue_kats(prov, oracle):
    a, b, row, col, alpha, beta, want = _epilogue_case_1()
    got = prov.download_matrix(prov.matmul_epilogue(prov.upload(a), prov.upload(b), alpha=alpha, beta=beta,
                                                    row_scale=prov.upload(row.reshape(3, 1)), col_scale=prov.upload(col.reshape(1, 4))))
    ora, _ = oracle.matmul_epilogue(a, b, alpha=alpha, beta=beta, row_scale=row, col_scale=col)
    assert got.shape == (3, 4) and np.max(np.abs(got - want)) < 1e-9 and np.max(np.abs(got - ora)) <= 8 * EPS * np.max(np.abs(ora))
    a2 = np.array([1.0, 2.0, 3.0, 4.0]).reshape(2, 2, order="F")
    got = prov.download(prov.matmul_epilogue(prov.upload(a2), prov.upload(np.eye(2)), col_scale=prov.upload(np.array([[2.0, 4.0]])),
                                             col_op="divide"))
    assert np.array_equal(got, [0.5, 1.0, 0.75, 1.0])
    b3 = np.array([2.0, 0.0, 0.0, 2.0]).reshape(2, 2, order="F")
    got = prov.download_matrix(prov.matmul_epilogue(prov.upload(a2), prov.upload(b3), clamp_min=4.0, clamp_max=10.0, pow_exponent=2.0))
    ora, _ = oracle.matmul_epilogue(a2, b3, clamp_min=4.0, clamp_max=10.0, pow_exponent=2.0)
    assert np.max(np.abs(got - ora)) <= 2 * EPS * 100.0 and np.max(np.abs(got - np.clip(a2 @ b3, 4.0, 10.0) ** 2.0)) < 5e-5


@pytest.mark.gpu
def test_gpu_fused_reduction_kats(prov, oracle):
    from planner_requests import FusionGroupPlan
    from runmat_amd.provider import ReductionFlavor

    # fused_square_mean_all_parity.rs / reduction_mean_all.rs: mean(x.*x, 'all') as ONE fused reduction request
    for n, gen in ((1024, lambda i: np.sin(i * 0.001)), (64, lambda i: (i + 1.0) * 0.01)):
        x = gen(np.arange(n, dtype=np.float64)).reshape(n, 1)
        red = FusionGroupPlan()
        e = red.input()
        sh = red.generate_reduction_wgsl(red.primitive("ElemMul", e, e), "f64", axis=0, is_mean=True)
        h = prov.fused_reduction(sh, [prov.upload(x)], (1,), n, 1, 256, ReductionFlavor.Mean())
        assert h.shape == (1,)  # reduction_mean_all.rs:175-182: scalar-shaped [1]
        got = prov.download(h)[0]
        want = float(np.sum(x.reshape(-1) ** 2) / n)
        ora = oracle.reduce_sum(oracle.binary("mul", x, x), "all", mean=True).reshape(-1)[0]
        assert abs(got - want) < 1e-6 and abs(got - ora) <= n * EPS * abs(ora)
    # fused_reduction_dim_n.rs: sum(X .* W, dim) for dim = 1 (reduce rows -> [cols]) and dim = 2 (reduce cols -> [rows])
    rows, cols = 6, 5
    X = np.fromfunction(lambda r, c: r + 1.0, (rows, cols))
    W = np.fromfunction(lambda r, c: c + 2.0, (rows, cols))
    hx, hw = prov.upload(X), prov.upload(W)
    for axis, reduce_len, slices in ((0, rows, cols), (1, cols, rows)):
        p, v = _fused_plan_mul(2)
        sh = p.generate_reduction_wgsl(v, "f64", axis=axis)
        out = prov.fused_reduction(sh, [hx, hw], (slices,), reduce_len, slices, 256, ReductionFlavor.Sum())
        assert out.shape == (slices,)
        want = (X * W).sum(axis=axis)
        assert np.array_equal(prov.download(out), want)  # small integers: exact in any order
        assert np.array_equal(prov.download(out), oracle.reduce_sum(oracle.binary("mul", X, W), (axis,)).reshape(-1))
    # fused_reduction_sum_square.rs: sum(x .* x) over rows, 5 x 4, x = r + c + 1
    x2 = np.fromfunction(lambda r, c: r + c + 1.0, (5, 4))
    p, v = _fused_plan_mul(1)
    out = prov.fused_reduction(p.generate_reduction_wgsl(v, "f64", axis=0), [prov.upload(x2)], (4,), 5, 4, 256, ReductionFlavor.Sum())
    assert np.array_equal(prov.download(out), (x2 * x2).sum(axis=0))


@pytest.mark.gpu
def test_gpu_reduction_broadcast_kats(prov, oracle):
    host = np.fromfunction(lambda r, c: r + 10.0 * c, (4, 3))  # reduction_broadcast.rs:20-72
    m = prov.upload(host)
    d0, d1 = prov.reduce_sum_dim(m, 0), prov.reduce_sum_dim(m, 1)
    assert d0.shape == (1, 3) and d1.shape == (4, 1)
    assert np.array_equal(prov.download(d0), [6.0, 46.0, 86.0]) and np.array_equal(prov.download(d1), [30.0, 33.0, 36.0, 39.0])
    xb = np.fromfunction(lambda r, c: r + 1.0, (4, 3))  # :75-110 P x C times 1 x C
    sb = np.array([[0.0, 2.0, 4.0]])
    y = prov.elem_mul(prov.upload(xb), prov.upload(sb))
    assert y.shape == (4, 3) and np.array_equal(prov.download_matrix(y), oracle.binary("mul", xb, sb))
    X = np.fromfunction(lambda r, c: r + 1.0, (5, 4))  # :112-159 per-column dot as mul + reduce_sum_dim(.., 0)
    W = np.fromfunction(lambda r, c: c + 1.0, (5, 4))
    got = prov.download(prov.reduce_sum_dim(prov.elem_mul(prov.upload(X), prov.upload(W)), 0))
    assert np.array_equal(got, (X * W).sum(axis=0))
    assert np.array_equal(prov.download(prov.dot(prov.upload(X), prov.upload(W), 0)), (X * W).sum(axis=0))


@pytest.mark.gpu
def test_gpu_transpose_kats(prov, oracle):
    rows, cols = 37, 29
    data = np.fromfunction(lambda r, c: ((r * 13 + c * 7) % 101) * 0.03125 + 0.5, (rows, cols))
    h = prov.upload(data)
    t = prov.transpose(h)
    assert t.shape == (cols, rows) and np.array_equal(prov.download_matrix(t), oracle.transpose(data))
    tt = prov.transpose(t)
    assert tt.shape == (rows, cols) and np.array_equal(prov.download_matrix(tt), data)  # transpose.rs:57-70
    m, k, n = 48, 32, 27
    a = np.fromfunction(lambda r, c: ((r * 5 + c * 3) % 17) * 0.125 - 0.75, (m, k))
    b = np.fromfunction(lambda r, c: ((r + c * 11) % 23) * 0.0625 + 0.25, (m, n))
    got = prov.download_matrix(prov.matmul(prov.transpose(prov.upload(a)), prov.upload(b)))  # the view is consumed in place
    want = oracle.matmul(oracle.transpose(a), b)
    assert got.shape == (k, n) and np.max(np.abs(got - want)) <= 1e-9
    assert np.max(np.abs(got - want)) <= (m + 2) * EPS * np.max(np.abs(a).T @ np.abs(b))


@pytest.mark.gpu
def test_gpu_matmul_tall_skinny_never_collapses(prov, oracle):
    """Shape of matmul_pca_regression.rs (1024 x 1024 times 1024 x 8, fifteen products in a row, the result must not
    collapse to zero and must match the CPU triple loop); that test is #[ignore]d and reads dump files outside the tree,
    so the operands here are seeded uniforms."""
    rng = np.random.default_rng(26)
    lhs = rng.uniform(-1.0, 1.0, (1024, 1024)) / 32.0
    rhs = rng.uniform(-1.0, 1.0, (1024, 8))
    hl, hr = prov.upload(lhs), prov.upload(rhs)
    cur = rhs
    for it in range(15):
        hp = prov.matmul(hl, hr)
        got = prov.download_matrix(hp)
        want = oracle.matmul(lhs, cur) if it in (0, 14) else lhs @ cur
        bound = (1024 + 2) * EPS * (np.abs(lhs) @ np.abs(cur))
        assert np.max(np.abs(got)) > 1e-6, it
        assert np.all(np.abs(got - want) <= bound + 1e-300), it
        scale = np.max(np.abs(got))
        cur = got / scale  # keep the iteration in range, like the normalisation of the power iteration
        prov.free(hr)
        prov.free(hp)
        hr = prov.upload(cur)
