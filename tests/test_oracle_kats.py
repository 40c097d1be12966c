"""Pins the CPU oracle (oracle/oracle.c) against the reference's own known-answer tests and
sequence definitions. Data below are the inputs / expected outputs the reference's
tests hold; citations give the reference file:line of each vector."""
import math

import numpy as np
import pytest


def cm(data, shape):
    """column-major data -> ndarray"""
    return np.asarray(data, dtype=np.float64).reshape(shape, order="F")


# ---- matmul ------------------------------------------------------------------------------------
def test_matmul_2x3_3x2(oracle):
    # crates/runmat-runtime/src/builtins/math/linalg/ops/mtimes.rs:495-503
    a = np.array([[1, 2, 3], [4, 5, 6]], dtype=float)
    b = np.array([[7, 8], [9, 10], [11, 12]], dtype=float)
    assert np.array_equal(oracle.matmul(a, b), np.array([[58, 64], [139, 154]], dtype=float))


def test_matmul_column_major_kat(oracle):
    # mtimes.rs:688-706 (mtimes_gpu_roundtrip): column-major data [1,2,3,4] * [5,7,6,8] -> [26,38,30,44]
    out = oracle.matmul(cm([1, 2, 3, 4], (2, 2)), cm([5, 7, 6, 8], (2, 2)))
    assert list(out.reshape(-1, order="F")) == [26.0, 38.0, 30.0, 44.0]


def test_matmul_dim_mismatch(oracle):
    # mtimes.rs:628-637 / linalg.rs:7-15
    with pytest.raises(ValueError, match="Inner matrix dimensions must agree"):
        oracle.matmul(np.ones((2, 3)), np.ones((2, 3)))


def test_matmul_small_k_generators(oracle):
    # crates/runmat-accelerate/tests/matmul_small_k.rs:52-96: a[r,c]=(r+1)+0.25c, b[r,c]=(r+2c)%7, tol 1e-9
    m, n, k = 64, 32, 4
    a = np.fromfunction(lambda r, c: (r + 1) + 0.25 * c, (m, k))
    b = np.fromfunction(lambda r, c: (r + 2 * c) % 7, (k, n))
    assert np.max(np.abs(oracle.matmul(a, b) - a @ b)) < 1e-9


def test_matmul_profile_generator(oracle):
    # crates/runmat-accelerate/src/bin/wgpu_profile.rs:1219-1229: base + delta*((idx%128)/127)
    def gen(rows, cols, base, delta):
        idx = np.arange(rows * cols)
        return (base + delta * ((idx % 128) / 127.0)).reshape((rows, cols), order="F")

    a, b = gen(128, 96, 0.5, 1.5), gen(96, 64, -1.0, 2.0)
    ref = a @ b
    assert np.max(np.abs(oracle.matmul(a, b) - ref) / np.maximum(1.0, np.abs(ref))) < 1e-12


def test_matmul_is_sequential_k_sum(oracle):
    # linalg.rs:21-29: sum += a*b with k ascending, separate rounding of product and sum.
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((5, 300)), rng.standard_normal((300, 4))
    out = oracle.matmul(a, b)
    for i in range(5):
        for j in range(4):
            s = 0.0
            for kk in range(300):
                s += a[i, kk] * b[kk, j]
            assert out[i, j] == s


# ---- elementwise / broadcast -------------------------------------------------------------------
def test_plus_times_kats(oracle):
    # crates/runmat-runtime-integration-tests/tests/gpu.rs:28-60
    a, b = cm([1, 2, 3, 4], (2, 2)), cm([5, 6, 7, 8], (2, 2))
    assert np.array_equal(oracle.binary("add", a, b).reshape(-1, order="F"), [6, 8, 10, 12])
    assert np.array_equal(oracle.binary("mul", a, b).reshape(-1, order="F"), [5, 12, 21, 32])


def test_broadcast_shapes(oracle):
    # crates/runmat-accelerate/src/graph.rs:252-271; broadcast.rs tests
    assert oracle.binary("add", np.ones((4, 1)), np.ones((1, 3))).shape == (4, 3)
    assert oracle.binary("add", np.ones((2, 3)), np.ones((2, 1))).shape == (2, 3)
    with pytest.raises(ValueError):
        oracle.binary("add", np.ones((2, 3)), np.ones((3, 2)))


def test_broadcast_values_and_front_padding(oracle):
    # broadcast.rs:108-115: shorter shape is FRONT-padded with ones (trailing dims align)
    rng = np.random.default_rng(1)
    a = rng.standard_normal((4, 1))
    b = rng.standard_normal((1, 3))
    assert np.array_equal(oracle.binary("mul", a, b), a * b)
    c = rng.standard_normal((2, 3, 4))
    d = rng.standard_normal((4,))  # rank-1 [4] aligns with the LAST dim after front padding
    assert np.array_equal(oracle.binary("add", c, d), c + d.reshape(1, 1, 4))


def test_max_min_nan_and_signed_zero(oracle):
    # max.rs:2323-2344 (Include-NaN), :1715-1728 (-0 < +0); min.rs:1519-1531
    a = np.array([[np.nan, 1.0, -0.0, 0.0, 3.0]])
    b = np.array([[1.0, np.nan, 0.0, -0.0, 2.0]])
    mx = oracle.binary("max", a, b).reshape(-1)
    mn = oracle.binary("min", a, b).reshape(-1)
    assert math.isnan(mx[0]) and math.isnan(mx[1]) and math.isnan(mn[0]) and math.isnan(mn[1])
    assert mx[2] == 0.0 and not math.copysign(1, mx[2]) < 0 and not math.copysign(1, mx[3]) < 0
    assert math.copysign(1, mn[2]) < 0 and math.copysign(1, mn[3]) < 0
    assert mx[4] == 3.0 and mn[4] == 2.0


def test_unary_semantics(oracle):
    x = np.array([[-2.5, -0.5, 0.0, 0.5, 2.5, np.nan]])
    assert np.array_equal(oracle.unary("round", x)[0, :5], [-3, -1, 0, 1, 3])  # half away from zero (round.rs:305)
    assert np.array_equal(oracle.unary("fix", x)[0, :5], [-2, -0.0, 0, 0, 2])
    s = oracle.unary("sign", x)[0]
    assert list(s[:5]) == [-1, -1, 0, 1, 1] and math.isnan(s[5])  # sign.rs:236-246
    h = oracle.unary("heaviside", x)[0]
    assert list(h[:5]) == [0, 0, 0.5, 1, 1] and math.isnan(h[5])  # fusion.rs:2945-2953
    assert np.array_equal(oracle.unary("sin", x[:, :5]), np.sin(x[:, :5]))  # same libm


def test_mod_rem_select_chain(oracle):
    # fusion.rs:2954-2970; cases from crates/runmat-vm/tests/fusion_gpu.rs:2965-3020
    a = np.array([[5.0, -5.0, 5.0, -5.0, 5.0, -5.0, 0.0]])
    b = np.array([[3.0, 3.0, -3.0, -3.0, np.inf, np.inf, np.inf]])
    assert np.array_equal(oracle.binary("mod", a, b)[0], [2.0, 1.0, -1.0, -2.0, 5.0, np.inf, 0.0])
    assert np.array_equal(oracle.binary("rem", a, b)[0], [2.0, -2.0, 2.0, -2.0, 5.0, -5.0, 0.0])


# ---- reductions --------------------------------------------------------------------------------
def test_fused_reduction_sum_mul_kat(oracle):
    # crates/runmat-accelerate/tests/fused_reduction_sum_mul.rs:40-137: X[r,c]=r+1, W[r,c]=c+1,
    # sum over rows of X.*W per column, tol 1e-6
    rows, cols = 37, 11
    X = np.fromfunction(lambda r, c: r + 1.0, (rows, cols))
    W = np.fromfunction(lambda r, c: c + 1.0, (rows, cols))
    got = oracle.reduce_sum(oracle.binary("mul", X, W), [0]).reshape(-1)
    want = np.array([(c + 1) * rows * (rows + 1) / 2 for c in range(cols)])
    assert np.max(np.abs(got - want)) < 1e-6


def test_nlms_column_reductions(oracle):
    # crates/runmat-runtime/tests/reduction_parity.rs:71-125 (f32 data; here the same values in f64)
    x = cm([0.1, 0.2, 0.3, 0.4, -0.5, -0.4, -0.3, -0.2, 0.9, 0.7, 0.5, 0.3], (4, 3))
    w = cm([0.05, 0.1, 0.15, 0.2, 0.8, 0.6, 0.4, 0.2, -0.3, -0.2, -0.1, 0.0], (4, 3))
    got = oracle.reduce_sum(oracle.binary("mul", x, w), [0])
    assert got.shape == (1, 3)
    assert np.max(np.abs(got.reshape(-1) - np.sum(x * w, axis=0))) < 1e-12


def test_sum_output_shapes_and_order(oracle):
    # simple_provider.rs:6728-6806 shapes; sum.rs:1031-1053 ascending linear order per output
    rng = np.random.default_rng(2)
    X = rng.standard_normal((7, 5))
    assert oracle.reduce_sum(X, "all").shape == (1, 1)
    assert oracle.reduce_sum(X, [0]).shape == (1, 5)
    assert oracle.reduce_sum(X, [1]).shape == (7, 1)
    r = oracle.reduce_sum(X, [1]).reshape(-1)
    for i in range(7):
        s = 0.0
        for c in range(5):
            s += X[i, c]
        assert r[i] == s
    tot = 0.0
    for v in X.reshape(-1, order="F"):
        tot += v
    assert oracle.reduce_sum(X, "all")[0, 0] == tot


def test_sum_nan_policy_and_mean(oracle):
    # sum.rs:1038-1045,1058-1066 (include => NaN, omit skips); mean.rs:1134-1151 divides by the count
    X = np.array([[1.0, np.nan], [2.0, 4.0]])
    inc = oracle.reduce_sum(X, [0]).reshape(-1)
    assert inc[0] == 3.0 and math.isnan(inc[1])
    om = oracle.reduce_sum(X, [0], omitnan=True).reshape(-1)
    assert list(om) == [3.0, 4.0]
    assert oracle.reduce_sum(np.array([[1.0, 2.0, 4.0]]), "all", mean=True)[0, 0] == 7.0 / 3.0
    assert oracle.reduce_sum(X, [0], omitnan=True, mean=True).reshape(-1)[1] == 4.0


def test_sum_rows_of_sin_x_times_x(oracle):
    # crates/runmat-vm/tests/fusion_gpu.rs:1429-1449: X(r,c)=10c+r (1-based), S=sum(sin(X).*X+2, 2)
    rows, cols = 64, 48
    X = np.fromfunction(lambda r, c: (c + 1) * 10.0 + (r + 1), (rows, cols))
    Y = oracle.binary("add", oracle.binary("mul", oracle.unary("sin", X), X), np.array([[2.0]]))
    S = oracle.reduce_sum(Y, [1]).reshape(-1)
    ref = np.sum(np.sin(X) * X + 2.0, axis=1)
    assert np.max(np.abs(S - ref)) < 1e-9


# ---- RNG ---------------------------------------------------------------------------------------
def test_rng_constants_and_seed(oracle):
    # random.rs:7-13, 128-141: default seed, rng(0) == default, splitmix mixing for nonzero seeds
    assert oracle.rng_default_seed() == 0x9E3779B97F4A7C15
    assert oracle.rng_mix_seed(0) == 0x9E3779B97F4A7C15
    z = (42 + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    assert oracle.rng_mix_seed(42) == z ^ (z >> 31)


def test_rng_uniform_sequence_definition(oracle):
    # random.rs:271-278 + expected_uniform_sequence :613-621: s<-s*6364136223846793005+1, (s>>11)*2^-53
    s = 0x9E3779B97F4A7C15
    want = []
    for _ in range(9):
        s = (s * 6364136223846793005 + 1) & (2**64 - 1)
        want.append((s >> 11) * (1.0 / (1 << 53)))
    got, state = oracle.rng_uniform(0x9E3779B97F4A7C15, 9)
    assert list(got) == want and state == s
    assert oracle.rng_advance(0x9E3779B97F4A7C15, 9) == s  # advance_state :238-256


def test_rng_normal_sequence_definition(oracle):
    # random.rs:279-288, 530-543, expected_normal_sequence :631-643: consecutive (z0, z1) pairs
    s = 0x9E3779B97F4A7C15
    want = []

    def nxt():
        nonlocal s
        s = (s * 6364136223846793005 + 1) & (2**64 - 1)
        return (s >> 11) * (1.0 / (1 << 53))

    while len(want) < 7:
        u1 = nxt() or 2.2250738585072014e-308
        u2 = nxt()
        r = math.sqrt(-2.0 * math.log(u1))
        ang = 2.0 * math.pi * u2
        want.append(r * math.cos(ang))
        if len(want) < 7:
            want.append(r * math.sin(ang))
    got, state = oracle.rng_normal(0x9E3779B97F4A7C15, 7)
    assert list(got) == want
    assert state == s  # odd length still consumes the whole last pair


def test_rng_moments(oracle):
    # crates/runmat-runtime/tests/rng.rs:19-63: |mean| < 0.01, |var-1| < 0.02 at n = 50 000
    z, _ = oracle.rng_normal(oracle.rng_default_seed(), 50000)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02


# ---- LU / mldivide -----------------------------------------------------------------------------
def test_lu_matches_scipy_and_pivot_rule(oracle):
    import scipy.linalg as sl

    rng = np.random.default_rng(3)
    A = rng.standard_normal((9, 9))
    comb, L, U, P, piv = oracle.lu(A)
    assert np.allclose(P @ A, L @ U, atol=1e-12)          # host_lu.rs: P*A = L*U
    p_s, l_s, u_s = sl.lu(A)                                 # A = p l u
    assert np.allclose(L, l_s, atol=1e-12) and np.allclose(U, u_s, atol=1e-12)
    assert np.array_equal(P, p_s.T)
    assert np.array_equal(piv.reshape(-1) - 1, np.argmax(P, axis=1))  # 1-based row ids, host_lu.rs:107


def test_lu_first_max_tie_break_and_singular_cut(oracle):
    # host_lu.rs:38-47: strict '>' keeps the FIRST maximal row; :54-59: |pivot| <= 1e-12 zeroes the column
    A = np.array([[1.0, 2.0], [-1.0, 5.0]])
    _, _, _, _, piv = oracle.lu(A)
    assert list(piv.reshape(-1)) == [1.0, 2.0]
    S = np.array([[1e-13, 1.0], [5e-13, 2.0]])
    comb, L, U, P, piv = oracle.lu(S)
    assert list(piv.reshape(-1)) == [2.0, 1.0]  # swap happens before the cut-off check
    assert comb[1, 0] == 0.0                     # sub-column zeroed, no elimination
    assert comb[1, 1] == 1.0


def test_lu_rectangular_shapes(oracle):
    rng = np.random.default_rng(4)
    for shape in [(6, 4), (4, 6)]:
        A = rng.standard_normal(shape)
        comb, L, U, P, piv = oracle.lu(A)
        assert L.shape == (shape[0], shape[0]) and U.shape == shape  # host_lu.rs:72-100
        assert np.allclose(P @ A, L @ U, atol=1e-12)


def test_mldivide_square_residual(oracle):
    # mldivide.rs:662-680: A=[1 2;3 4] (column-major 1,3,2,4), b=[5;6], ||Ax-b|| < 1e-12
    A, b = cm([1, 3, 2, 4], (2, 2)), cm([5, 6], (2, 1))
    for solve in (oracle.mldivide_svd, oracle.mldivide_lu):
        x = solve(A, b)
        assert np.linalg.norm(A @ x - b) < 1e-12


def test_mldivide_least_squares_residual(oracle):
    # mldivide.rs:682-696: 3x2 least squares, compare with the minimum-norm solution, tol 1e-10
    A, b = cm([1, 3, 5, 2, 4, 6], (3, 2)), cm([7, 8, 9], (3, 1))
    x = oracle.mldivide_svd(A, b)
    ref = np.linalg.lstsq(A, b, rcond=None)[0]
    assert np.linalg.norm(x - ref) < 1e-10
    assert np.linalg.norm(A.T @ (A @ x - b)) < 1e-10


def test_mldivide_scalar_and_rank_deficient(oracle):
    # mldivide.rs:321-325 scalar lhs; :396-404 tolerance drops tiny singular values (min-norm solution)
    assert np.array_equal(oracle.mldivide_svd(np.array([[4.0]]), np.array([[2.0, 8.0]])), [[0.5, 2.0]])
    A = np.array([[1.0, 2.0], [2.0, 4.0]])
    b = np.array([[1.0], [2.0]])
    x = oracle.mldivide_svd(A, b)
    assert np.allclose(x, np.linalg.pinv(A) @ b, atol=1e-12)


def test_mldivide_svd_vs_lu_well_conditioned(oracle):
    rng = np.random.default_rng(5)
    n = 40
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    b = A @ np.ones((n, 1))
    assert np.max(np.abs(oracle.mldivide_svd(A, b) - 1.0)) < 1e-12
    assert np.max(np.abs(oracle.mldivide_lu(A, b) - 1.0)) < 1e-12


# ---- workloads ---------------------------------------------------------------------------------
def test_benchmark_chains_equal_composed_builtins(oracle):
    rng = np.random.default_rng(6)
    A, B, C = rng.uniform(-3, 3, (33, 17)), rng.uniform(-1, 1, (33, 17)), rng.uniform(-1, 1, (33, 17))
    d = oracle.sin_mul_add(A, B, C)
    assert np.array_equal(d, oracle.binary("add", oracle.binary("mul", oracle.unary("sin", A), B), C))
    x = np.linspace(0, 4 * np.pi, 1001).reshape(-1, 1)
    y0 = np.sin(x) * np.exp(-x / 10.0)
    y1 = y0 * np.cos(x / 4.0) + 0.25 * np.power(y0, 2.0)
    y2 = np.tanh(y1) + 0.1 * y1
    assert np.max(np.abs(oracle.elementwise_math_chain(x) - y2)) < 1e-15


def test_fill_uniform_is_counter_based(oracle):
    a = oracle.fill_uniform(7, -1.0, 1.0, 1000)
    assert np.all(a >= -1.0) and np.all(a < 1.0) and abs(a.mean()) < 0.1
    assert np.array_equal(a[:10], oracle.fill_uniform(7, -1.0, 1.0, 10))


def test_matmul_epilogue_order(oracle):
    # simple_provider.rs:7800-7836: alpha/beta, row scale, col scale, clamp, pow, diag -- in that order
    a, b = cm([1, 2, 3, 4], (2, 2)), cm([5, 7, 6, 8], (2, 2))  # product [26 30; 38 44]
    c, d = oracle.matmul_epilogue(a, b, alpha=0.5, beta=1.0, row_scale=np.array([2.0, 4.0]), col_scale=np.array([1.0, 10.0]),
                                  col_op="divide", clamp_max=30.0, pow_exponent=2.0, diag=True)
    want = np.array([[(26 * 0.5 + 1) * 2 / 1, (30 * 0.5 + 1) * 2 / 10], [(38 * 0.5 + 1) * 4 / 1, (44 * 0.5 + 1) * 4 / 10]])
    want = np.minimum(want, 30.0) ** 2.0
    assert np.allclose(c, want, rtol=1e-15) and np.allclose(d, np.diag(want), rtol=1e-15)


def test_erf_sinc_single_kats(oracle):
    # erf.rs:279-287 scalar KATs (tol 1e-15); sinc.rs:302-311; single = f64 storage rounded through f32
    e = oracle.unary("erf", np.array([[1.0, -0.5]]))[0]
    assert abs(e[0] - 0.8427007929497149) < 1e-15 and abs(e[1] + 0.5204998778130465) < 1e-15
    s = oracle.unary("sinc", np.array([[0.0, 1.0, -2.0, 0.5]]))[0]
    assert s[0] == 1.0 and s[1] == 0.0 and s[2] == 0.0 and abs(s[3] - 2.0 / np.pi) < 1e-16
    assert oracle.unary("single", np.array([[0.1]]))[0, 0] == float(np.float32(0.1))


def test_linsolve_triangular_kats(oracle):
    # linsolve.rs:1224-1247 (LT hint) and 1249-1285 (LT + TRANSA='T' == plain solve with A')
    a = np.array([3.0, -1.0, 4.0, 0.0, 2.0, 1.0, 0.0, 0.0, 5.0]).reshape(3, 3, order="F")
    x, rcond = oracle.linsolve(a, np.array([9.0, 1.0, 19.0]), lower=True)
    assert np.allclose(x[:, 0], [3.0, 2.0, 1.0], atol=1e-12) and rcond == 2.0 / 5.0
    a2 = np.array([3.0, 1.0, 0.0, 0.0, 4.0, 2.0, 0.0, 0.0, 5.0]).reshape(3, 3, order="F")
    b2 = np.array([5.0, 14.0, 23.0])
    xt, _ = oracle.linsolve(a2, b2, lower=True, transposed=True)
    assert np.allclose(a2.T @ xt[:, 0], b2, atol=1e-12)
    xg, _ = oracle.linsolve(np.array([[2.0, 1.0], [1.0, 2.0]]), np.array([4.0, 5.0]))  # linsolve.rs:1209-1222
    assert np.allclose(xg[:, 0], [1.0, 2.0], atol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        oracle.linsolve(np.array([[1.0, 0.0], [1.0, 0.0]]), np.ones(2), lower=True)


def test_stochastic_evolution_kat(oracle):
    # stochastic_evolution.rs:39-51 (zero scale => pure drift) and accelerate/tests/stochastic_evolution.rs:17-47
    out, st = oracle.stochastic_evolution(oracle.rng_default_seed(), np.array([[1.0], [2.0]]), 0.1, 0.0, 3)
    assert np.max(np.abs(out[:, 0] - np.array([1.0, 2.0]) * np.exp(0.3))) < 1e-12
    assert st == oracle.rng_advance(oracle.rng_default_seed(), 3 * 2)  # one pair per step
    out3, st3 = oracle.stochastic_evolution(oracle.rng_default_seed(), np.array([[1.0], [2.0], [3.0]]), 0.05, 0.0, 4)
    assert np.max(np.abs(out3[:, 0] - np.array([1.0, 2.0, 3.0]) * np.exp(0.2))) < 1e-9
    assert st3 == oracle.rng_advance(oracle.rng_default_seed(), 4 * 4)  # odd length consumes whole pairs


def test_syrk_kat(oracle):
    # accelerate/tests/syrk.rs:55-100: data[r + c*rows] = r + 1 + 3c, 16 x 5, against A' * A
    rows, cols = 16, 5
    a = np.array([[r + 1 + 3 * c for c in range(cols)] for r in range(rows)], dtype=np.float64)
    got = oracle.syrk(a)
    assert got.shape == (cols, cols) and np.array_equal(got, a.T @ a)  # small integers: exact in any order
    assert np.array_equal(got, got.T)


def _image_case():
    # accelerate/tests/image_normalize.rs:74-92: value(b,h,w) = b + 0.1 h + 0.01 w on a 3 x 4 x 5 tensor
    b, h, w = np.meshgrid(np.arange(3), np.arange(4), np.arange(5), indexing="ij")
    return b + 0.1 * h + 0.01 * w


def test_image_normalize_kat(oracle):
    x = _image_case()
    y = oracle.image_normalize(x, 1e-6, gain=1.05, bias=-0.02, gamma=1.8, clamp_zero=True)
    mu = x.mean(axis=(1, 2), keepdims=True)
    sd = np.sqrt(((x - mu) ** 2).mean(axis=(1, 2), keepdims=True) + 1e-6)
    want = np.maximum((x - mu) / sd * 1.05 - 0.02, 0.0) ** 1.8
    assert y.shape == x.shape and np.max(np.abs(y - want)) < 1e-12   # the reference's own tolerance is 5e-4
    flat = oracle.image_normalize(np.ones((2, 3, 3)), 0.0, clamp_zero=False)   # sigma == 0 -> inv_sigma = 0, not inf
    assert np.array_equal(flat, np.zeros((2, 3, 3)))


def test_matmul_power_step_kat(oracle):
    a = np.array([[1.0, 2.0], [3.0, 4.0]])
    b = np.array([[5.0, 6.0], [7.0, 8.0]])
    p = a @ b
    got = oracle.matmul_power_step(a, b, 0.0)
    assert np.max(np.abs(got - p / np.sqrt((p * p).sum(axis=0)))) < 1e-15
    assert np.max(np.abs((got * got).sum(axis=0) - 1.0)) < 1e-15


def test_covariance_kat(oracle):
    x = np.array([[1.0, 2.0, 0.5], [2.0, 1.0, 0.25], [4.0, 3.0, 1.5], [7.0, 5.0, 2.0]])
    assert np.max(np.abs(oracle.covariance(x) - np.cov(x, rowvar=False))) < 1e-14            # unbiased: rows - 1
    assert np.max(np.abs(oracle.covariance(x, biased=True) - np.cov(x, rowvar=False, bias=True))) < 1e-14
    assert np.all(np.isnan(oracle.covariance(x[:1])))                                          # rows - 1 <= 0 (cov.rs:931-934)
    xn = x.copy(); xn[2, 1] = np.nan
    c = oracle.covariance(xn)
    assert np.isnan(c[1, :]).all() and np.isnan(c[:, 1]).all() and np.isfinite(c[0, 2])       # a non-finite column poisons its pairs only


def test_comparison_and_logical_kats(oracle):
    a = np.array([[1.0, 2.0, np.nan, 0.0]])
    b = np.array([[1.0, 3.0, np.nan, -0.0]])
    assert np.array_equal(oracle.binary("eq", a, b), [[1.0, 0.0, 0.0, 1.0]])
    assert np.array_equal(oracle.binary("ne", a, b), [[0.0, 1.0, 1.0, 0.0]])
    assert np.array_equal(oracle.binary("lt", a, b), [[0.0, 1.0, 0.0, 0.0]])
    assert np.array_equal(oracle.binary("ge", a, b), [[1.0, 0.0, 0.0, 1.0]])
    assert np.array_equal(oracle.binary("and", a, b), [[1.0, 1.0, 1.0, 0.0]])   # NaN counts as non-zero
    assert np.array_equal(oracle.binary("xor", a, np.zeros((1, 4))), [[1.0, 1.0, 1.0, 0.0]])
    assert np.array_equal(oracle.unary("not", a), [[0.0, 0.0, 0.0, 1.0]])


# ---- special functions: the reference's own known-answer tests (crates/runmat-runtime/src/builtins/math/elementwise) ----
def test_special_function_kats(oracle):
    import math

    u = lambda op, xs: oracle.unary(op, np.array(xs, dtype=np.float64).reshape(-1, 1)).reshape(-1)
    # gamma.rs tests `gamma_positive_integer`, `gamma_half_integer`, `gamma_negative_non_integer`, `gamma_matrix` (tol 1e-12),
    # `gamma_pole_returns_inf`, `gamma_small_negative_not_infinite`
    g = u("gamma", [5.0, 0.5, -0.5, 1.0, 3.0, 2.0, 4.0])
    want = [24.0, math.sqrt(math.pi), -2.0 * math.sqrt(math.pi), 1.0, 2.0, 1.0, 6.0]
    assert np.max(np.abs(g - want)) <= 1e-12
    poles = u("gamma", [0.0, -3.0])
    assert np.isposinf(poles[0]) and np.isinf(poles[1])
    small = u("gamma", [-1.0e-10])[0]
    assert np.isfinite(small) and small < 0 and abs(small) > 1e9
    assert np.isnan(u("gamma", [np.nan, -np.inf])).all() and np.isposinf(u("gamma", [np.inf])[0])
    # gammaln.rs `gammaln_scalar_values` (1e-14 / 1e-13), `gammaln_avoids_overflow_for_large_values` (1e-10),
    # `gammaln_tiny_positive_values_use_log_asymptote` (1e-12), zero / inf / negative
    gl = u("gammaln", [1.0, 5.0, 0.5, 171.0])
    assert abs(gl[0]) <= 1e-14 and abs(gl[1] - math.log(24.0)) <= 1e-13 and abs(gl[2] - math.log(math.sqrt(math.pi))) <= 1e-14
    assert abs(gl[3] - 706.5730622457875) <= 1e-10
    tiny = 2.2250738585072014e-308 / 2.0
    assert abs(u("gammaln", [tiny])[0] + math.log(tiny)) <= 1e-12
    assert np.isposinf(u("gammaln", [0.0, np.inf])).all() and np.isnan(u("gammaln", [np.nan, -1.0])).all()
    # factorial.rs: 5! = 120, 0! = 1, [0 1 3 5] -> [1 1 6 120], non-integers and negatives NaN, 171 -> Inf
    assert np.array_equal(u("factorial", [5.0, 0.0, 1.0, 3.0, 4.0]), [120.0, 1.0, 1.0, 6.0, 24.0])
    f = u("factorial", [2.5, -1.0, 171.0, np.inf, -np.inf, np.nan, 170.0])
    assert np.isnan(f[0]) and np.isnan(f[1]) and np.isposinf(f[2]) and np.isposinf(f[3]) and np.isnan(f[4]) and np.isnan(f[5])
    assert f[6] == math.prod(float(i) for i in range(1, 171)) or abs(f[6] / 7.257415615307994e306 - 1) < 1e-15
    # nextpow2.rs: 9 -> 4, 0 -> 0, -3 -> 2, [0 1 3 9] -> [0 0 2 4], Inf -> Inf, NaN -> NaN
    assert np.array_equal(u("nextpow2", [9.0, 0.0, -3.0, 1.0, 3.0]), [4.0, 0.0, 2.0, 0.0, 2.0])
    assert np.isposinf(u("nextpow2", [np.inf])[0]) and np.isnan(u("nextpow2", [np.nan])[0])
    # erfcinv.rs `scalar_values_match_reference_points` (values and tolerances verbatim), end points, domain
    cases = [(0.3, 0.7328690779592166, 2e-14), (0.5, 0.4769362762044698, 2e-14), (1.5, -0.4769362762044698, 2e-14),
             (0.999999999999, 8.862073205887489e-13, 2e-16), (1.000000000001, -8.863057115425171e-13, 2e-16),
             (1e-100, 15.065574702592645, 5e-13), (2.2250738585072014e-308, 26.54325845425098, 5e-13)]
    for x, want_v, tol in cases:
        assert abs(u("erfcinv", [x])[0] - want_v) <= tol, x
    e = u("erfcinv", [1.0, 0.0, 2.0, -0.1, 2.1, np.nan, 5e-324, 2.2250738585072014e-308])
    assert e[0] == 0.0 and np.isposinf(e[1]) and np.isneginf(e[2]) and np.isnan(e[3:6]).all()
    assert np.isfinite(e[6]) and e[6] > e[7] and e[6] < 32.0  # `tiny_tail_inputs_remain_ordered_and_finite`


def test_mrdivide_kats(oracle):
    # mrdivide.rs tests `solves_square_system` ([1 2;3 4] / [5 6;7 8] = [3 -2;2 -1], 1e-12), `divides_matrix_by_scalar`
    a = np.array([1.0, 3.0, 2.0, 4.0]).reshape(2, 2, order="F")
    b = np.array([5.0, 7.0, 6.0, 8.0]).reshape(2, 2, order="F")
    x = oracle.mrdivide(a, b)
    assert np.max(np.abs(x.reshape(-1, order="F") - [3.0, 2.0, -2.0, -1.0])) < 1e-12
    assert np.array_equal(oracle.mrdivide(np.array([[2.0, 4.0, 6.0]]), np.array([[2.0]])), [[1.0, 2.0, 3.0]])


# ---- round 3: min / max with indices, std, nnz / any / all, cumulative scans (oracle.c "reductions next to sum / mean") ----
def test_minmax_with_indices_reference_kats(oracle):
    """max.rs / min.rs unit tests: `max_vector_with_indices` (:2470-2477), `max_row_vector_reduces_across_columns` (:2568-2575),
    `max_matrix_default_dimension` (:2607-2626), `max_with_omitnan` (:2651-2659), `max_omitnan_all_nan_slice` (:2662-2676), and the
    rules they do not cover but the code states: first occurrence, -0 < +0 (min.rs:1519-1531), first NaN wins (min.rs:1443-1455)."""
    v, i = oracle.minmax_dim(np.array([[3.0], [1.0], [5.0]]), 0, True)
    assert v.item() == 5.0 and i.item() == 3.0
    v, i = oracle.minmax_dim(np.array([[3.0, 1.0, 5.0]]), 1, True)
    assert v.item() == 5.0 and i.item() == 3.0
    A = np.array([3.0, 4.0, 1.0, 2.0, 5.0, 6.0]).reshape(2, 3, order="F")
    v, i = oracle.minmax_dim(A, 0, True)
    assert v.tolist() == [[4.0, 2.0, 6.0]] and i.tolist() == [[2.0, 2.0, 2.0]]
    v, i = oracle.minmax_dim(np.array([[np.nan], [4.0], [2.0]]), 0, True, omitnan=True)
    assert v.item() == 4.0 and i.item() == 2.0
    v, i = oracle.minmax_dim(np.array([[np.nan], [np.nan]]), 0, True, omitnan=True)
    assert np.isnan(v.item()) and np.isnan(i.item())
    v, i = oracle.minmax_dim(np.array([[2.0, np.nan, 1.0, np.nan]]), 1, False)  # includenan: the FIRST NaN
    assert np.isnan(v.item()) and i.item() == 2.0
    v, i = oracle.minmax_dim(np.array([[0.0, -0.0, 0.0, -0.0]]), 1, False)
    assert np.signbit(v.item()) and i.item() == 2.0
    v, i = oracle.minmax_dim(np.array([[-0.0, 0.0, -0.0]]), 1, True)
    assert not np.signbit(v.item()) and i.item() == 2.0
    v, i = oracle.minmax_dim(np.array([[7.0, 1.0, 1.0, 7.0]]), 1, False)  # ties: first occurrence
    assert v.item() == 1.0 and i.item() == 2.0


def test_std_truth_and_scans_reference_rules(oracle):
    """std.rs:858-935 (sample / population, NaN modes, single value), nnz.rs:358, any.rs:722-733, all.rs:671-703,
    cumsum.rs:586-650 / cumprod.rs (include: NaN from the first NaN on; omit: NaNs leave the running value; reverse)."""
    x = np.array([[1.0, 2.0, 3.0, 4.0]])
    assert abs(oracle.std_dim(x, 1).item() - np.std([1, 2, 3, 4], ddof=1)) < 1e-15
    assert abs(oracle.std_dim(x, 1, population=True).item() - np.std([1, 2, 3, 4])) < 1e-15
    assert oracle.std_dim(np.array([[5.0]]), None).item() == 0.0
    assert np.isnan(oracle.std_dim(np.array([[1.0, np.nan, 3.0]]), 1).item())
    assert abs(oracle.std_dim(np.array([[1.0, np.nan, 3.0]]), 1, omitnan=True).item() - np.sqrt(2.0)) < 1e-15
    assert np.isnan(oracle.std_dim(np.array([[np.nan, np.nan]]), 1, omitnan=True).item())
    t = np.array([[0.0, np.nan, 2.0], [0.0, 0.0, np.nan]])
    assert oracle.truth_dim(t, 0, "nnz").tolist() == [[0.0, 1.0, 2.0]]
    assert oracle.truth_dim(t, 0, "any").tolist() == [[0.0, 1.0, 1.0]] and oracle.truth_dim(t, 0, "any", True).tolist() == [[0.0, 0.0, 1.0]]
    assert oracle.truth_dim(t, 0, "all").tolist() == [[0.0, 0.0, 1.0]]  # NaNs are skipped; [2, NaN] has no zero left
    assert oracle.truth_dim(np.array([[np.nan, np.nan]]), 1, "all").item() == 1.0
    assert oracle.cumulative(np.array([[1.0, 2.0], [3.0, 4.0]]), 0).tolist() == [[1.0, 2.0], [4.0, 6.0]]
    r = oracle.cumulative(np.array([[1.0, np.nan, 3.0]]), 1)
    assert r[0, 0] == 1.0 and np.isnan(r[0, 1]) and np.isnan(r[0, 2])
    assert oracle.cumulative(np.array([[1.0, np.nan, 3.0]]), 1, omitnan=True).tolist() == [[1.0, 1.0, 4.0]]
    r = oracle.cumulative(np.array([[1.0, np.nan, 3.0]]), 1, reverse=True)
    assert np.isnan(r[0, 0]) and np.isnan(r[0, 1]) and r[0, 2] == 3.0
    assert oracle.cumulative(np.array([[2.0, 3.0, 4.0]]), 1, prod=True).tolist() == [[2.0, 6.0, 24.0]]
    assert oracle.cumulative(np.array([[2.0, np.nan, 4.0]]), 1, prod=True, omitnan=True, reverse=True).tolist() == [[8.0, 4.0, 4.0]]


def test_oracle_reductions_and_scans_on_random_3d_shapes_against_numpy(oracle):
    """The GPU shape-fuzz tests (tests/test_gpu_fuzz_nd.py, test_gpu_reductions2.py) check the device against these oracle functions on
    3-D shapes along every dimension: here the oracle itself against independent numpy formulations on quarter-valued data (sums and
    scans exact, ties frequent), so a dimension-handling slip in the checker cannot hide one in the product."""
    rng = np.random.default_rng(2024)
    for _ in range(25):
        shape = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 17, 33, 64, 100], 3))
        x = np.round(rng.uniform(-8, 8, shape) * 4) / 4
        x[rng.random(shape) < 0.1] = 0.0
        for dim in (0, 1, 2):
            for is_max in (False, True):
                v, i = oracle.minmax_dim(x, dim, is_max)
                ref = x.max(axis=dim, keepdims=True) if is_max else x.min(axis=dim, keepdims=True)
                assert np.array_equal(v, ref), (shape, dim, is_max)
                # the index is 1-based and names an element that attains the extreme value; no earlier element does
                taken = np.take_along_axis(x, (i - 1).astype(np.int64), axis=dim)
                assert np.array_equal(taken, ref), (shape, dim, is_max)
                first = np.argmax(x == ref, axis=dim)
                # (-0.0 == +0.0 in numpy: where both occur the CPU rule orders -0 below +0, so only compare slices without a signed-zero tie)
                no_zero_tie = ~np.any((x == 0.0) & (ref == 0.0), axis=dim)
                assert np.array_equal((np.squeeze(i, axis=dim) - 1)[no_zero_tie], first[no_zero_tie]), (shape, dim, is_max)
            assert np.array_equal(oracle.truth_dim(x, dim, "nnz"), np.count_nonzero(x, axis=dim, keepdims=True).astype(np.float64))
            assert np.array_equal(oracle.truth_dim(x, dim, "any"), x.any(axis=dim, keepdims=True).astype(np.float64))
            assert np.array_equal(oracle.truth_dim(x, dim, "all"), x.all(axis=dim, keepdims=True).astype(np.float64))
            want = x.std(axis=dim, ddof=1, keepdims=True) if shape[dim] > 1 else np.zeros_like(x.sum(axis=dim, keepdims=True))
            assert np.max(np.abs(oracle.std_dim(x, dim) - want)) <= 1e-12 * max(1.0, float(np.abs(want).max())), (shape, dim)
            for reverse in (False, True):
                flip = (lambda a: np.flip(a, axis=dim)) if reverse else (lambda a: a)
                assert np.array_equal(oracle.cumulative(x, dim, reverse=reverse), flip(np.cumsum(flip(x), axis=dim))), (shape, dim, reverse)
        s = np.where(rng.random(shape) < 0.5, 1.0, -1.0)
        for dim in (0, 1, 2):
            assert np.array_equal(oracle.cumulative(s, dim, prod=True), np.cumprod(s, axis=dim)), (shape, dim)


def test_oracle_special_hooks_against_numpy(oracle):
    """image_normalize for any batch extent, covariance of many samples of a few variables, least squares of regression shapes, dot and
    the moments along a dimension: what the new GPU paths of late round 3 are compared with, against numpy's own formulations."""
    rng = np.random.default_rng(11)
    for shape in [(1, 30, 22), (3, 5, 7), (300, 6, 5), (1000, 4, 4)]:
        x = rng.uniform(0.0, 1.0, shape)
        got = oracle.image_normalize(x, 1e-6, gain=1.5, bias=0.1, gamma=1.8, clamp_zero=True)
        mu = x.mean(axis=(1, 2), keepdims=True)
        var = ((x - mu) ** 2).mean(axis=(1, 2), keepdims=True)
        want = np.maximum((x - mu) / np.sqrt(var + 1e-6) * 1.5 + 0.1, 0.0) ** 1.8
        assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, float(np.abs(want).max())), shape
    for rows, cols in [(5000, 1), (4096, 8), (20003, 17), (9000, 24)]:
        x = rng.uniform(-1, 1, (rows, cols)) + np.arange(cols)
        assert np.allclose(oracle.covariance(x, False), np.cov(x, rowvar=False).reshape(cols, cols), rtol=1e-10, atol=1e-12), (rows, cols)
        assert np.allclose(oracle.covariance(x, True), np.cov(x, rowvar=False, bias=True).reshape(cols, cols), rtol=1e-10, atol=1e-12), (rows, cols)
    for m, n, nrhs in [(500, 7, 1), (1200, 17, 3)]:
        A, B = rng.uniform(-1, 1, (m, n)), rng.uniform(-1, 1, (m, nrhs))
        want = np.linalg.lstsq(A, B, rcond=None)[0]
        assert np.max(np.abs(oracle.mldivide_svd(A, B) - want)) <= 1e-10 * max(1.0, float(np.abs(want).max())), (m, n, nrhs)
    a, b = rng.uniform(-1, 1, (32, 300)), rng.uniform(-1, 1, (32, 300))
    for dim in (0, 1):
        prod = oracle.binary("mul", a, b)
        assert np.allclose(oracle.reduce_sum(prod, [dim]), (a * b).sum(axis=dim, keepdims=True), rtol=1e-13, atol=1e-14)
        assert np.allclose(oracle.reduce_sum(a, [dim], mean=True), a.mean(axis=dim, keepdims=True), rtol=1e-13, atol=1e-15)
