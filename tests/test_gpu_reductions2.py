"""GPU parity of the reduction hooks added in round 3 (reduce2.hip) against the oracle's restatement of the CPU builtins:
`reduce_min_dim` / `reduce_max_dim` with indices (lib.rs:2864-2883, integer work: bit-exact), `reduce_std(_dim)` (:2786-2802),
`reduce_nnz / any / all (_dim)` (:2730-2742, :2803-2850, exact), `cumsum_scan` / `cumprod_scan` (:2884-2915)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    """bit-equal including the sign of zero; NaN matches NaN"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nan = np.isnan(a) & np.isnan(b)
    return a.shape == b.shape and bool(np.all(nan | (a.view(np.uint64) == b.view(np.uint64))))


def _special(rng, shape):
    """values with ties, signed zeros, infinities and NaNs sprinkled in"""
    x = rng.integers(-3, 4, size=shape).astype(np.float64)
    x[rng.random(shape) < 0.05] = -0.0
    x[rng.random(shape) < 0.03] = np.inf
    x[rng.random(shape) < 0.03] = -np.inf
    x[rng.random(shape) < 0.04] = np.nan
    return x


@pytest.mark.parametrize("shape", [(1, 1), (7, 1), (1, 9), (33, 5), (257, 130), (1000, 3), (3, 1000), (64, 64, 3), (5, 7, 11), (2, 70000),
                                   (70000, 2), (513, 37), (1001, 9), (2049, 7), (4099, 3)])  # odd extents: the unaligned-pair kernels
def test_minmax_dim_values_and_indices_bit_exact(prov, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    for x in (rng.uniform(-1, 1, shape), _special(rng, shape)):
        h = prov.upload(x)
        for dim in range(len(shape)):
            for is_max in (False, True):
                for omit in (False, True):
                    r = (prov.reduce_max_dim if is_max else prov.reduce_min_dim)(h, dim, omitnan=omit)
                    wv, wi = oracle.minmax_dim(x, dim, is_max, omit)
                    gv = prov.download(r.values).reshape(wv.shape, order="F")
                    gi = prov.download(r.indices).reshape(wi.shape, order="F")
                    assert tuple(r.values.shape) == wv.shape and tuple(r.indices.shape) == wi.shape
                    assert _same(gv, wv) and _same(gi, wi), (shape, dim, is_max, omit)
                    prov.free(r.values)
                    prov.free(r.indices)
        prov.free(h)


def test_minmax_dim_edge_slices(prov, oracle):
    """NaN-only columns, all-equal columns, +-0 mixes, a single NaN at the end / the start, +-inf"""
    cols = [[np.nan] * 5, [2.0] * 5, [0.0, -0.0, 0.0, -0.0, 0.0], [-0.0, 0.0, -0.0, 0.0, -0.0], [1.0, 2.0, 3.0, 4.0, np.nan],
            [np.nan, 4.0, 3.0, 2.0, 1.0], [np.inf, -np.inf, np.inf, -np.inf, 0.0], [5.0, 1.0, 1.0, 5.0, 5.0]]
    x = np.array(cols).T.copy()
    h = prov.upload(x)
    for dim, xx in ((0, x), (1, x)):
        for is_max in (False, True):
            for omit in (False, True):
                r = (prov.reduce_max_dim if is_max else prov.reduce_min_dim)(h, dim, omitnan=omit)
                wv, wi = oracle.minmax_dim(xx, dim, is_max, omit)
                assert _same(prov.download(r.values).reshape(wv.shape, order="F"), wv)
                assert _same(prov.download(r.indices).reshape(wi.shape, order="F"), wi)


@pytest.mark.parametrize("dim", [0, 1])
def test_minmax_dim_full_size_against_numpy(prov, dim):
    """8192 x 8192 (BASELINE E2): numpy's argmin / argmax take the first occurrence too and let a NaN win, so on data without
    signed-zero ties they are an independent check at a size the oracle's loops are too slow for."""
    n = 8192
    h = prov.fill_uniform(77, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for is_max in (False, True):
        r = (prov.reduce_max_dim if is_max else prov.reduce_min_dim)(h, dim)
        gv, gi = prov.download(r.values).ravel(), prov.download(r.indices).ravel()
        wi = (np.argmax if is_max else np.argmin)(x, axis=dim)
        assert np.array_equal(gi, wi + 1.0)
        assert np.array_equal(gv, np.take_along_axis(x, np.expand_dims(wi, dim), dim).ravel())
        prov.free(r.values)
        prov.free(r.indices)
    prov.free(h)


@pytest.mark.parametrize("shape", [(1, 1), (9, 1), (1, 9), (300, 40), (40, 300), (17, 5, 9), (100000, 2), (2, 100000), (513, 37), (2049, 7)])
def test_std_matches_welford_oracle(prov, oracle, shape):
    rng = np.random.default_rng(5 + sum(shape))
    x = rng.normal(3.0, 2.0, shape)
    xn = x.copy()
    xn[rng.random(shape) < 0.02] = np.nan
    for data in (x, xn):
        h = prov.upload(data)
        for dim in [None] + list(range(len(shape))):
            for pop in (False, True):
                for omit in (False, True):
                    want = oracle.std_dim(data, dim, pop, omit)
                    norm = "population" if pop else "sample"
                    got = prov.download(prov.reduce_std(h, norm, omit) if dim is None else prov.reduce_std_dim(h, dim, norm, omit))
                    got = got.reshape(want.shape, order="F")
                    assert np.array_equal(np.isnan(got), np.isnan(want)), (shape, dim, pop, omit)
                    ok = ~np.isnan(want)
                    # chunks (and the threads of a contiguous slice) merge with Chan's formula where the CPU runs one Welford chain:
                    # a few ulp of the result, plus the cancellation scale eps * max|x| for nearly equal values
                    scale = float(np.nanmax(np.abs(data)))
                    assert np.all(np.abs(got[ok] - want[ok]) <= 2e-13 * np.abs(want[ok]) + 8 * 2.3e-16 * scale), (shape, dim, pop, omit)
        prov.free(h)


@pytest.mark.parametrize("shape", [(1, 1), (6, 1), (33, 7), (257, 129), (4, 5, 6), (3, 90000), (1001, 9), (4099, 3)])
def test_truth_reductions_exact(prov, oracle, shape):
    rng = np.random.default_rng(9 + sum(shape))
    x = (rng.random(shape) < 0.3).astype(np.float64) * rng.uniform(-2, 2, shape)
    x[rng.random(shape) < 0.05] = np.nan
    x[rng.random(shape) < 0.05] = -0.0
    h = prov.upload(x)
    calls = {"nnz": (prov.reduce_nnz, prov.reduce_nnz_dim), "any": (prov.reduce_any, prov.reduce_any_dim), "all": (prov.reduce_all, prov.reduce_all_dim)}
    for op, (f_all, f_dim) in calls.items():
        for omit in ((False,) if op == "nnz" else (False, True)):
            for dim in [None] + list(range(len(shape))):
                want = oracle.truth_dim(x, dim, op, omit)
                if op == "nnz":
                    got = f_all(h) if dim is None else f_dim(h, dim)
                else:
                    got = f_all(h, omit) if dim is None else f_dim(h, dim, omit)
                assert np.array_equal(prov.download(got).reshape(want.shape, order="F"), want), (op, omit, dim)
    prov.free(h)


@pytest.mark.parametrize("shape", [(1, 1), (11, 1), (1, 11), (65, 33), (33, 65), (6, 5, 4), (300000, 1), (1, 300000), (70000, 3), (3, 70000)])
def test_cumulative_scans(prov, oracle, shape):
    rng = np.random.default_rng(3 + sum(shape))
    ints = rng.integers(-5, 6, size=shape).astype(np.float64)  # integer-valued: every grouping of the additions is exact
    reals = rng.uniform(-1, 1, shape)
    nans = reals.copy()
    nans[rng.random(shape) < 0.01] = np.nan
    for data, exact in ((ints, True), (reals, False), (nans, False)):
        h = prov.upload(data)
        for dim in range(len(shape)):
            for rev in (False, True):
                for omit in (False, True):
                    want = oracle.cumulative(data, dim, prod=False, reverse=rev, omitnan=omit)
                    got = prov.download(prov.cumsum_scan(h, dim, rev, omit)).reshape(want.shape, order="F")
                    assert np.array_equal(np.isnan(got), np.isnan(want)), (shape, dim, rev, omit)
                    ok = ~np.isnan(want)
                    if exact:
                        assert np.array_equal(got[ok], want[ok]), (shape, dim, rev, omit)
                    else:
                        bound = 4 * shape[dim] * 2.3e-16 * max(1.0, float(np.nanmax(np.abs(want))))
                        assert np.max(np.abs(got[ok] - want[ok]), initial=0.0) <= bound, (shape, dim, rev, omit)
        prov.free(h)
    # products: values near 1 so that nothing over- or underflows
    p = 1.0 + rng.uniform(-1e-3, 1e-3, shape)
    p[rng.random(shape) < 0.01] = np.nan
    h = prov.upload(p)
    for dim in range(len(shape)):
        for rev, omit in ((False, False), (True, True)):
            want = oracle.cumulative(p, dim, prod=True, reverse=rev, omitnan=omit)
            got = prov.download(prov.cumprod_scan(h, dim, rev, omit)).reshape(want.shape, order="F")
            assert np.array_equal(np.isnan(got), np.isnan(want))
            ok = ~np.isnan(want)
            assert np.max(np.abs(got[ok] / want[ok] - 1.0), initial=0.0) <= 4 * shape[dim] * 2.3e-16
    prov.free(h)


@pytest.mark.parametrize("shape,dim", [((513, 700), 1), ((64, 256), 1), ((100, 1000, 3), 1), ((65, 300, 2), 1), ((300, 70), 1), ((7, 5000), 1),
                                       ((70, 33, 400), 2), ((8, 3000), 1), ((33, 1000, 2), 1),
                                       # short contiguous lines (len < 256 along dim 0): a tile of lines per block, thread per line
                                       ((32, 5000), 0), ((3, 70000), 0), ((255, 300), 0), ((17, 9, 40), 0)])
def test_strided_scan_is_the_cpu_sequence_bit_for_bit(prov, oracle, shape, dim):
    """Along a strided dimension every line is the CPU's own left-to-right chain - whether one thread walks it (many or short lines)
    or a block stages tiles of 64 lines x 64 steps through LDS and one wave runs the chains (few long lines): sums, products, both
    directions, both NaN modes."""
    rng = np.random.default_rng(21 + sum(shape))
    x = rng.uniform(-1, 1, shape)
    xn = x.copy()
    xn.ravel()[rng.integers(0, x.size, max(1, x.size // 97))] = np.nan
    h, hn = prov.upload(x), prov.upload(xn)
    for prod in (False, True):
        f = prov.cumprod_scan if prod else prov.cumsum_scan
        for reverse in (False, True):
            got = prov.download(f(h, dim, reverse=reverse)).reshape(shape, order="F")
            assert np.array_equal(got, oracle.cumulative(x, dim, prod=prod, reverse=reverse))
            for omit in (False, True):
                gn = prov.download(f(hn, dim, reverse=reverse, omitnan=omit)).reshape(shape, order="F")
                assert np.array_equal(gn, oracle.cumulative(xn, dim, prod=prod, reverse=reverse, omitnan=omit), equal_nan=True)


@pytest.mark.parametrize("shape,dim", [((40, 9001), 1), ((300, 20000), 1), ((17, 9001, 3), 1), ((64, 4096), 1), ((9, 70000), 1), ((2000, 8192), 1)])
def test_few_long_strided_lines_are_scanned_in_chunks(prov, oracle, shape, dim):
    """A handful of workgroups would each run chains of `len` steps: the line is cut into chunks (totals, carries, scan from the carry -
    reduce2.hip k_scan_lines_staged).  The grouping differs from the CPU's one chain, so: exact on quarters (no rounding anywhere), to
    rounding on general data, NaN policies and directions as ever."""
    rng = np.random.default_rng(5 + sum(shape))
    q = np.round(rng.uniform(-8, 8, shape) * 4) / 4
    x = rng.uniform(-1, 1, shape)
    xn = q.copy()
    xn.ravel()[rng.integers(0, q.size, max(1, q.size // 4001))] = np.nan
    hq, hx, hn = prov.upload(q), prov.upload(x), prov.upload(xn)
    n = shape[dim]
    for reverse in (False, True):
        got = prov.download(prov.cumsum_scan(hq, dim, reverse=reverse)).reshape(shape, order="F")
        assert np.array_equal(got, oracle.cumulative(q, dim, reverse=reverse)), (shape, reverse, "quarters")
        got = prov.download(prov.cumsum_scan(hx, dim, reverse=reverse)).reshape(shape, order="F")
        want = oracle.cumulative(x, dim, reverse=reverse)
        assert np.max(np.abs(got - want)) <= 4 * n * 2.3e-16 * max(1.0, float(np.abs(want).max())), (shape, reverse, "uniform")
        for omit in (False, True):
            gn = prov.download(prov.cumsum_scan(hn, dim, reverse=reverse, omitnan=omit)).reshape(shape, order="F")
            assert np.array_equal(gn, oracle.cumulative(xn, dim, reverse=reverse, omitnan=omit), equal_nan=True), (shape, reverse, omit)
    s = np.where(rng.random(shape) < 0.5, 1.0, -1.0)
    got = prov.download(prov.cumprod_scan(prov.upload(s), dim)).reshape(shape, order="F")
    assert np.array_equal(got, oracle.cumulative(s, dim, prod=True)), (shape, "cumprod of signs")


def test_reductions_random_shapes_fuzz(prov, oracle):
    """60 random 2-D shapes around the kernels' switching points (512 / 2048 extents, odd and even, few and many lines) through sum,
    min / max with indices (against the oracle: the rounding produces -0.0, which the CPU builtins order below +0), nnz, std and
    cumsum along both dimensions, against numpy."""
    rng = np.random.default_rng(31)
    longs = [511, 512, 513, 1023, 1025, 2047, 2048, 2049, 3000, 4097, 6001]
    for it in range(60):
        a, b = int(rng.choice(longs)), int(rng.integers(1, 48))
        shape = (a, b) if it % 2 == 0 else (b, a)
        x = np.round(rng.uniform(-8, 8, shape) * 4) / 4  # quarters: sums are exact, ties are frequent
        x[rng.random(shape) < 0.1] = 0.0
        h = prov.upload(x)
        for dim in (0, 1):
            assert np.array_equal(prov.download(prov.reduce_sum_dim(h, dim)).ravel(), x.sum(axis=dim)), (shape, dim, "sum")
            for is_max, f in ((False, prov.reduce_min_dim), (True, prov.reduce_max_dim)):
                r = f(h, dim)
                wv, wi = oracle.minmax_dim(x, dim, is_max)
                gv, gi = prov.download(r.values).ravel(), prov.download(r.indices).ravel()
                assert np.array_equal(gv.view(np.uint64), wv.ravel(order="F").view(np.uint64)) and np.array_equal(gi, wi.ravel(order="F")), (shape, dim, is_max)
            assert np.array_equal(prov.download(prov.reduce_nnz_dim(h, dim)).ravel(), np.count_nonzero(x, axis=dim).astype(np.float64)), (shape, dim, "nnz")
            sd = prov.download(prov.reduce_std_dim(h, dim)).ravel()
            want = x.std(axis=dim, ddof=1) if x.shape[dim] > 1 else np.zeros(x.shape[1 - dim])
            assert np.max(np.abs(sd - want)) <= 1e-12 * max(1.0, np.abs(want).max()), (shape, dim, "std")
            cs = prov.download(prov.cumsum_scan(h, dim)).reshape(shape, order="F")
            assert np.array_equal(cs, np.cumsum(x, axis=dim)), (shape, dim, "cumsum")  # exact on quarters
        prov.free(h)


def test_new_reductions_on_a_precision32_provider(built):
    """f32 storage: operands are widened (exact, order preserving), results narrowed once - indices stay exact integers."""
    from runmat_amd import HipProvider

    p32 = HipProvider(0, precision="F32")
    try:
        rng = np.random.default_rng(4)
        x = rng.uniform(-1, 1, (200, 30)).astype(np.float32).astype(np.float64)
        h = p32.upload(x)
        r = p32.reduce_max_dim(h, 0)
        assert np.array_equal(p32.download(r.indices).ravel(), np.argmax(x, axis=0) + 1.0)
        assert np.array_equal(p32.download(r.values).ravel(), x.max(axis=0))
        assert np.array_equal(p32.download(p32.reduce_nnz_dim(h, 1)).ravel(), np.count_nonzero(x, axis=1).astype(np.float64))
    finally:
        p32.close()
