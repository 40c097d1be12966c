"""Random 3-D shapes (pre x red x post with every factor drawn from 1, small, odd, around the kernels' switching points) through the
reduction, scan, broadcast and transpose entry points, on an f64 and on a precision-32 provider.  The data are quarters (or +-1 for
products), so sums, scans and their f32 roundings are exact and every comparison is bit for bit against the oracle / numpy: the point is
the dispatch (which kernel a shape lands on, its edges), not the arithmetic - that has its own tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = [1, 2, 3, 5, 8, 17, 31, 33, 64, 65, 100, 127, 129]
LONG = [255, 256, 257, 511, 513, 1000, 2047, 2049, 4099, 9001]


def _shapes(rng, count, limit=1 << 21):
    out = []
    while len(out) < count:
        pick = lambda: int(rng.choice(LONG)) if rng.random() < 0.35 else int(rng.choice(SMALL))
        s = (pick(), pick(), pick())
        if s[0] * s[1] * s[2] <= limit:
            out.append(s)
    return out


def _quarters(rng, shape):
    x = np.round(rng.uniform(-8, 8, shape) * 4) / 4
    x[rng.random(shape) < 0.1] = 0.0
    return x


def _dl(p, h, shape):
    return p.download(h).reshape(shape, order="F")


@pytest.fixture(scope="module")
def prov32(built):
    from runmat_amd import HipProvider

    p = HipProvider(0, precision="F32")
    yield p
    p.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_reductions_along_every_dim_of_random_3d_shapes(prov, prov32, oracle, precision):
    p = prov if precision == "f64" else prov32
    rng = np.random.default_rng(101 if precision == "f64" else 102)
    for shape in _shapes(rng, 36):
        x = _quarters(rng, shape)
        h = p.upload(x)
        for dim in (0, 1, 2):
            red = shape[dim]
            oshape = tuple(1 if d == dim else e for d, e in enumerate(shape))
            tag = (precision, shape, dim)
            assert np.array_equal(_dl(p, p.reduce_sum_dim(h, dim), oshape), x.sum(axis=dim, keepdims=True)), tag + ("sum",)
            got = _dl(p, p.reduce_mean_dim(h, dim), oshape)
            want = x.sum(axis=dim, keepdims=True) / red
            assert np.max(np.abs(got - want)) <= (4.5e-16 if precision == "f64" else 1.2e-7) * max(1.0, float(np.abs(want).max())), tag + ("mean",)
            for is_max, f in ((False, p.reduce_min_dim), (True, p.reduce_max_dim)):
                r = f(h, dim)
                wv, wi = oracle.minmax_dim(x, dim, is_max)
                assert np.array_equal(_dl(p, r.values, oshape).view(np.uint64), wv.view(np.uint64)), tag + ("minmax", is_max)
                assert np.array_equal(_dl(p, r.indices, oshape), wi), tag + ("argminmax", is_max)
            assert np.array_equal(_dl(p, p.reduce_nnz_dim(h, dim), oshape), np.count_nonzero(x, axis=dim, keepdims=True).astype(np.float64)), tag + ("nnz",)
            assert np.array_equal(_dl(p, p.reduce_any_dim(h, dim), oshape), x.any(axis=dim, keepdims=True).astype(np.float64)), tag + ("any",)
            assert np.array_equal(_dl(p, p.reduce_all_dim(h, dim), oshape), x.all(axis=dim, keepdims=True).astype(np.float64)), tag + ("all",)
            sd = _dl(p, p.reduce_std_dim(h, dim), oshape)
            want = oracle.std_dim(x, dim)
            tol = 1e-12 if precision == "f64" else 2e-6
            assert np.max(np.abs(sd - want)) <= tol * max(1.0, float(np.abs(want).max())), tag + ("std",)
        # the whole-array forms
        assert p.download(p.reduce_sum(h))[0] == x.sum(), (precision, shape, "sum all")
        assert p.download(p.reduce_nnz(h))[0] == np.count_nonzero(x), (precision, shape, "nnz all")
        assert p.download(p.reduce_max(h))[0] == x.max() and p.download(p.reduce_min(h))[0] == x.min(), (precision, shape, "minmax all")
        p.free(h)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_scans_along_every_dim_of_random_3d_shapes(prov, prov32, precision):
    p = prov if precision == "f64" else prov32
    rng = np.random.default_rng(103 if precision == "f64" else 104)
    for it, shape in enumerate(_shapes(rng, 36)):
        x = _quarters(rng, shape)
        s = np.where(rng.random(shape) < 0.5, 1.0, -1.0)  # products of +-1: exact at any length
        s[rng.random(shape) < 0.002] = 2.0
        hx, hs = p.upload(x), p.upload(s)
        for dim in (0, 1, 2):
            rev = bool((it + dim) & 1)
            flip = (lambda a: np.flip(a, axis=dim)) if rev else (lambda a: a)
            got = _dl(p, p.cumsum_scan(hx, dim, reverse=rev), shape)
            assert np.array_equal(got, flip(np.cumsum(flip(x), axis=dim))), (precision, shape, dim, rev, "cumsum")
            if shape[dim] <= 1000:  # a 2 every ~500 elements: stays far below f32's 2^127
                got = _dl(p, p.cumprod_scan(hs, dim, reverse=rev), shape)
                assert np.array_equal(got, flip(np.cumprod(flip(s), axis=dim))), (precision, shape, dim, rev, "cumprod")
        p.free(hx)
        p.free(hs)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_broadcast_pairs_of_random_3d_shapes(prov, prov32, precision):
    """a (with some extents collapsed to 1) + b (with others collapsed): every mix of stride-0 / stride-1 leading dimensions, short and
    long, through the per-op kernels; exact on quarters"""
    p = prov if precision == "f64" else prov32
    rng = np.random.default_rng(105 if precision == "f64" else 106)
    for shape in _shapes(rng, 40, limit=1 << 20):
        ka = rng.random(3) < 0.6
        kb = rng.random(3) < 0.6
        sa = tuple(e if k else 1 for e, k in zip(shape, ka))
        sb = tuple(e if k else 1 for e, k in zip(shape, kb))
        a, b = _quarters(rng, sa), _quarters(rng, sb)
        ha, hb = p.upload(a), p.upload(b)
        want = a + b
        h = p.elem_add(ha, hb)
        assert int(np.prod(h.shape)) == want.size
        assert np.array_equal(p.download(h).reshape(want.shape, order="F"), want), (precision, sa, sb, "add")
        h2 = p.elem_lt(ha, hb)
        assert np.array_equal(p.download(h2).reshape(want.shape, order="F"), (a < b).astype(np.float64)), (precision, sa, sb, "lt")
        for x in (ha, hb, h, h2):
            p.free(x)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_transposes_and_products_of_views_random_2d_shapes(prov, prov32, precision):
    p = prov if precision == "f64" else prov32
    rng = np.random.default_rng(107 if precision == "f64" else 108)
    dims = SMALL + LONG
    for _ in range(40):
        m, n = int(rng.choice(dims)), int(rng.choice(dims))
        x = _quarters(rng, (m, n))
        h = p.upload(x)
        t = p.transpose(h)
        assert tuple(t.shape) == (n, m)
        assert np.array_equal(p.download(p.unary_neg(t)).reshape((n, m), order="F"), -x.T), (precision, m, n, "neg(T)")
        assert np.array_equal(p.download(p.reduce_sum_dim(t, 0)).ravel(), x.sum(axis=1)), (precision, m, n, "sum(T,1)")
        # integers in +-8: the products are exact in f32 too (k * 8 * 8 < 2^24 for k <= 9001)
        xi = np.round(x)
        hi = p.upload(xi)
        g = p.matmul(p.transpose(hi), hi)  # A' * A read in place
        assert np.array_equal(p.download(g).reshape((n, n), order="F"), xi.T @ xi), (precision, m, n, "A'*A")
        p.free(h)
        p.free(hi)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_dot_and_moments_of_random_3d_shapes(prov, prov32, precision):
    """dot along every dimension (products of quarters and their sums are exact in f64; on f32 storage the result rounds once) and
    reduce_moments_nd over dimension subsets (one fused pass for the first dimension, the rest on the intermediates)"""
    p = prov if precision == "f64" else prov32
    rng = np.random.default_rng(109 if precision == "f64" else 110)
    r32 = (lambda a: a.astype(np.float32).astype(np.float64)) if precision == "f32" else (lambda a: a)
    for it, shape in enumerate(_shapes(rng, 30)):
        a, b = _quarters(rng, shape), _quarters(rng, shape)
        ha, hb = p.upload(a), p.upload(b)
        for dim in (0, 1, 2):
            oshape = tuple(1 if d == dim else e for d, e in enumerate(shape))
            got = _dl(p, p.dot(ha, hb, dim), oshape)
            assert np.array_equal(got, r32((a * b).sum(axis=dim, keepdims=True))), (precision, shape, dim, "dot")
        dims = [(0,), (1,), (2,), (0, 1), (1, 2), (0, 2), (0, 1, 2)][it % 7]
        mean, ex2 = p.reduce_moments_nd(ha, dims)
        wm, w2 = a, a * a
        for d in dims:
            wm, w2 = wm.mean(axis=d, keepdims=True), w2.mean(axis=d, keepdims=True)
        tol = 1e-13 if precision == "f64" else 2e-6
        gm, g2 = _dl(p, mean, wm.shape), _dl(p, ex2, w2.shape)
        assert np.max(np.abs(gm - wm)) <= tol * max(1.0, float(np.abs(wm).max())), (precision, shape, dims, "mean")
        assert np.max(np.abs(g2 - w2)) <= tol * max(1.0, float(np.abs(w2).max())), (precision, shape, dims, "ex2")
        p.free(ha)
        p.free(hb)
