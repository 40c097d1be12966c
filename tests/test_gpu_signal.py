"""GPU parity of conv1d / conv2d / hann_window / hamming_window / blackman_window (include/rmhip.h, signal_ops.hip).  The convolutions
are direct sums in the CPU's order with every product rounded before it is added: bit-exact against the oracle.  The windows take one
cosine per point (two for Blackman): within 2 ulp of the cosine's magnitude of the oracle's libm value, i.e. 4.5e-16 absolute."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = json.loads((Path(__file__).parent / "golden" / "signal_kats.json").read_text())


def bits_equal(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_reference_kats(prov):
    for k in K["conv1d"]:
        h = prov.conv1d(prov.upload(np.array(k["a"], dtype=np.float64).reshape(1, -1)), prov.upload(np.array(k["b"], dtype=np.float64).reshape(1, -1)), k["mode"])
        assert list(h.shape) == [1, len(k["out"])] and np.array_equal(prov.download(h), np.array(k["out"], dtype=np.float64)), k
    for k in K["conv2d"]:
        h = prov.conv2d(prov.upload(np.array(k["a"], dtype=np.float64)), prov.upload(np.array(k["b"], dtype=np.float64)), k["mode"])
        assert np.array_equal(prov.download_matrix(h), np.array(k["out"], dtype=np.float64)), k


@pytest.mark.parametrize("la,lb", [(1, 1), (5, 3), (3, 5), (64, 7), (1000, 100), (100, 1000), (4097, 255), (300, 5000), (70000, 33)], ids=str)
def test_conv1d(prov, oracle, la, lb):
    rng = np.random.default_rng(la * 7 + lb)
    a, b = rng.standard_normal(la), rng.standard_normal(lb)
    a[rng.integers(0, la)] = 0.0
    ha, hb = prov.upload(a.reshape(-1, 1)), prov.upload(b.reshape(1, -1))
    for mode in ("full", "same", "valid"):
        for orient in ("row", "column"):
            want = oracle.conv1d(a, b, mode)
            h = prov.conv1d(ha, hb, mode, orient)
            assert list(h.shape) == ([1, want.size] if orient == "row" else [want.size, 1])
            assert bits_equal(prov.download(h), want), (mode, orient)
    e = prov.conv1d(prov.upload(np.zeros((1, 0))), hb, "full", "column")
    assert list(e.shape) == [0, 1]


@pytest.mark.parametrize("sa,sb", [((1, 1), (1, 1)), ((3, 3), (3, 3)), ((9, 7), (3, 4)), ((3, 4), (9, 7)), ((64, 65), (5, 5)), ((200, 300), (17, 1)), ((200, 300), (1, 17)),
                                   ((40, 40), (70, 66)), ((5,), (3,)), ((130, 50), (64, 3)), ((70, 45), (2, 40)), ((3, 100), (5, 2))], ids=str)
def test_conv2d(prov, oracle, sa, sb):
    rng = np.random.default_rng(sum(sa) * 13 + sum(sb))
    a, b = rng.standard_normal(sa), rng.standard_normal(sb)
    up = lambda x: prov.upload(x.ravel(order="F"), x.shape)
    for mode in ("full", "same", "valid"):
        want = oracle.conv2d(a, b, mode)
        h = prov.conv2d(up(a), up(b), mode)
        assert list(h.shape) == list(want.shape), (mode, h.shape, want.shape)
        assert bits_equal(prov.download_matrix(h), want), mode
    with pytest.raises(Exception):
        prov.conv2d(prov.upload(np.zeros((2, 2, 2))), up(b))
    z = prov.upload(np.zeros((0, 3)))
    assert list(prov.conv2d(z, up(b)).shape) == [0, 0] and list(prov.conv2d(z, up(b), "same").shape) == [0, 3]


def test_conv2d_infinite_tap_against_a_zero(prov, oracle):
    """The CPU adds 0 * inf = NaN like any other term (conv2.rs:603-616 has no zero test; the in-process provider's skip of zero signal
    entries, simple_provider.rs:1860-1862, is not what the oracle restates)."""
    a = np.array([[0.0, 1.0], [2.0, 3.0]])
    b = np.array([[np.inf, 1.0], [1.0, 1.0]])
    want = oracle.conv2d(a, b)
    got = prov.download_matrix(prov.conv2d(prov.upload(a), prov.upload(b)))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).any()


def test_windows(prov, oracle):
    for n in (0, 1, 2, 5, 64, 255, 4096, 100001):
        for periodic in (False, True):
            for kind in ("hann", "hamming", "blackman"):
                h = getattr(prov, kind + "_window")(n, periodic)
                want = oracle.window(kind, n, periodic)
                assert list(h.shape) == [n, 1]
                if n:
                    assert np.max(np.abs(prov.download(h) - want.ravel())) <= 4.5e-16, (kind, n, periodic)


def test_full_size_image_filter(prov, oracle):
    """An 8192 x 8192 operand through a 5 x 5 'same' convolution: sampled outputs against the oracle on the sample's neighbourhood,
    and linearity (conv(a, 2 b) == 2 conv(a, b) exactly: scaling by two commutes with every rounding)."""
    n = 8192
    h = prov.fill_uniform(9, -1.0, 1.0, (n, n))
    rng = np.random.default_rng(1)
    b = rng.standard_normal((5, 5))
    hb = prov.upload(b)
    out = prov.conv2d(h, hb, "same")
    assert list(out.shape) == [n, n]
    x, y = prov.download_matrix(h), prov.download_matrix(out)
    for r, c in ((0, 0), (n - 1, n - 1), (4000, 17), (2, n - 3), (8191, 0), (1234, 5678)):
        r0, r1, c0, c1 = max(0, r - 4), min(n, r + 5), max(0, c - 4), min(n, c + 5)
        want = oracle.conv2d(x[r0:r1, c0:c1], b, "same")
        assert y[r, c] == want[r - r0, c - c0] or (r - r0 < 2 or c - c0 < 2 or r1 - r <= 2 or c1 - c <= 2) and abs(y[r, c] - want[r - r0, c - c0]) < 1e-12
    y2 = prov.download_matrix(prov.conv2d(h, prov.upload(2.0 * b), "same"))
    assert np.array_equal(y2, 2.0 * y)


def _nan(v):
    return np.nan if v == "nan" else v


def test_moving_window_kats(prov):
    for k in K["moving"]:
        x = np.array([_nan(v) for v in k["x"]], dtype=np.float64)
        h = prov.moving_window(prov.upload(x, k["shape"]), k["shape"], k["dim"], k["before"], k["after"], k["op"], k["endpoints"], k["nan"], k["norm"])
        assert np.array_equal(prov.download(h), np.array(k["out"], dtype=np.float64)), k


@pytest.mark.parametrize("shape,dim", [((9, 11), 0), ((9, 11), 1), ((300,), 0), ((5, 40, 3), 1), ((64, 3), 2), ((1, 1), 0), ((2049, 5), 0)], ids=str)
def test_moving_window(prov, oracle, shape, dim):
    rng = np.random.default_rng(sum(shape) + dim)
    x = rng.standard_normal(shape)
    x.ravel()[rng.integers(0, x.size, size=max(1, x.size // 17))] = np.nan
    x.ravel()[rng.integers(0, x.size, size=max(1, x.size // 23))] = 0.5          # repeated values (median ties)
    h = prov.upload(x.ravel(order="F"), shape)
    for before, after in ((1, 1), (0, 3), (4, 0), (7, 9)):
        for op in ("sum", "mean", "prod", "min", "max", "median", "std", "var"):
            for endpoints in ("shrink", "discard", 0.0, 1.0, float("nan")) + (() if op == "prod" else (2.5, -3.0)):
                for nan_mode in ("include", "omit"):
                    for norm in (("sample", "population") if op in ("std", "var") else ("sample",)):
                        want = oracle.moving_window(x, dim, before, after, op, endpoints, nan_mode, norm)
                        got = prov.moving_window(h, want.shape, dim, before, after, op, endpoints, nan_mode, norm)
                        assert list(got.shape) == list(want.shape)
                        assert bits_equal(prov.download(got).reshape(want.shape, order="F"), want), (before, after, op, endpoints, nan_mode, norm)
                        prov.free(got)


def test_moving_window_limits(prov):
    h = prov.upload(np.arange(200.0).reshape(200, 1))
    with pytest.raises(Exception):
        prov.moving_window(h, (200, 1), 0, 40, 40, "median")                       # more than 64 points per window
    with pytest.raises(Exception):
        prov.moving_window(h, (200, 1), 0, 1, 1, "prod", 2.0)                      # the CPU multiplies by powf(2.0, count)
    with pytest.raises(Exception):
        prov.moving_window(h, (199, 1), 0, 1, 1, "sum")                            # output shape of another request
    got = prov.moving_window(h, (200, 1), 0, 100, 100, "mean")                     # wide windows are fine for the other statistics
    assert abs(prov.download(got)[100] - 99.5) < 1e-12                             # [0, 200] clipped to the 200 points: their mean


def test_moving_mean_at_baseline_size(prov, oracle):
    n = 8192
    h = prov.fill_uniform(12, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for dim in (0, 1):
        got = prov.download_matrix(prov.moving_window(h, (n, n), dim, 2, 2, "mean"))
        sl = (slice(0, 64), slice(None)) if dim == 1 else (slice(None), slice(0, 64))
        want = oracle.moving_window(x[sl], dim, 2, 2, "mean")
        assert bits_equal(got[sl], want)


def test_polyval(prov, oracle):
    rng = np.random.default_rng(15)
    for m, shape in ((1, (3, 3)), (4, (5, 7)), (12, (1000,)), (5000, (33, 2)), (3, (0, 4))):
        c, x = rng.standard_normal(m), rng.uniform(-0.9, 0.9, shape)                    # (|x| < 1: a degree-4999 polynomial stays finite)
        hc, hx = prov.upload(c.reshape(1, -1)), prov.upload(x.ravel(order="F"), shape)
        for mu in (None, (0.25, 3.0), (-0.5, 2.0)):
            got = prov.polyval(hc, hx, mu)
            assert list(got.shape) == list(shape) and bits_equal(prov.download(got).reshape(shape, order="F"), oracle.polyval(c, x, mu)), (m, shape, mu)
    with pytest.raises(Exception):                                                  # inf intermediate: NaN + NaN i on the CPU, a complex result
        prov.polyval(prov.upload(np.array([[1e308, 1e308, 1.0]])), prov.upload(np.array([[1e10]])))
    big = prov.fill_uniform(2, -1.0, 1.0, (8192, 8192))
    coef = rng.standard_normal(9)
    y = prov.download_matrix(prov.polyval(prov.upload(coef.reshape(1, -1)), big))
    assert bits_equal(y[:, :8], oracle.polyval(coef, prov.download_matrix(big)[:, :8]))


def test_meshgrid_and_complex_zeros(prov, oracle):
    for axes in ([[1.0, 2.0, 3.0], [10.0, 20.0]], [[1.0, 2.0], [5.0], [7.0, 8.0, 9.0]], [[1.0, 2.0], [5.0, 6.0], [4.0]], [np.arange(300.0), np.arange(70.0)],
                 [[], [1.0, 2.0]]):
        got, want = prov.meshgrid(axes), oracle.meshgrid(axes)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert list(g.shape) == list(w.shape) and np.array_equal(prov.download(g).reshape(w.shape, order="F"), w)
    with pytest.raises(Exception):
        prov.meshgrid([[1.0]])
    z = prov.zeros_with_storage((3, 5), "complex")
    assert prov.is_complex(z) and list(z.shape) == [3, 5] and not prov.download(z).any() and prov.download(z).size == 15
    r = prov.zeros_with_storage((3, 5))
    assert not prov.is_complex(r) and not prov.download(r).any()


def test_filter_kats(prov):
    for k in K["filter"]:
        x = np.array(k["x"], dtype=np.float64)
        y, zf = prov.iir_filter(prov.upload(np.array(k["b"]).reshape(1, -1)), prov.upload(np.array(k["a"]).reshape(1, -1)), prov.upload(x, k["shape"]), k["dim"])
        assert list(y.shape) == k["shape"] and np.allclose(prov.download(y), k["y"], rtol=0, atol=1e-9), k
        if "zf" in k:
            assert list(zf.shape) == k["zshape"] and np.allclose(prov.download(zf), k["zf"], rtol=0, atol=1e-9), k


@pytest.mark.parametrize("shape,dim", [((40, 7, 3), 0), ((40, 7, 3), 1), ((40, 7, 3), 2), ((300, 260), 0), ((260, 300), 1), ((5,), 0), ((64, 300), 2)], ids=str)
def test_iir_filter(prov, oracle, shape, dim):
    rng = np.random.default_rng(sum(shape) + dim)
    x = rng.standard_normal(shape)
    hx = prov.upload(x.ravel(order="F"), shape)
    for b, a in (([0.2, 0.3, 0.1], [1.0, -0.5, 0.25]), ([1.0, -1.0], [1.0]), ([0.5], [2.0, 0.4, 0.1, 0.05]), ([3.0], [1.5]),
                 (list(rng.standard_normal(12)), [1.0] + list(0.05 * rng.standard_normal(9)))):
        hb, ha = prov.upload(np.array(b).reshape(1, -1)), prov.upload(np.array(a).reshape(1, -1))
        wy, wz = oracle.iir_filter(b, a, x, dim)
        y, zf = prov.iir_filter(hb, ha, hx, dim)
        assert list(y.shape) == list(shape) and bits_equal(prov.download(y).reshape(shape, order="F"), wy), (b, a)
        assert list(zf.shape) == list(wz.shape) and bits_equal(prov.download(zf).reshape(wz.shape, order="F"), wz), (b, a)
        if wz.size:
            zi = rng.standard_normal(wz.shape)
            wy2, wz2 = oracle.iir_filter(b, a, x, dim, zi)
            y2, zf2 = prov.iir_filter(hb, ha, hx, dim, prov.upload(zi.ravel(order="F"), zi.shape))
            assert bits_equal(prov.download(y2).reshape(shape, order="F"), wy2) and bits_equal(prov.download(zf2).reshape(wz.shape, order="F"), wz2)
        if len(a) == 1 and a[0] == 1.0:
            y3, _ = prov.iir_filter(hb, ha, hx, dim, unit_denominator=True)
            assert bits_equal(prov.download(y3).reshape(shape, order="F"), wy)


def test_iir_filter_limits_and_baseline_size(prov, oracle):
    b, a = prov.upload(np.array([[0.1, 0.2]])), prov.upload(np.array([[1.0, -0.3]]))
    with pytest.raises(Exception):
        prov.iir_filter(b, a, prov.upload(np.zeros((100000, 1))), 0)                 # one long channel: the host's one core is faster
    with pytest.raises(Exception):
        prov.iir_filter(b, prov.upload(np.array([[0.0, 1.0]])), prov.upload(np.zeros((8, 300))), 0)   # a(1) == 0
    with pytest.raises(Exception):
        prov.iir_filter(b, a, prov.upload(np.zeros((8, 300))), 0, prov.upload(np.zeros((2, 300))))    # one state per channel expected
    n = 8192
    h = prov.fill_uniform(14, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for dim in (0, 1):
        y, zf = prov.iir_filter(b, a, h, dim)
        sl = (slice(None), slice(0, 48)) if dim == 0 else (slice(0, 48), slice(None))
        wy, wz = oracle.iir_filter([0.1, 0.2], [1.0, -0.3], x[sl], dim)
        assert bits_equal(prov.download_matrix(y)[sl], wy) and bits_equal(prov.download(zf).reshape(zf.shape, order="F")[sl], wz)


@pytest.mark.parametrize("n,series,nq", [(2, 1, 5), (40, 3, 205), (1000, 1, 70000), (5000, 16, 4096)], ids=str)
def test_interp1(prov, oracle, n, series, nq):
    rng = np.random.default_rng(n + nq)
    x = np.cumsum(rng.uniform(0.1, 1.0, n))
    y = rng.standard_normal((n, series))
    q = rng.uniform(x[0] - 2, x[-1] + 2, nq)
    q[:min(nq, n):3] = x[:min(nq, n):3]                                              # exact hits, the last sample among them
    q[-1] = x[-1]
    if nq > 4:
        q[1], q[2] = np.nan, np.inf
    hx, hy, hq = prov.upload(x.reshape(-1, 1)), prov.upload(y), prov.upload(q.reshape(1, -1))
    for method in ("linear", "nearest"):
        for extrapolation in ("nan", "extrapolate", -3.5):
            want = oracle.interp1(x, y, q, method, extrapolation) if nq * series <= 20000 else None
            got = prov.download(prov.interp1(hx, hy, hq, n, series, nq, (nq, series), method, extrapolation)).reshape((nq, series), order="F")
            if want is None:                                                      # the python restatement is slow: a sample of the queries
                pick = rng.integers(0, nq, size=400)
                want, got = oracle.interp1(x, y, q[pick], method, extrapolation), got[pick]
            assert bits_equal(got, want), (method, extrapolation)
    with pytest.raises(Exception):
        prov.interp1(hx, hy, hq, n, series, nq, (nq + 1, series))


@pytest.mark.parametrize("ishape,kshape", [((9, 11), (3, 3)), ((9, 11), (2, 5)), ((9, 11), (4, 1)), ((9, 11), (1, 1)), ((4, 5, 3), (3, 3, 3)), ((7,), (3,)), ((6, 6), (9, 9)),
                                           ((300, 257), (5, 5)), ((3, 4, 5, 2), (2, 2, 2, 2)),
                                           ((70, 37, 3), (3, 4)), ((66, 18, 2, 2), (1, 7)), ((130, 20), (65, 2)), ((40, 90), (6, 70))], ids=str)
def test_imfilter(prov, oracle, ishape, kshape):
    rng = np.random.default_rng(sum(ishape) * 3 + sum(kshape))
    img, ker = rng.standard_normal(ishape), rng.standard_normal(kshape)
    hi, hk = prov.upload(img.ravel(order="F"), ishape), prov.upload(ker.ravel(order="F"), kshape)
    for padding in (0.0, 2.5, "replicate", "symmetric", "circular"):
        for shape in ("same", "full", "valid"):
            for mode in ("correlation", "convolution"):
                want = oracle.imfilter(img, ker, padding, shape, mode)
                got = prov.imfilter(hi, hk, padding, shape, mode)
                assert list(got.shape) == list(want.shape), (padding, shape, mode, got.shape, want.shape)
                assert bits_equal(prov.download(got).reshape(want.shape, order="F"), want), (padding, shape, mode)
                prov.free(got)


def test_imfilter_limits_and_baseline_size(prov, oracle):
    for k in K["imfilter"]:
        got = prov.imfilter(prov.upload(np.array(k["image"], dtype=np.float64), k["ishape"]), prov.upload(np.array(k["kernel"], dtype=np.float64), k["kshape"]), k["padding"], k["shape"])
        assert list(got.shape) == k["oshape"] and np.array_equal(prov.download(got), k["out"]), k
    with pytest.raises(Exception):
        prov.imfilter(prov.upload(np.zeros((4, 4))), prov.upload(np.zeros((0, 3))))
    with pytest.raises(Exception):
        prov.imfilter(prov.upload(np.zeros((4, 4))), prov.upload(np.ones((2, 2, 2))))     # the image has no third axis
    n = 8192
    h = prov.fill_uniform(19, -1.0, 1.0, (n, n))
    ker = np.random.default_rng(2).standard_normal((5, 5))
    y = prov.download_matrix(prov.imfilter(h, prov.upload(ker), "replicate"))
    x = prov.download_matrix(h)
    want = oracle.imfilter(x[:40, :40], ker, "replicate")
    assert bits_equal(y[:38, :38], want[:38, :38])                                    # (the block's own lower / right padding differs from the image's interior)
    want = oracle.imfilter(x[-40:, -40:], ker, "replicate")
    assert bits_equal(y[-38:, -38:], want[-38:, -38:])


def test_polyder_polyint(prov, oracle):
    """lib.rs:1674-1710 against the restatement of simple_provider.rs:3137-3215 (bit-exact) and the reference's own vectors."""
    shaped = lambda values, shape: np.array(values, dtype=np.float64).reshape(shape, order="F")
    up = lambda x: prov.upload(np.asarray(x, dtype=np.float64).ravel(order="F"), x.shape)

    def same(handle, want):
        values, shape = want
        assert list(handle.shape) == shape, (handle.shape, shape)
        assert bits_equal(prov.download(handle).ravel(), values), (prov.download(handle), values)

    for k in K["polyder"]:
        p = shaped(k["p"], k["pshape"])
        if "q" not in k:
            got = prov.polyder_single(up(p))
        elif k.get("quotient"):
            got, den = prov.polyder_quotient(up(p), up(shaped(k["q"], k["qshape"])))
            assert list(den.shape) == k["dshape"] and np.allclose(prov.download(den), k["den"], rtol=0, atol=1e-12)
        else:
            got = prov.polyder_product(up(p), up(shaped(k["q"], k["qshape"])))
        assert list(got.shape) == k["oshape"] and np.allclose(prov.download(got), k["out"], rtol=0, atol=1e-12), k
    for k in K["polyint"]:
        got = prov.polyint(up(shaped(k["p"], k["pshape"])), k["constant"])
        assert list(got.shape) == k["oshape"] and np.allclose(prov.download(got), k["out"], rtol=0, atol=1e-12), k

    rng = np.random.default_rng(77)
    for np_, nq in ((0, 0), (1, 1), (0, 3), (3, 0), (2, 1), (1, 4), (5, 3), (3, 9), (64, 65), (700, 300), (1, 1000)):
        for orient in ((1, -1), (-1, 1)):
            p, q = rng.standard_normal(np_).reshape(orient), rng.standard_normal(nq).reshape(orient[::-1])
            if np_ > 3:
                p.ravel()[:2] = (0.0, 1e-13)                                           # trimmed from the derivative
            hp, hq = up(p), up(q)
            same(prov.polyder_single(hp), oracle.polyder_single(p))
            same(prov.polyder_product(hp, hq), oracle.polyder_product(p, q))
            num, den = prov.polyder_quotient(hp, hq)
            wnum, wden = oracle.polyder_quotient(p, q)
            same(num, wnum)
            same(den, wden)
            same(prov.polyint(hp, -1.25), oracle.polyint(p, -1.25))
    special = np.array([[np.nan, 0.0, np.inf, -2.0, 0.0, 1.0]])
    same(prov.polyder_single(up(special)), oracle.polyder_single(special))
    same(prov.polyder_product(up(special), up(special.T.copy())), oracle.polyder_product(special, special.T))
    same(prov.polyint(up(special), np.nan), oracle.polyint(special, np.nan))
    zeros = np.zeros((1, 5))
    same(prov.polyder_product(up(zeros), up(zeros)), oracle.polyder_product(zeros, zeros))      # nothing left: [0], 1 x 1
    for bad in (lambda: prov.polyder_single(up(np.ones((2, 2)))), lambda: prov.polyint(up(np.ones((2, 3))), 0.0),
                lambda: prov.polyder_product(up(np.ones((1, 3))), up(np.ones((2, 2))))):
        with pytest.raises(Exception):
            bad()
