"""The oracle's restatements of conv / conv2 / the window generators against the reference's own unit-test vectors
(tests/golden/signal_kats.json) and against numpy / scipy where they compute the same thing."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle

K = json.loads((Path(__file__).parent / "golden" / "signal_kats.json").read_text())


def test_reference_kats():
    for k in K["conv1d"]:
        assert np.array_equal(oracle.conv1d(k["a"], k["b"], k["mode"]), np.array(k["out"], dtype=np.float64)), k
    for k in K["conv2d"]:
        assert np.array_equal(oracle.conv2d(np.array(k["a"], dtype=np.float64), np.array(k["b"], dtype=np.float64), k["mode"]), np.array(k["out"], dtype=np.float64)), k


def test_against_numpy():
    rng = np.random.default_rng(4)
    for la, lb in ((1, 1), (5, 3), (3, 5), (64, 7), (100, 100), (257, 16)):
        a, b = rng.standard_normal(la), rng.standard_normal(lb)
        assert np.allclose(oracle.conv1d(a, b), np.convolve(a, b), rtol=0, atol=1e-13)
        assert np.allclose(oracle.conv1d(a, b, "same"), np.convolve(a, b)[(lb - 1) // 2:(lb - 1) // 2 + la], rtol=0, atol=1e-13)
        if la >= lb:
            assert np.allclose(oracle.conv1d(a, b, "valid"), np.convolve(a, b, "valid"), rtol=0, atol=1e-13)
    from scipy.signal import convolve2d
    a, b = rng.standard_normal((9, 7)), rng.standard_normal((3, 4))
    # conv2.rs:609-614 indexes the kernel from its far corner while walking the output forwards: the sum it forms is the convolution
    # with the kernel rotated by 180 degrees (its own test `conv2_same_flips_kernel` pins the signs); the restatement follows the code
    assert np.allclose(oracle.conv2d(a, b), convolve2d(a, b[::-1, ::-1], "full"), rtol=0, atol=1e-13)
    assert oracle.conv2d(a, b, "valid").shape == (7, 4) and oracle.conv2d(a, b, "same").shape == (9, 7)
    assert oracle.conv2d(np.zeros((0, 3)), b).shape == (0, 0) and oracle.conv2d(np.zeros((0, 3)), b, "same").shape == (0, 3)
    for n in (1, 2, 5, 64, 255):
        assert np.allclose(oracle.window("hann", n).ravel(), np.hanning(n) if n > 1 else [1.0], rtol=0, atol=1e-15)
        assert np.allclose(oracle.window("hamming", n).ravel(), np.hamming(n) if n > 1 else [1.0], rtol=0, atol=1e-15)
        assert np.allclose(oracle.window("blackman", n).ravel(), np.blackman(n) if n > 1 else [1.0], rtol=0, atol=1e-15)
        if n > 1:
            assert np.allclose(oracle.window("hann", n, True).ravel(), np.hanning(n + 1)[:-1], rtol=0, atol=1e-15)
    assert oracle.window("hann", 0).shape == (0, 1)


def test_moving_window_kats_and_definitions():
    for k in K["moving"]:
        x = np.array([np.nan if v == "nan" else v for v in k["x"]], dtype=np.float64).reshape(k["shape"], order="F")
        got = oracle.moving_window(x, k["dim"], k["before"], k["after"], k["op"], k["endpoints"], k["nan"], k["norm"])
        assert np.array_equal(got.ravel(order="F"), np.array(k["out"], dtype=np.float64)), k
    rng = np.random.default_rng(8)
    x = rng.standard_normal((9, 11))
    for dim in (0, 1):
        for before, after in ((1, 1), (0, 3), (4, 0), (20, 20)):
            lo = lambda p: max(0, p - before)
            hi = lambda p: min(x.shape[dim], p + after + 1)
            win = lambda p: np.take(x, range(lo(p), hi(p)), axis=dim)
            n = x.shape[dim]
            for op, f in (("sum", np.sum), ("mean", np.mean), ("prod", np.prod), ("min", np.min), ("max", np.max), ("median", np.median),
                          ("var", lambda w, axis: np.var(w, axis=axis, ddof=1 if w.shape[axis] > 1 else 0)),
                          ("std", lambda w, axis: np.std(w, axis=axis, ddof=1 if w.shape[axis] > 1 else 0))):
                want = np.stack([f(win(p), axis=dim) for p in range(n)], axis=dim)
                got = oracle.moving_window(x, dim, before, after, op)
                assert got.shape == x.shape and np.allclose(got, want, rtol=1e-12, atol=1e-13), (dim, before, after, op)
            d = oracle.moving_window(x, dim, before, after, "sum", "discard")
            assert d.shape[dim] == max(0, n - before - after)
            if d.size:
                assert np.allclose(d, np.take(oracle.moving_window(x, dim, before, after, "sum"), range(before, n - after), axis=dim))
    z = oracle.moving_window(x, 0, 2, 2, "sum", 10.0)                                # padding counts as values
    assert np.isclose(z[0, 0], x[:3, 0].sum() + 20.0)
    assert np.isnan(oracle.moving_window(x, 0, 2, 2, "mean", float("nan"))[0, 0])      # NaN padding, include-mode
    assert np.isclose(oracle.moving_window(x, 0, 2, 2, "mean", float("nan"), "omit")[0, 0], x[:3, 0].mean())
    assert oracle.moving_window(np.zeros((0, 3)), 0, 1, 1, "sum").shape == (0, 3)
    assert oracle.moving_window(x, 2, 1, 1, "max").shape == (9, 11, 1)                 # a dimension beyond the rank


def test_polyval_and_meshgrid():
    rng = np.random.default_rng(6)
    c, x = rng.standard_normal(7), rng.standard_normal((5, 4))
    assert np.allclose(oracle.polyval(c, x), np.polyval(c, x), rtol=1e-13, atol=1e-13)
    assert np.allclose(oracle.polyval(c, x, (0.3, 2.0)), np.polyval(c, (x - 0.3) / 2.0), rtol=1e-13, atol=1e-13)
    assert np.array_equal(oracle.polyval([2.0], x), np.full(x.shape, 2.0)) and np.array_equal(oracle.polyval([1.0, 2.0, 3.0], np.array([2.0])), [11.0])
    X, Y = oracle.meshgrid([[1.0, 2.0, 3.0], [10.0, 20.0]])
    assert np.array_equal(X, [[1, 2, 3], [1, 2, 3]]) and np.array_equal(Y, [[10, 10, 10], [20, 20, 20]])
    X, Y, Z = oracle.meshgrid([[1.0, 2.0], [5.0], [7.0, 8.0, 9.0]])
    want = np.meshgrid([1.0, 2.0], [5.0], [7.0, 8.0, 9.0])
    assert all(np.array_equal(a, b) for a, b in zip((X, Y, Z), want))
    assert oracle.meshgrid([[1.0, 2.0], [5.0, 6.0], [4.0]])[2].shape == (2, 2)


def test_filter_kats_and_scipy():
    for k in K["filter"]:
        y, zf = oracle.iir_filter(k["b"], k["a"], np.array(k["x"], dtype=np.float64).reshape(k["shape"], order="F"), k["dim"])
        assert np.allclose(y.ravel(order="F"), k["y"], rtol=0, atol=1e-9), k
        if "zf" in k:
            assert list(zf.shape) == k["zshape"] and np.allclose(zf.ravel(order="F"), k["zf"], rtol=0, atol=1e-9), k
    from scipy.signal import lfilter, lfilter_zi
    rng = np.random.default_rng(12)
    x = rng.standard_normal((40, 7, 3))
    for b, a in (([0.2, 0.3, 0.1], [1.0, -0.5, 0.25]), ([1.0, -1.0], [1.0]), ([0.5], [2.0, 0.4, 0.1, 0.05]), ([3.0], [1.5])):
        for dim in (0, 1, 2):
            y, zf = oracle.iir_filter(b, a, x, dim)
            order = max(len(b), len(a))
            if order > 1:
                want, wz = lfilter(b, a, x, axis=dim, zi=np.zeros([order - 1 if d == dim else x.shape[d] for d in range(3)]))
                assert np.allclose(y, want, rtol=1e-12, atol=1e-12) and np.allclose(zf, wz, rtol=1e-12, atol=1e-12)
                zi = rng.standard_normal(wz.shape)
                y2, zf2 = oracle.iir_filter(b, a, x, dim, zi)
                w2, wz2 = lfilter(b, a, x, axis=dim, zi=zi)
                assert np.allclose(y2, w2, rtol=1e-12, atol=1e-12) and np.allclose(zf2, wz2, rtol=1e-12, atol=1e-12)
            else:
                assert np.allclose(y, lfilter(b, a, x, axis=dim), rtol=1e-14, atol=0) and zf.shape[dim] == 0


def test_interp1_against_numpy():
    for k in K["interp1"]:
        got = oracle.interp1(k["x"], np.array(k["y"], dtype=np.float64).reshape(k["yshape"], order="F"), k["xq"])
        assert np.allclose(got.ravel(order="F"), k["out"], rtol=0, atol=1e-12), k
    rng = np.random.default_rng(31)
    x = np.cumsum(rng.uniform(0.1, 1.0, 40))
    y = rng.standard_normal((40, 3))
    q = np.concatenate([rng.uniform(x[0] - 2, x[-1] + 2, 200), x[[0, 5, 39]], [np.nan, np.inf]])
    got = oracle.interp1(x, y, q, "linear", "nan")
    inside = np.isfinite(q) & (q >= x[0]) & (q <= x[-1])
    for s in range(3):
        assert np.allclose(got[inside, s], np.interp(q[inside], x, y[:, s]), rtol=1e-13, atol=1e-13)
    assert np.isnan(got[~inside]).all()
    ex = oracle.interp1(x, y, q, "linear", "extrapolate")
    lo = np.isfinite(q) & (q < x[0])
    assert np.allclose(ex[lo, 0], y[0, 0] + (q[lo] - x[0]) / (x[1] - x[0]) * (y[1, 0] - y[0, 0]), rtol=1e-12)
    assert np.all(oracle.interp1(x, y, q, "linear", 7.5)[np.isfinite(q) & ~inside] == 7.5)
    nn = oracle.interp1(x, y, q, "nearest", "nan")
    near = np.abs(q[inside, None] - x[None, :]).argmin(axis=1)
    assert np.array_equal(nn[inside, 1], y[near, 1])
    assert np.array_equal(oracle.interp1([0.0, 1.0, 2.0], [10.0, 20.0, 30.0], [0.5, 1.5], "nearest")[:, 0], [10.0, 20.0])   # ties to the left


def test_imfilter_against_scipy():
    for k in K["imfilter"]:
        got = oracle.imfilter(np.array(k["image"], dtype=np.float64).reshape(k["ishape"], order="F"), np.array(k["kernel"], dtype=np.float64).reshape(k["kshape"], order="F"),
                              k["padding"], k["shape"])
        assert list(got.shape) == k["oshape"] and np.array_equal(got.ravel(order="F"), k["out"]), k
    from scipy import ndimage
    rng = np.random.default_rng(44)
    img = rng.standard_normal((9, 11))
    for kshape in ((3, 3), (2, 5), (4, 1), (1, 1)):
        ker = rng.standard_normal(kshape)
        origin = [0] * len(kshape)                                                 # both centre a kernel at floor(extent / 2)
        for pad, smode in (("replicate", "nearest"), ("symmetric", "mirror"), ("circular", "wrap"), (0.0, "constant"), (2.5, "constant")):
            want = ndimage.correlate(img, ker, mode=smode, cval=pad if smode == "constant" else 0.0, origin=origin)
            assert np.allclose(oracle.imfilter(img, ker, pad), want, rtol=0, atol=1e-13), (kshape, pad)
            wconv = ndimage.convolve(img, ker, mode=smode, cval=pad if smode == "constant" else 0.0, origin=origin)
            got = oracle.imfilter(img, ker, pad, mode="convolution")
            assert got.shape == img.shape
            if all(k % 2 for k in kshape):
                assert np.allclose(got, wconv, rtol=0, atol=1e-13), (kshape, pad)
    ker = rng.standard_normal((3, 4))
    from scipy.signal import correlate2d
    assert np.allclose(oracle.imfilter(img, ker, 0.0, "full"), correlate2d(img, ker, "full"), atol=1e-13)
    assert oracle.imfilter(img, ker, 0.0, "valid").shape == (7, 8) and oracle.imfilter(img, np.ones((12, 3)), 0.0, "valid").shape == (0, 9)
    vol = rng.standard_normal((4, 5, 3))
    k3 = rng.standard_normal((3, 3, 3))
    assert np.allclose(oracle.imfilter(vol, k3, "replicate"), ndimage.correlate(vol, k3, mode="nearest"), rtol=0, atol=1e-13)


def test_polyder_polyint_reference_kats():
    shaped = lambda values, shape: np.array(values, dtype=np.float64).reshape(shape, order="F")
    for k in K["polyder"]:
        p = shaped(k["p"], k["pshape"])
        if "q" not in k:
            got = [oracle.polyder_single(p)]
        elif k.get("quotient"):
            got = list(oracle.polyder_quotient(p, shaped(k["q"], k["qshape"])))
        else:
            got = [oracle.polyder_product(p, shaped(k["q"], k["qshape"]))]
        assert got[0][1] == k["oshape"] and np.allclose(got[0][0], k["out"], rtol=0, atol=1e-12), k
        if k.get("quotient"):
            assert got[1][1] == k["dshape"] and np.allclose(got[1][0], k["den"], rtol=0, atol=1e-12), k
    for k in K["polyint"]:
        values, shape = oracle.polyint(shaped(k["p"], k["pshape"]), k["constant"])
        assert shape == k["oshape"] and np.allclose(values, k["out"], rtol=0, atol=1e-12), k


def test_polyder_polyint_against_numpy_and_edge_rules():
    rng = np.random.default_rng(12)
    for np_, nq in ((1, 1), (2, 1), (1, 4), (5, 3), (3, 9), (12, 12)):
        p, q = rng.standard_normal(np_), rng.standard_normal(nq)
        assert np.allclose(np.polyder(p) if np_ > 1 else [0.0], oracle.polyder_single(p.reshape(1, -1))[0], rtol=0, atol=1e-12)
        want = np.polyder(np.polymul(p, q))
        assert np.allclose(want if want.size else [0.0], oracle.polyder_product(p.reshape(1, -1), q.reshape(1, -1))[0], rtol=0, atol=1e-10)
        (num, _), (den, _) = oracle.polyder_quotient(p.reshape(1, -1), q.reshape(-1, 1))
        want = np.polysub(np.polymul(np.polyder(p), q), np.polymul(p, np.polyder(q)))
        assert np.allclose(np.trim_zeros(np.where(np.abs(want) > 1e-12, want, 0.0), "f") if np.any(np.abs(want) > 1e-12) else [0.0], num, rtol=0, atol=1e-10) or np_ == 1 or nq == 1
        assert np.allclose(np.polymul(q, q), den, rtol=0, atol=1e-10)
        assert np.allclose(np.polyint(p, k=2.5), oracle.polyint(p.reshape(-1, 1), 2.5)[0], rtol=0, atol=1e-14)
    assert oracle.polyder_single(np.zeros((1, 0)))[1] == [1, 1] and oracle.polyder_single(np.array([[0.0, 0.0, 1e-13, 0.0]]))[0].tolist() == [0.0]
    assert oracle.polyder_single(np.array([[0.0, 0.0, 3.0, 1.0]]))[0].tolist() == [3.0]           # leading zeros trimmed
    assert oracle.polyder_single(np.array([[np.nan, 2.0, 1.0]]))[0].tolist() == [2.0]             # a NaN is not above the threshold
    assert oracle.polyint(np.zeros((0, 0)), 4.0) [0].tolist() == [4.0] and oracle.polyint(np.zeros((0, 0)), 4.0)[1] == [1, 1]
    with pytest.raises(ValueError):
        oracle.polyder_single(np.ones((2, 2)))
