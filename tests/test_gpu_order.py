"""GPU parity of the order-statistics hooks through the C ABI (include/rmhip.h "order statistics along a dimension"): cummin_scan /
cummax_scan, diff_dim, sort_dim, reduce_median(_dim).  Copies of input elements, positions and single rounded operations: bit-exact
against the oracle, on the reference's own known-answer vectors, on shapes that reach every launch regime (wave- and thread-scans, chunked
lines, LDS-only and multi-pass sorting networks), with NaNs, ties, signed zeros and infinities; at BASELINE sizes through properties."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = json.loads((Path(__file__).parent / "golden" / "order_hooks_kats.json").read_text())


def arr(v, shape=None):
    a = np.array([np.nan if x is None else x for x in v], dtype=np.float64)
    return a.reshape(shape, order="F") if shape is not None else a


def bits_equal(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return False
    g, w = got.copy(), want.copy()
    gn, wn = np.isnan(g), np.isnan(w)
    if not np.array_equal(gn, wn):
        return False
    g[gn], w[wn] = 0.0, 0.0                        # any NaN payload is a NaN; everything else to the bit (signed zeros included)
    return np.array_equal(g.view(np.uint64), w.view(np.uint64))


def spiced(rng, shape, nan_frac=0.0, ties=False):
    x = rng.standard_normal(shape)
    if ties:
        x = np.round(x * 2.0) / 2.0                  # many equal values, +0 and -0 among them
        x[x == 0.0] = np.where(rng.random(np.count_nonzero(x == 0.0)) < 0.5, -0.0, 0.0)
    flat = x.reshape(-1)
    if flat.size > 8:
        flat[rng.integers(0, flat.size, 3)] = [np.inf, -np.inf, 0.0]
    if nan_frac > 0:
        flat[rng.random(flat.size) < nan_frac] = np.nan
    return x


def test_reference_kats(prov):
    for k in K["cumextreme"]:
        h = prov.upload(arr(k["data"], k["shape"]))
        r = (prov.cummax_scan if k["op"] == "max" else prov.cummin_scan)(h, k["dim"], k["reverse"], k["omit"])
        assert bits_equal(prov.download_matrix(r.values).ravel(order="F"), arr(k["values"])), k
        assert bits_equal(prov.download_matrix(r.indices).ravel(order="F"), arr(k["indices"])), k
    for k in K["diff"]:
        h = prov.upload(arr(k["data"], k["shape"]))
        d = prov.diff_dim(h, k["order"], k["dim"])
        assert list(d.shape) == k["out_shape"] and bits_equal(prov.download(d).ravel(order="F"), arr(k["out"])), k
    for k in K["median"]:
        h = prov.upload(arr(k["data"], k["shape"]))
        m = prov.reduce_median(h) if k["dim"] == "all" else prov.reduce_median_dim(h, k["dim"])
        assert bits_equal(prov.download_matrix(m).ravel(order="F"), arr(k["out"])), k
    for k in K["sort"]:
        h = prov.upload(arr(k["data"], k["shape"]))
        r = prov.sort_dim(h, k["dim"], "descend" if k["descend"] else "ascend", "abs" if k["abs"] else "auto")
        assert bits_equal(r.values.ravel(order="F"), arr(k["sorted"])) and bits_equal(r.indices.ravel(order="F"), arr(k["indices"])), k


SCAN_SHAPES = [(1, 1), (5, 1), (1, 7), (64, 3), (65, 2), (1000, 3), (3, 1000), (257, 129), (40000, 1), (1, 40000), (70001, 2), (2, 70001),
               (7, 11, 13), (2, 3000, 3), (300000, 1), (16, 100000)]


@pytest.mark.parametrize("shape", SCAN_SHAPES, ids=str)
def test_cummin_cummax_match_the_oracle(prov, oracle, shape):
    rng = np.random.default_rng(hash(shape) % 1000)
    for nan_frac, ties in ((0.0, False), (0.02, True)):
        x = spiced(rng, shape, nan_frac, ties)
        h = prov.upload(x)
        for dim in range(len(shape)):
            for is_max in (False, True):
                for reverse in (False, True):
                    for omit in (False, True):
                        r = (prov.cummax_scan if is_max else prov.cummin_scan)(h, dim, reverse, omit)
                        wv, wi = oracle.cumextreme(x, dim, is_max, reverse, omit)
                        assert bits_equal(prov.download_matrix(r.values), wv), (shape, dim, is_max, reverse, omit, "values")
                        assert bits_equal(prov.download_matrix(r.indices), wi), (shape, dim, is_max, reverse, omit, "indices")
                        prov.free(r.values)
                        prov.free(r.indices)
        prov.free(h)


@pytest.mark.parametrize("shape", [(1, 1), (9, 1), (1, 9), (300, 7), (7, 300), (5, 6, 7), (2, 1, 50), (100003, 2)], ids=str)
def test_diff_dim_matches_the_oracle(prov, oracle, shape):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    h = prov.upload(x)
    for dim in range(len(shape) + 1):
        for order in (0, 1, 2, 3):
            for colmaj in (False, True):
                d = prov.diff_dim(h, order, dim, colmaj)
                if order == 0:
                    assert bits_equal(prov.download_matrix(d), x)
                else:
                    want, wshape = oracle.diff(x, order, dim, colmaj)
                    got = prov.download(d)
                    # shapes as diff_tensor_host reports them (trailing ones appended up to the dimension)
                    assert [int(s) for s in d.shape][:len(wshape)] == [int(s) for s in wshape] or want.size == 0, (shape, dim, order, d.shape, wshape)
                    assert bits_equal(np.asarray(got).ravel(order="F"), want.ravel(order="F")), (shape, dim, order, colmaj)
                prov.free(d)
    prov.free(h)


SORT_SHAPES = [(1, 1), (2, 1), (3, 1), (1, 5), (64, 1), (100, 3), (3, 100), (2048, 2), (2049, 1), (5000, 3), (3, 5000), (7, 9, 11), (40, 2100, 2),
               (70001, 1), (1, 70001), (300000, 1)]


@pytest.mark.parametrize("shape", SORT_SHAPES, ids=str)
def test_sort_dim_matches_the_oracle(prov, oracle, shape):
    rng = np.random.default_rng(len(shape) * 100 + shape[0])
    for nan_frac, ties in ((0.0, False), (0.03, True)):
        x = spiced(rng, shape, nan_frac, ties)
        h = prov.upload(x)
        for dim in range(len(shape) + 1):
            for order in ("ascend", "descend"):
                for cmp in ("auto", "abs"):
                    r = prov.sort_dim(h, dim, order, cmp)
                    if dim < len(shape):
                        ws, wi = oracle.sort_dim(x, dim, order == "descend", cmp == "abs")
                    else:
                        ws, wi = x, np.ones_like(x)
                    assert bits_equal(r.values, ws), (shape, dim, order, cmp, "values")
                    assert bits_equal(r.indices, wi), (shape, dim, order, cmp, "indices")
        prov.free(h)


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (5, 1), (4, 1), (100, 3), (3, 100), (101, 7), (2049, 2), (4, 4100), (6, 5, 4), (70001, 1), (3, 70000), (300001, 1), (2, 200000), (200000, 3), (3, 1, 140000)], ids=str)
def test_median_matches_the_oracle(prov, oracle, shape):
    rng = np.random.default_rng(shape[0])
    for nan_frac, ties in ((0.0, False), (0.0, True), (0.001, False)):
        x = spiced(rng, shape, nan_frac, ties)
        h = prov.upload(x)
        for dim in range(len(shape)):
            m = prov.reduce_median_dim(h, dim)
            assert bits_equal(prov.download_matrix(m), oracle.median_dim(x, dim)), (shape, dim, nan_frac, ties)
            prov.free(m)
        m = prov.reduce_median(h)
        want = oracle.median_all(x)
        got = float(prov.download_matrix(m).reshape(-1)[0])
        assert (np.isnan(got) and np.isnan(want)) or got == want, (shape, got, want)
        assert m.shape == (1, 1)
        prov.free(m)
        prov.free(h)


def test_empty_and_degenerate_operands(prov):
    h = prov.upload(np.zeros((0, 3)))
    r = prov.cummin_scan(h, 0)
    assert r.values.shape == (0, 3) and r.indices.shape == (0, 3)
    s = prov.sort_dim(h, 0)
    assert s.values.shape == (0, 3)
    m = prov.reduce_median_dim(h, 0)                      # median.rs:668-672: an empty slice is NaN
    assert m.shape == (1, 3) and np.all(np.isnan(prov.download_matrix(m)))
    d = prov.diff_dim(h, 1, 0)
    assert d.shape[0] == 0
    with pytest.raises(Exception):
        prov.cummin_scan(h, 2)                            # dim >= rank: the caller never sends it (cummin.rs:676-688); refused here


def test_full_size_properties(prov):
    """BASELINE's 8192 x 8192 operand: sortedness, permutation, agreement of the pieces with each other (no oracle at this size)."""
    n = 8192
    h = prov.fill_uniform(77, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for dim in (0, 1):
        r = prov.sort_dim(h, dim)
        assert np.all(np.diff(r.values, axis=dim) >= 0)
        assert np.array_equal(np.take_along_axis(x, (r.indices - 1).astype(np.int64), axis=dim), r.values)
        med = prov.download_matrix(prov.reduce_median_dim(h, dim))
        mid = np.take(r.values, [n // 2 - 1, n // 2], axis=dim)
        assert np.array_equal(med.reshape(-1), (0.5 * (np.take(mid, 0, axis=dim) + np.take(mid, 1, axis=dim))).reshape(-1))
        c = prov.cummin_scan(h, dim)
        cv = prov.download_matrix(c.values)
        assert np.array_equal(cv, np.minimum.accumulate(x, axis=dim))
        ci = prov.download_matrix(c.indices)
        assert np.array_equal(np.take_along_axis(x, (ci - 1).astype(np.int64), axis=dim), cv)
        prov.free(c.values)
        prov.free(c.indices)
        d = prov.download_matrix(prov.diff_dim(h, 1, dim, True))
        assert np.array_equal(d, np.diff(x, axis=dim))


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (64, 1), (65, 3), (1024, 1), (1025, 1), (300, 700), (7, 5, 11), (2000003, 1), (3, 700001)], ids=str)
def test_find_matches_the_oracle(prov, oracle, shape):
    rng = np.random.default_rng(shape[0])
    for density in (0.0, 0.01, 0.5, 1.0):
        x = rng.standard_normal(shape) * (rng.random(shape) < density)
        if x.size > 4:
            x.reshape(-1)[rng.integers(0, x.size, 2)] = [np.nan, -0.0]
        h = prov.upload(x)
        for limit in (None, 0, 1, 7, x.size, x.size + 5):
            for direction in ("first", "last"):
                got = prov.find(h, limit, direction)
                want = oracle.find(x, limit, direction == "last")
                for g, w, name in zip((got.linear, got.rows, got.cols, got.values), want, ("linear", "rows", "cols", "values")):
                    assert tuple(g.shape) == w.shape, (shape, density, limit, direction, name, g.shape, w.shape)
                    assert bits_equal(prov.download_matrix(g), w), (shape, density, limit, direction, name)
                    prov.free(g)
        prov.free(h)


def test_find_at_full_size(prov):
    n = 8192
    h = prov.fill_uniform(5, -1.0, 1.0, (n, n))
    m = prov.elem_gt(h, prov.fill((n, n), 0.999))
    x = prov.download_matrix(m)
    got = prov.find(m)
    want = np.flatnonzero(x.ravel(order="F")) + 1
    assert np.array_equal(prov.download_matrix(got.linear).ravel(), want)
    assert np.array_equal(prov.download_matrix(got.rows).ravel(), (want - 1) % n + 1) and np.array_equal(prov.download_matrix(got.cols).ravel(), (want - 1) // n + 1)
    assert np.array_equal(prov.download_matrix(prov.find(m, 3, "last").linear).ravel(), want[::-1][:3])


def test_median_selection_lands_on_signed_zeros(prov, oracle):
    """Long lines take the radix-selection path: a median that falls on a zero must carry the sign of the zero the stable sort puts
    there (odd length), or of the sum of the two middle zeros (even length)."""
    rng = np.random.default_rng(17)
    for n in (262145, 262144):
        for trial in range(4):
            x = rng.standard_normal(n)
            zeros = rng.random(n) < 0.5                                   # half of the elements are zeros: the median is one
            x[zeros] = np.where(rng.random(np.count_nonzero(zeros)) < 0.5, -0.0, 0.0)
            nz = np.count_nonzero(~zeros)
            x[~zeros] = np.abs(x[~zeros]) * np.where(np.arange(nz) % 2 == 0, 1.0, -1.0)
            h = prov.upload(x.reshape(-1, 1))
            got = prov.download_matrix(prov.reduce_median(h))
            want = oracle.median_dim(x.reshape(-1, 1), 0)
            assert bits_equal(got, want), (n, trial, got, want)
            prov.free(h)
    x = np.full(200000, -0.0)
    assert bits_equal(prov.download_matrix(prov.reduce_median(prov.upload(x.reshape(-1, 1)))), np.array([[-0.0]]))
