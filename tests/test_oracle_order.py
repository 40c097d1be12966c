"""The oracle's restatements of cummin / cummax, diff, median and sort against the reference's own known-answer tests
(tests/golden/order_hooks_kats.json: the vectors of cummin.rs / cummax.rs / diff.rs / median.rs / sort.rs unit tests), plus the
properties the GPU parity tests rely on (numpy agreement on inputs without ties / NaNs, stability, NaN placement)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle

K = json.loads((Path(__file__).parent / "golden" / "order_hooks_kats.json").read_text())


def arr(v, shape=None):
    a = np.array([np.nan if x is None else x for x in v], dtype=np.float64)
    return a.reshape(shape, order="F") if shape is not None else a


def same(got, want):
    got, want = np.asarray(got, dtype=np.float64).ravel(order="F"), np.asarray(want, dtype=np.float64).ravel(order="F")
    return got.shape == want.shape and np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("k", K["cumextreme"], ids=lambda k: f"{k['op']}-{k['shape']}-{k['dim']}-{k['reverse']}-{k['omit']}")
def test_cumextreme_reference_kats(k):
    v, i = oracle.cumextreme(arr(k["data"], k["shape"]), k["dim"], k["op"] == "max", k["reverse"], k["omit"])
    assert same(v, arr(k["values"])) and same(i, arr(k["indices"]))


@pytest.mark.parametrize("k", K["diff"], ids=lambda k: f"{k['shape']}-{k['order']}-{k['dim']}")
def test_diff_reference_kats(k):
    out, shape = oracle.diff(arr(k["data"], k["shape"]), k["order"], k["dim"])
    assert same(out, arr(k["out"])) and list(shape) == k["out_shape"]


@pytest.mark.parametrize("k", K["median"], ids=lambda k: f"{k['shape']}-{k['dim']}")
def test_median_reference_kats(k):
    x = arr(k["data"], k["shape"])
    if k["dim"] == "all":
        assert oracle.median_all(x) == k["out"][0] and oracle.median_all(x, successive=True) == k["out"][0]
    else:
        assert same(oracle.median_dim(x, k["dim"]), arr(k["out"]))


@pytest.mark.parametrize("k", K["sort"], ids=lambda k: f"{k['shape']}-{k['dim']}-{k['descend']}-{k['abs']}")
def test_sort_reference_kats(k):
    s, i = oracle.sort_dim(arr(k["data"], k["shape"]), k["dim"], k["descend"], k["abs"])
    assert same(s, arr(k["sorted"])) and same(i, arr(k["indices"]))


def test_against_numpy_on_generic_inputs():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((37, 11, 5))
    for dim in range(3):
        v, i = oracle.cumextreme(x, dim, False)
        assert np.array_equal(v, np.minimum.accumulate(x, axis=dim))
        assert np.array_equal(np.take_along_axis(x, (i - 1).astype(np.int64), axis=dim), v)
        v, _ = oracle.cumextreme(x, dim, True, reverse=True)
        assert np.array_equal(v, np.flip(np.maximum.accumulate(np.flip(x, dim), axis=dim), dim))
        s, i = oracle.sort_dim(x, dim)
        assert np.array_equal(s, np.sort(x, axis=dim)) and np.array_equal(i - 1, np.argsort(x, axis=dim, kind="stable"))
        s, i = oracle.sort_dim(x, dim, descend=True)
        assert np.array_equal(s, -np.sort(-x, axis=dim))
        assert np.array_equal(oracle.median_dim(x, dim), np.median(x, axis=dim, keepdims=True))
        d, shape = oracle.diff(x, 2, dim, column_major=True)
        assert np.array_equal(d.reshape(shape, order="F"), np.diff(x, 2, axis=dim))
    # the reference's own output order of diff along dim >= 1: k fastest inside every (before, after) line
    d, shape = oracle.diff(x, 1, 1)
    want = np.diff(x, 1, axis=1)
    assert np.array_equal(d.reshape(shape[1], shape[0], shape[2], order="F"), want.transpose(1, 0, 2))


def test_sort_is_stable_and_places_nans_and_zeros_like_the_reference():
    x = np.array([0.0, -0.0, 2.0, np.nan, -0.0, 0.0, 2.0, np.nan, -3.0]).reshape(-1, 1)
    s, i = oracle.sort_dim(x, 0)
    assert list(i.ravel()) == [9, 1, 2, 5, 6, 3, 7, 4, 8]                     # equal keys (incl. +0 / -0) keep their order; NaNs last
    assert np.array_equal(np.signbit(s.ravel()[1:5]), [False, True, True, False])
    s, i = oracle.sort_dim(x, 0, descend=True)
    assert list(i.ravel()) == [4, 8, 3, 7, 1, 2, 5, 6, 9]                     # NaNs first, ties still in input order
    s, i = oracle.sort_dim(np.array([2.0, -2.0, 1.0, -1.0, -2.0]).reshape(-1, 1), 0, by_abs=True)
    assert list(s.ravel()) == [-1.0, 1.0, -2.0, -2.0, 2.0] and list(i.ravel()) == [4, 3, 2, 5, 1]
    s, i = oracle.sort_dim(np.array([2.0, -2.0, 1.0, -1.0, -2.0]).reshape(-1, 1), 0, descend=True, by_abs=True)
    assert list(s.ravel()) == [2.0, -2.0, -2.0, 1.0, -1.0] and list(i.ravel()) == [1, 2, 5, 3, 4]


def test_median_all_hook_and_host_path_differ():
    x = np.array([[1.0, 2.0, 30.0], [4.0, 50.0, 6.0], [7.0, 8.0, 9.0]])
    assert oracle.median_all(x) == 7.0 and oracle.median_all(x, successive=True) == 8.0


def test_random_distribution_restatements_follow_their_definitions():
    """generate_uniform_scaled / generate_exponential / generate_normal_scaled / random_integer_range written out over the oracle's own
    (pinned) uniform and normal streams - the reference's tests define the expected sequences the same way (random.rs:574-606)."""
    s0 = oracle.rng_default_seed()
    u, s1 = oracle.rng_uniform(s0, 9)
    got, state = oracle.rng_unifrnd(s0, 2.0, 5.0, 9)
    assert np.array_equal(got, 2.0 + (5.0 - 2.0) * u) and state == s1
    got, state = oracle.rng_exponential(s0, 2.5, 9)
    assert np.array_equal(got, -2.5 * np.log(np.maximum(u, 2.2250738585072014e-308))) and state == s1
    z, s2 = oracle.rng_normal(s0, 9)
    got, state = oracle.rng_normrnd(s0, 1.0, 3.0, 9)
    assert np.array_equal(got, 1.0 + 3.0 * z) and state == s2
    got, state = oracle.rng_integer_range(s0, -3, 3, 9)
    assert np.array_equal(got, -3 + np.minimum(np.floor(u * 7.0), 6.0)) and state == s1
    got, state = oracle.rng_integer_range(s0, 4, 4, 9)
    assert np.array_equal(got, np.full(9, 4.0)) and state == s0
    assert oracle.rng_integer_range(s0, 2, 1, 3) is None and oracle.rng_integer_range(s0, 0, 2**53, 3) is None


def test_find_restatement():
    """find.rs tests: find([0 2 0 4]) = [2 4]; 'last' walks from the end (descending); NaN is nonzero; limits clamp."""
    l, r, c, v = oracle.find(np.array([[0.0, 2.0, 0.0, 4.0]]))
    assert list(l.ravel()) == [2, 4] and list(r.ravel()) == [1, 1] and list(c.ravel()) == [2, 4] and list(v.ravel()) == [2, 4]
    x = np.array([[0.0, 3.0, np.nan], [5.0, 0.0, -0.0]])
    l, r, c, v = oracle.find(x)
    assert list(l.ravel()) == [2, 3, 5] and list(r.ravel()) == [2, 1, 1] and list(c.ravel()) == [1, 2, 3] and np.isnan(v.ravel()[2])
    assert list(oracle.find(x, 2)[0].ravel()) == [2, 3] and list(oracle.find(x, None, True)[0].ravel()) == [5]
    assert list(oracle.find(x, 2, True)[0].ravel()) == [5, 3] and oracle.find(x, 0)[0].shape == (0, 1) and list(oracle.find(x, 99)[0].ravel()) == [2, 3, 5]
