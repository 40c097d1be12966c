"""GPU parity of unique / ismember for elements (include/rmhip.h, order_ops.hip): integer work on sorted (key, position) pairs, host
results - bit-exact against the oracle's restatement of the CPU's hash-map forms."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = json.loads((Path(__file__).parent / "golden" / "set_kats.json").read_text())


def arr(v):
    return np.array([np.nan if e == "nan" else e for e in v], dtype=np.float64)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_reference_kats(prov):
    for k in K["unique"]:
        values, ia, ic = prov.unique(prov.upload(arr(k["x"]).reshape(-1, 1)), order=k["order"], occurrence=k["occ"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True), k
        if "ia" in k:
            assert list(ia.ravel()) == k["ia"] and list(ic.ravel()) == k["ic"], k
    for k in K["sort_rows"]:
        r = prov.sort_rows(prov.upload(arr(k["a"]), k["shape"]), [tuple(c) for c in k["columns"]])
        assert np.array_equal(r.values.ravel(order="F"), k["values"]), k
        if "indices" in k:
            assert list(r.indices.ravel()) == k["indices"], k
    for k in K["union"]:
        values, ia, ib = prov.union(prov.upload(arr(k["a"]).reshape(-1, 1)), prov.upload(arr(k["b"]).reshape(-1, 1)), order=k["order"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True) and list(ia.ravel()) == k["ia"] and list(ib.ravel()) == k["ib"], k
    for k in K["setdiff"]:
        values, ia = prov.setdiff(prov.upload(arr(k["a"]).reshape(-1, 1)), prov.upload(arr(k["b"]).reshape(-1, 1)), order=k["order"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True) and list(ia.ravel()) == k["ia"], k
    for k in K["ismember"]:
        mask, loc = prov.ismember(prov.upload(arr(k["a"]).reshape(1, -1)), prov.upload(arr(k["b"]).reshape(1, -1)))
        assert list(mask.ravel()) == k["mask"] and list(loc.ravel()) == k["loc"], k


@pytest.mark.parametrize("shape,span", [((1, 1), 3), ((4, 1), 2), ((7, 9), 5), ((1000,), 50), ((2049, 3), 400), ((5000, 13), 100000), ((70000,), 9)], ids=str)
def test_unique(prov, oracle, shape, span):
    rng = np.random.default_rng(sum(shape) + span)
    x = rng.integers(-span, span + 1, size=shape).astype(np.float64)
    flat = x.reshape(-1)
    flat[rng.integers(0, flat.size, size=max(1, flat.size // 11))] = np.nan
    flat[rng.integers(0, flat.size, size=max(1, flat.size // 13))] = -0.0
    flat[rng.integers(0, flat.size, size=max(1, flat.size // 17))] = np.inf
    h = prov.upload(x.ravel(order="F"), shape)
    for order in ("sorted", "stable"):
        for occ in ("first", "last"):
            got, want = prov.unique(h, order=order, occurrence=occ), oracle.unique(x, order, occ)
            for g, w in zip(got, want):
                assert same_bits(g, w), (order, occ)
    with pytest.raises(Exception):
        prov.unique(h, rows=True)


def test_unique_edges(prov, oracle):
    e = prov.unique(prov.upload(np.zeros((0, 3))))
    assert e[0].shape == (0, 1) and e[1].shape == (0, 1) and e[2].shape == (0, 1)
    z = np.array([-0.0, 0.0, np.nan, 1.0, np.nan, -0.0])
    got = prov.unique(prov.upload(z.reshape(1, -1)))
    assert np.signbit(got[0][0, 0]) and list(got[1].ravel()) == [1, 4, 3] and list(got[2].ravel()) == [1, 1, 3, 2, 3, 1]
    c = prov.unique(prov.fill((300, 300), 7.0), order="stable", occurrence="last")
    assert c[0].tolist() == [[7.0]] and c[1].tolist() == [[90000.0]] and np.all(c[2] == 1.0)


@pytest.mark.parametrize("sa,nb", [((1, 1), 1), ((4, 5), 11), ((300, 7), 50), ((2049,), 3000), ((50000,), 70000), ((8, 8), 0)], ids=str)
def test_ismember(prov, oracle, sa, nb):
    rng = np.random.default_rng(sum(sa) + nb)
    a = rng.integers(-40, 41, size=sa).astype(np.float64)
    b = rng.integers(-20, 60, size=nb).astype(np.float64)
    for v in (a.reshape(-1), b):
        if v.size > 4:
            v[rng.integers(0, v.size, size=2)] = np.nan
            v[rng.integers(0, v.size)] = -0.0
    got = prov.ismember(prov.upload(a.ravel(order="F"), sa), prov.upload(b.reshape(-1, 1)))
    want = oracle.ismember(a, b)
    assert got[0].dtype == np.uint8 and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_unique_at_baseline_size(prov):
    """8192 x 8192 values drawn from 1000 integers: every value found, ia / ic consistent with the data (the definition, checked on the
    whole result with numpy)."""
    n = 8192
    h = prov.fill_uniform(21, 0.0, 1000.0, (n, n))
    hf = prov.unary_floor(h)
    x = prov.download(hf)
    values, ia, ic = prov.unique(hf)
    assert np.array_equal(values.ravel(), np.arange(1000.0))
    assert np.array_equal(x[ia.ravel().astype(np.int64) - 1], values.ravel())
    assert np.array_equal(values.ravel()[ic.ravel().astype(np.int64) - 1], x)
    first = np.full(1000, x.size, dtype=np.int64)
    np.minimum.at(first, x.astype(np.int64), np.arange(x.size))
    assert np.array_equal(ia.ravel(), first + 1.0)


@pytest.mark.parametrize("na,nb,span", [(1, 1, 2), (40, 25, 30), (3000, 5000, 800), (70000, 1000, 100000), (5, 0, 3), (0, 5, 3), (0, 0, 1)], ids=str)
def test_union_and_setdiff(prov, oracle, na, nb, span):
    rng = np.random.default_rng(na + 3 * nb)
    a, b = rng.integers(-span, span, size=na).astype(np.float64), rng.integers(-span // 2, 2 * span, size=nb).astype(np.float64)
    for v in (a, b):
        if v.size > 6:
            v[rng.integers(0, v.size, size=2)] = np.nan
            v[rng.integers(0, v.size)] = -0.0
    ha, hb = prov.upload(a.reshape(-1, 1)), prov.upload(b.reshape(1, -1))
    for order in ("sorted", "stable"):
        for got, want in ((prov.union(ha, hb, order=order), oracle.union(a, b, order)), (prov.setdiff(ha, hb, order=order), oracle.setdiff(a, b, order))):
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert same_bits(g, w), (order, g.shape, w.shape)
    with pytest.raises(Exception):
        prov.union(ha, hb, rows=True)


@pytest.mark.parametrize("shape", [(1, 3), (2, 2), (50, 3), (3000, 5), (70000, 2), (5, 0), (0, 4), (300, 1)], ids=str)
def test_sort_rows(prov, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    m = rng.integers(-3, 4, size=shape).astype(np.float64)
    if m.size > 20:
        flat = m.reshape(-1)
        flat[rng.integers(0, flat.size, size=max(2, flat.size // 9))] = np.nan
        flat[rng.integers(0, flat.size, size=max(2, flat.size // 11))] = -0.0
    h = prov.upload(m.ravel(order="F"), shape)
    cols = shape[1]
    specs = [[(c, "ascend") for c in range(cols)], [(c, "descend") for c in range(cols)], [(cols - 1, "descend"), (0, "ascend")], [(0, "ascend"), (7, "descend")], []]
    for columns in specs:
        for comparison in ("auto", "abs"):
            want_v, want_i = oracle.sort_rows(m, columns, comparison)
            r = prov.sort_rows(h, columns, comparison)
            assert list(r.indices.shape) == [shape[0], 1] and np.array_equal(r.indices, want_i), (columns, comparison)
            assert same_bits(r.values, want_v), (columns, comparison)
    with pytest.raises(Exception):
        prov.sort_rows(prov.upload(np.zeros((2, 2, 2))), [(0, "ascend")])
