"""The oracle's restatements of unique / ismember (elements) against the reference's own unit-test vectors (tests/golden/set_kats.json)
and against numpy's set routines where they agree on the definition."""
import json
from pathlib import Path

import numpy as np

from oracle import oracle

K = json.loads((Path(__file__).parent / "golden" / "set_kats.json").read_text())


def arr(v):
    return np.array([np.nan if e == "nan" else e for e in v], dtype=np.float64)


def test_reference_kats():
    for k in K["unique"]:
        values, ia, ic = oracle.unique(arr(k["x"]), k["order"], k["occ"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True), k
        if "ia" in k:
            assert np.array_equal(ia.ravel(), k["ia"]) and np.array_equal(ic.ravel(), k["ic"]), k
    for k in K["sort_rows"]:
        values, idx = oracle.sort_rows(arr(k["a"]).reshape(k["shape"], order="F"), [tuple(c) for c in k["columns"]])
        assert np.array_equal(values.ravel(order="F"), k["values"]), k
        if "indices" in k:
            assert list(idx.ravel()) == k["indices"], k
    for k in K["union"]:
        values, ia, ib = oracle.union(arr(k["a"]), arr(k["b"]), k["order"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True) and list(ia.ravel()) == k["ia"] and list(ib.ravel()) == k["ib"], k
    for k in K["setdiff"]:
        values, ia = oracle.setdiff(arr(k["a"]), arr(k["b"]), k["order"])
        assert np.array_equal(values.ravel(), arr(k["values"]), equal_nan=True) and list(ia.ravel()) == k["ia"], k
    for k in K["ismember"]:
        mask, loc = oracle.ismember(arr(k["a"]), arr(k["b"]))
        assert list(mask) == k["mask"] and list(loc) == k["loc"], k


def test_against_numpy():
    rng = np.random.default_rng(3)
    x = rng.integers(-5, 6, size=(7, 9)).astype(np.float64)
    values, ia, ic = oracle.unique(x)
    u, first, inv = np.unique(x.ravel(order="F"), return_index=True, return_inverse=True)
    assert np.array_equal(values.ravel(), u) and np.array_equal(ia.ravel(), first + 1) and np.array_equal(ic.ravel(), inv + 1)
    z = np.array([-0.0, 0.0, np.nan, 1.0, np.nan, -0.0])
    values, ia, ic = oracle.unique(z)
    assert values.shape == (3, 1) and np.signbit(values[0, 0]) and np.isnan(values[2, 0]) and list(ia.ravel()) == [1, 4, 3] and list(ic.ravel()) == [1, 1, 3, 2, 3, 1]
    assert list(oracle.unique(z, "stable", "last")[1].ravel()) == [6, 5, 4]
    a, b = rng.integers(0, 8, size=(4, 5)).astype(np.float64), rng.integers(3, 12, size=11).astype(np.float64)
    mask, loc = oracle.ismember(a, b)
    assert np.array_equal(mask.astype(bool), np.isin(a, b)) and mask.shape == a.shape
    for v, l in zip(a.ravel(), loc.ravel()):
        assert (l == 0 and v not in b) or b[int(l) - 1] == v and v not in b[:int(l) - 1]
    m2, l2 = oracle.ismember(np.array([np.nan, -0.0, 5.0]), np.array([1.0, 0.0, np.nan, np.nan]))
    assert list(m2) == [1, 1, 0] and list(l2) == [3, 2, 0]
    p, q = rng.integers(0, 30, size=40).astype(np.float64), rng.integers(10, 50, size=25).astype(np.float64)
    assert np.array_equal(oracle.union(p, q)[0].ravel(), np.union1d(p, q)) and np.array_equal(oracle.setdiff(p, q)[0].ravel(), np.setdiff1d(p, q))
    m = rng.integers(0, 4, size=(50, 3)).astype(np.float64)
    got, idx = oracle.sort_rows(m, [(0, "ascend"), (1, "ascend"), (2, "ascend")])
    assert np.array_equal(got, m[np.lexsort((m[:, 2], m[:, 1], m[:, 0]))]) and np.array_equal(m[idx.ravel().astype(int) - 1], got)
    e = oracle.unique(np.zeros((0, 3)))
    assert e[0].shape == (0, 1) and e[2].shape == (0, 1)
