"""GPU parity of the small construction / linear-algebra hooks (include/rmhip.h, misc_ops.hip): diag_from_vector(_sized), kron, cross,
gradient_dim(_with_coordinates), issymmetric - one or two rounded operations per element: bit-exact against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = json.loads((Path(__file__).parent / "golden" / "linear_hooks_kats.json").read_text())


def arr(v, shape):
    return np.array(v, dtype=np.float64).reshape(shape, order="F")


def bits_equal(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_reference_kats(prov):
    for k in K["kron"]:
        h = prov.kron(prov.upload(arr(k["a"], k["sa"])), prov.upload(arr(k["b"], k["sb"])))
        assert list(h.shape) == k["so"] and np.array_equal(prov.download_matrix(h).ravel(order="F"), k["out"])
    for k in K["cross"]:
        h = prov.cross(prov.upload(arr(k["a"], k["shape"])), prov.upload(arr(k["b"], k["shape"])), k["dim"])
        assert np.array_equal(prov.download_matrix(h).ravel(order="F"), k["out"])
    for k in K["gradient"]:
        hx = prov.upload(arr(k["x"], k["shape"]))
        h = prov.gradient_dim(hx, k["dim"], k["spacing"]) if k["coords"] is None else \
            prov.gradient_dim_with_coordinates(hx, k["dim"], prov.upload(np.array(k["coords"]).reshape(1, -1)))
        assert np.array_equal(prov.download_matrix(h).ravel(order="F"), np.array(k["out"])), k
    for k in K["issymmetric"]:
        assert prov.issymmetric(prov.upload(arr(k["a"], k["shape"])), "skew" if k["skew"] else "symmetric", k["tol"]) == k["out"]
    for k in K["ishermitian"]:
        assert prov.ishermitian(prov.upload(arr(k["a"], k["shape"])), "skew" if k["skew"] else "hermitian", k["tol"]) == k["out"], k
    for k in K["bandwidth"]:
        assert list(prov.bandwidth(prov.upload(arr(k["a"], k["shape"])))) == k["out"], k


@pytest.mark.parametrize("sa,sb", [((1, 1), (1, 1)), ((3, 4), (2, 5)), ((64, 3), (5, 70)), ((2, 3, 2), (3, 1, 2)), ((7,), (1, 9)), ((300, 200), (4, 3)),
                                   ((0, 2), (1, 2)), ((2, 2, 2, 2), (1, 3))], ids=str)
def test_kron(prov, oracle, sa, sb):
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(sa), rng.standard_normal(sb)
    h = prov.kron(prov.upload(a), prov.upload(b))
    want = oracle.kron(a.reshape(-1, 1) if a.ndim == 1 else a, b)
    assert tuple(h.shape) == want.shape and bits_equal(prov.download_matrix(h), want)


@pytest.mark.parametrize("shape", [(3, 1), (1, 3), (5, 3), (3, 5), (3, 3), (2, 3, 4), (1000, 3), (3, 100000), (7, 5, 3), (3, 3, 3)], ids=str)
def test_cross(prov, oracle, shape):
    rng = np.random.default_rng(6)
    a, b = rng.standard_normal(shape), rng.standard_normal(shape)
    ha, hb = prov.upload(a), prov.upload(b)
    assert bits_equal(prov.download_matrix(prov.cross(ha, hb)), oracle.cross(a, b))
    for d, ext in enumerate(shape):
        if ext == 3:
            assert bits_equal(prov.download_matrix(prov.cross(ha, hb, d + 1)), oracle.cross(a, b, d + 1))
        else:
            with pytest.raises(Exception):
                prov.cross(ha, hb, d + 1)                                        # cross.rs:453-458
    with pytest.raises(Exception):
        prov.cross(ha, hb, len(shape) + 1)
    with pytest.raises(Exception):
        prov.cross(ha, prov.upload(rng.standard_normal(shape[::-1] if shape[::-1] != shape else (3, 2))))


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (1, 2), (3, 1), (1, 9), (100, 7), (7, 100), (5, 6, 7), (100003, 2), (2, 100003)], ids=str)
def test_gradient(prov, oracle, shape):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(shape)
    hx = prov.upload(x)
    for dim in range(len(shape) + 1):
        for h in (1.0, 0.37):
            assert bits_equal(prov.download_matrix(prov.gradient_dim(hx, dim, h)), oracle.gradient(x, dim, h)), (shape, dim, h)
        ext = shape[dim] if dim < len(shape) else 1
        if ext >= 2:
            c = np.cumsum(rng.uniform(0.5, 1.5, ext))
            got = prov.gradient_dim_with_coordinates(hx, dim, prov.upload(c.reshape(-1, 1)))
            assert bits_equal(prov.download_matrix(got), oracle.gradient(x, dim, coords=c)), (shape, dim)
            with pytest.raises(Exception):
                prov.gradient_dim_with_coordinates(hx, dim, prov.upload(np.zeros((ext + 1, 1))))   # gradient.rs:1022-1029


def test_diag_from_vector(prov, oracle):
    rng = np.random.default_rng(8)
    for n in (1, 2, 5, 300, 4097):
        v = rng.standard_normal(n)
        for vec in (v.reshape(-1, 1), v.reshape(1, -1)):
            hv = prov.upload(vec)
            for off in (0, 1, -1, 7, -3):
                h = prov.diag_from_vector(hv, off)
                assert tuple(h.shape) == (n + abs(off),) * 2 and bits_equal(prov.download_matrix(h), oracle.diag_from_vector(v, off))
                for rows, cols in ((n, n), (3, n + 9), (n + 2, 2), (1, 1)):
                    hs = prov.diag_from_vector_sized(hv, off, rows, cols)
                    assert bits_equal(prov.download_matrix(hs), oracle.diag_from_vector(v, off, rows, cols)), (n, off, rows, cols)
    with pytest.raises(Exception):
        prov.diag_from_vector(prov.upload(np.zeros((2, 3))), 0)                   # a matrix: simple_provider.rs:3225-3228


def test_issymmetric(prov, oracle):
    rng = np.random.default_rng(9)
    for n in (1, 2, 17, 300, 2049):
        a = rng.standard_normal((n, n))
        s, k = a + a.T, a - a.T
        cases = [(s, False, 0.0), (s, True, 0.0), (k, True, 0.0), (k, False, 0.0), (a, False, 0.0), (a, False, 100.0)]
        if n > 1:
            s2 = s.copy()
            s2[0, n - 1] += 1e-9
            cases += [(s2, False, 0.0), (s2, False, 1e-8)]
            s3 = s.copy()
            s3[1, 0] = s3[0, 1] = np.inf
            cases += [(s3, False, 0.0)]
            s4 = s.copy()
            s4[1, 0] = np.nan
            cases += [(s4, False, 1e300)]
            k2 = k.copy()
            k2[n - 1, n - 1] = 1e-12
            cases += [(k2, True, 0.0), (k2, True, 1e-10)]
        for m, skew, tol in cases:
            assert prov.issymmetric(prov.upload(m), "skew" if skew else "symmetric", tol) == oracle.issymmetric(m, skew, tol), (n, skew, tol)
            assert prov.ishermitian(prov.upload(m), "skew" if skew else "hermitian", tol) == oracle.ishermitian(m, skew, tol), (n, skew, tol)
        d = s.copy()
        d[n // 2, n // 2] = np.nan                                                 # the NaN diagonal: the one case the two predicates split on
        hd = prov.upload(d)
        assert prov.issymmetric(hd) is True and prov.ishermitian(hd) is False and prov.ishermitian(hd, "skew", 1e300) is False
    assert prov.issymmetric(prov.upload(np.zeros((3, 4)))) is False
    with pytest.raises(Exception):
        prov.issymmetric(prov.upload(np.zeros((2, 2, 2))))


@pytest.mark.parametrize("shape", [(1, 1), (5, 5), (64, 64), (300, 200), (200, 300), (1, 4097), (4097, 1), (2049, 2049), (0, 3), (7,)], ids=str)
def test_bandwidth(prov, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    rows, cols = shape if len(shape) == 2 else (1, shape[0])
    up_ = lambda m: prov.upload(m.ravel(order="F"), shape)                      # keeps a rank-1 shape rank-1 (a row: bandwidth.rs:306)
    assert prov.bandwidth(up_(np.zeros(shape))) == (0, 0)
    for lo, up in ((0, 0), (1, 2), (rows // 2, cols // 3), (rows, cols)):
        m = np.tril(np.triu(rng.standard_normal((rows, cols)), -lo), up).reshape(shape)
        assert prov.bandwidth(up_(m)) == oracle.bandwidth(m)
    if rows * cols:
        m = np.zeros((rows, cols))
        m[rows - 1, 0] = np.nan                                                    # a NaN counts (bandwidth.rs:354)
        m[0, cols - 1] = -0.0                                                      # a negative zero does not
        assert prov.bandwidth(up_(m)) == oracle.bandwidth(m.reshape(shape)) == (rows - 1, 0)
    with pytest.raises(Exception):
        prov.bandwidth(prov.upload(np.zeros((2, 2, 2))))
    assert prov.bandwidth(prov.upload(np.ones((3, 3, 1)))) == (2, 2)


def test_full_size_properties(prov):
    """BASELINE's 8192 x 8192 operand through gradient (against numpy's formula element for element) and issymmetric."""
    n = 8192
    h = prov.fill_uniform(3, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for dim in (0, 1):
        g = prov.download_matrix(prov.gradient_dim(h, dim, 0.25))
        inner = (np.take(x, range(2, n), axis=dim) - np.take(x, range(0, n - 2), axis=dim)) / 0.5
        assert np.array_equal(np.take(g, range(1, n - 1), axis=dim), inner)
    assert prov.issymmetric(h) is False
    ht = prov.transpose(h)
    sym = prov.elem_add(h, ht)
    assert prov.issymmetric(sym) is True and prov.ishermitian(sym) is True
    assert prov.bandwidth(h) == (n - 1, n - 1)
    assert prov.bandwidth(prov.tril(prov.triu(h, -5), 77)) == (5, 77)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 300, 1153])
def test_inv_matches_the_oracle(prov, oracle, n):
    """inv = A \\ I on the LU path against the restated partial-pivoting inverse: forward error within cond * eps * n, residual
    ||A X - I|| at rounding level (the reference's own tests ask 1e-12 on a 2 x 2, inv.rs:365-383)."""
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n)) + np.sqrt(n) * np.eye(n)
    h = prov.inv(prov.upload(a))
    x = prov.download_matrix(h)
    assert x.shape == (n, n)
    want = oracle.inv(a)
    cond = np.linalg.cond(a)
    assert np.max(np.abs(x - want)) <= 50 * cond * np.finfo(float).eps * np.max(np.abs(want)), (n, cond)
    assert np.max(np.abs(a @ x - np.eye(n))) <= 1e-12 * n


def test_inv_shapes_and_errors(prov):
    assert np.array_equal(prov.download_matrix(prov.inv(prov.upload(np.array([[4.0]])))), [[0.25]])
    h = prov.inv(prov.upload(np.array([4.0, 0.0, 0.0, 2.0]).reshape(2, 2, 1)))
    assert tuple(h.shape) == (2, 2, 1) and np.array_equal(prov.download(h).ravel(), [0.25, 0.0, 0.0, 0.5])   # inv.rs:402-412
    assert tuple(prov.inv(prov.upload(np.zeros((0, 0)))).shape) == (0, 0)                                  # inv.rs:388-396
    for bad in (np.zeros((2, 3)), np.ones((2, 2, 2)), np.array([[1.0, 2.0], [2.0, 4.0]]), np.zeros((3, 1))):
        with pytest.raises(Exception):
            prov.inv(prov.upload(bad))
    tel = prov.telemetry_snapshot()
    assert any("inv:singular" in str(r) for r in tel.get("solve_fallbacks", [])), tel.get("solve_fallbacks")


def test_inv_residual_at_4096(prov):
    n = 4096
    a = np.random.default_rng(1).standard_normal((n, n)) + np.sqrt(n) * np.eye(n)
    x = prov.download_matrix(prov.inv(prov.upload(a)))
    assert np.max(np.abs(a @ x - np.eye(n))) <= 1e-12 * n


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (1, 2), (9, 1), (100, 7), (7, 100), (5, 6, 7), (100003, 2), (2, 100003), (4096, 300)], ids=str)
def test_trapz_and_cumtrapz(prov, oracle, shape):
    """Terms formed exactly as on the CPU, summed by the library's reduction / scan: within the summation-order tolerance of sum and
    cumsum (n eps sum|t|) of the oracle's in-order accumulation (simple_provider.rs:2421-2598)."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal(shape)
    hx = prov.upload(x)
    for dim in range(len(shape) + 1):
        ext = shape[dim] if dim < len(shape) else 1
        spacings = [(None, None), (0.37, 0.37)]
        if ext >= 2:
            c = np.cumsum(rng.uniform(0.5, 1.5, ext))
            spacings.append((("vector", prov.upload(c.reshape(-1, 1))), c))
            t = np.cumsum(rng.uniform(0.5, 1.5, shape), axis=dim) if dim < len(shape) else None
            if t is not None:
                spacings.append((("tensor", prov.upload(t)), t))
        sh = prov.upload(np.array([[0.37]]))
        spacings.append((("scalar_handle", sh), 0.37))
        for sp_dev, sp_host in spacings:
            for cumulative in (False, True):
                got = (prov.cumtrapz_dim if cumulative else prov.trapz_dim)(hx, dim, sp_dev)
                want = oracle.trapz(x, dim, sp_host, cumulative)
                g = prov.download(got)
                assert g.size == want.size, (shape, dim, cumulative, got.shape, want.shape)
                scale = max(ext, 2) * 2.3e-16 * (np.max(np.abs(x)) * 2.0 * (np.max(np.abs(np.diff(sp_host, axis=dim if np.ndim(sp_host) > 1 else 0))) if np.ndim(sp_host) else (abs(sp_host) if sp_host else 1.0))) * max(ext, 1)
                assert np.max(np.abs(np.asarray(g).ravel(order="F") - want.ravel(order="F")), initial=0.0) <= scale + 1e-300, (shape, dim, cumulative)
                prov.free(got)
    if shape[0] > 1:
        with pytest.raises(Exception):                                               # simple_provider.rs:2523-2526
            prov.trapz_dim(hx, 0, ("vector", prov.upload(np.zeros((shape[0] - 1, 1)))))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 100, 128, 129, 300, 1000, 2049])
def test_chol_matches_the_oracle(prov, oracle, n):
    """The blocked factorisation against the reference's Cholesky-Crout restatement: forward error within cond * eps (the two sum in
    different orders), R'R = A to rounding, the strict other triangle exactly zero; `lower` is the transpose."""
    rng = np.random.default_rng(n)
    b = rng.standard_normal((n, n))
    a = b @ b.T + n * np.eye(n)
    a = 0.5 * (a + a.T)
    h = prov.upload(a)
    res = prov.chol(h)
    r = prov.download_matrix(res.factor)
    want, info = oracle.chol(a)
    assert info == 0 and res.info == 0
    cond = np.linalg.cond(a)
    assert np.max(np.abs(r - want)) <= 20 * cond * np.finfo(float).eps * np.max(np.abs(want)), n
    assert np.max(np.abs(r.T @ r - a)) <= 50 * n * np.finfo(float).eps * np.max(np.abs(a))
    assert np.array_equal(np.tril(r, -1), np.zeros((n, n)))
    low = prov.download_matrix(prov.chol(h, True).factor)
    assert np.array_equal(low, r.T)


def test_chol_reference_vector_and_refusals(prov, oracle):
    a = np.array([[4.0, 12, -16], [12, 37, -43], [-16, -43, 98]])
    assert np.array_equal(prov.download_matrix(prov.chol(prov.upload(a)).factor), [[2, 6, -8], [0, 1, 5], [0, 0, 3]])   # chol.rs unit test
    assert tuple(prov.chol(prov.upload(np.zeros((0, 0)))).factor.shape) == (0, 0)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((200, 200))
    spd = b @ b.T + 200 * np.eye(200)
    bad_pd = spd.copy()
    bad_pd[150, 150] = -5.0                                  # not positive definite from column 151 on
    bad_sym = spd.copy()
    bad_sym[3, 170] += 1e-6                                  # pair (3, 170) differs by more than 1e-12 relative
    nan = spd.copy()
    nan[10, 10] = np.nan
    for m in (bad_pd, bad_sym, nan, np.zeros((3, 4)), np.array([[-1.0]])):
        if m.shape[0] == m.shape[1]:
            assert oracle.chol(m)[1] != 0                    # the host path reports a failure index for each of these
        with pytest.raises(Exception):
            prov.chol(prov.upload(m))


def test_chol_at_8192(prov):
    n = 8192
    h = prov.fill_uniform(4, -1.0, 1.0, (n, n))
    g = prov.syrk(h)                                         # A'A: symmetric by construction, SPD for a random square A... plus a shift
    shifted = prov.elem_add(g, prov.scalar_mul(prov.eye((n, n)), float(n)))
    a = prov.download_matrix(shifted)
    assert np.array_equal(a, a.T)
    res = prov.chol(shifted)
    r = prov.download_matrix(res.factor)
    assert res.info == 0 and np.array_equal(np.tril(r, -1), np.zeros((n, n)))
    x = np.random.default_rng(0).standard_normal((n, 4))
    assert np.max(np.abs(r.T @ (r @ x) - a @ x)) <= 1e-9 * np.max(np.abs(a @ x))


def test_norm(prov, oracle):
    """Every order the CPU computes without singular values, on vectors and matrices, against the restated loops (norm.rs:269-529):
    sums in another order (n eps), maxima / counts exact; the reference's unit-test values; NaN, infinities, huge and tiny magnitudes."""
    rng = np.random.default_rng(12)

    def val(h, order, p=2.0):
        return float(prov.download(prov.norm(h, order, p)).ravel()[0])

    kats = [(np.array([[3.0], [4.0]]), "two", 2.0, 5.0), (np.array([[2.0], [-7.0], [4.0]]), "inf", 2.0, 7.0), (np.array([[2.0], [-7.0], [4.0]]), "-inf", 2.0, 2.0),
            (np.array([[0.0], [0.0], [5.0], [0.0]]), "zero", 2.0, 1.0), (np.array([[2.0, 0.0], [0.0, 1.0]]), "fro", 2.0, np.sqrt(5.0)), (np.array([[2.0], [-3.0]]), "one", 2.0, 5.0)]
    for x, order, p, want in kats:                                                   # norm.rs:795-960
        assert abs(val(prov.upload(x), order, p) - want) < 1e-12
    assert abs(val(prov.upload(np.array([[1.0], [2.0], [3.0]])), "p", 1.5) - (1 + 2 ** 1.5 + 3 ** 1.5) ** (1 / 1.5)) < 1e-12
    for shape in ((1, 1), (7, 1), (1, 1000), (100003, 1), (300, 200), (2, 5000), (4096, 3)):
        x = rng.standard_normal(shape)
        h = prov.upload(x)
        is_matrix = min(shape) > 1
        for order in (("one", "inf", "fro") if is_matrix else ("one", "two", "inf", "-inf", "zero", "fro", "p")):
            want = oracle.norm(x, order, 3.0)
            got = val(h, order, 3.0)
            assert abs(got - want) <= 1e-13 * max(x.size, 16) * abs(want) + 1e-300, (shape, order, got, want)
        if is_matrix:
            sv = np.linalg.svd(x, compute_uv=False)                                  # spectral / nuclear norm: the Jacobi decomposition's singular values
            assert abs(val(h, "two") - sv.max()) <= 1e-12 * sv.max() and abs(val(h, "nuc") - sv.sum()) <= 1e-12 * sv.sum()
            for order in ("zero", "-inf", "p"):
                with pytest.raises(Exception):
                    prov.norm(h, order, 3.0)
        else:
            with pytest.raises(Exception):
                prov.norm(h, "nuc")
            with pytest.raises(Exception):
                prov.norm(h, "p", 0.5)
    big = np.array([1e200, -3e200, 2e-200, 0.0]).reshape(-1, 1)
    assert abs(val(prov.upload(big), "two") / oracle.norm(big, "two") - 1) < 1e-15       # squares overflow without the scale
    tiny = np.array([3e-200, 4e-200]).reshape(-1, 1)
    assert abs(val(prov.upload(tiny), "two") / 5e-200 - 1) < 1e-15
    for bad, order in ((np.array([[1.0], [np.nan]]), "inf"), (np.array([[1.0, np.nan], [2.0, 3.0]]), "one"), (np.array([[1.0, 2.0], [np.nan, 3.0]]), "fro")):
        assert np.isnan(val(prov.upload(bad), order))
    assert val(prov.upload(np.array([[1.0], [-np.inf]])), "two") == np.inf and val(prov.upload(np.array([[np.inf], [np.inf]])), "-inf") == 0.0
    assert val(prov.upload(np.zeros((0, 1))), "two") == 0.0 and val(prov.upload(np.zeros((5, 1))), "two") == 0.0
    with pytest.raises(Exception):
        prov.norm(prov.upload(np.zeros((2, 2, 2))), "fro")
    n = 8192
    h = prov.fill_uniform(6, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    assert abs(val(h, "fro") / np.linalg.norm(x) - 1) < 1e-13 and abs(val(h, "one") / np.abs(x).sum(axis=0).max() - 1) < 1e-13
    assert abs(val(h, "inf") / np.abs(x).sum(axis=1).max() - 1) < 1e-13


@pytest.mark.parametrize("shape", [(4, 3), (50, 6), (3000, 40), (100000, 8), (1, 3), (2, 2)], ids=str)
def test_corrcoef(prov, oracle, shape):
    """Sums of products: parity by tolerance, as for covariance - 1e-12 absolute on values in [-1, 1] (the reference's test allows 1e-10)."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape) * rng.uniform(0.1, 10.0, (1, shape[1])) + rng.uniform(-5, 5, (1, shape[1]))
    for norm in ("unbiased", "biased"):
        got = prov.download_matrix(prov.corrcoef(prov.upload(x), norm))
        want = oracle.corrcoef(x, norm) if shape[0] <= 5000 else np.corrcoef(x, rowvar=False)
        assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.all(np.abs(got[ok] - want[ok]) <= 1e-12) and (shape[0] < 2 or np.all(np.diag(got) == 1.0))
    if shape[0] >= 4:
        y = x.copy()
        y[:, 0] = 2.5                                                               # constant column
        y[1, -1] = np.inf                                                           # non-finite sample
        got, want = prov.download_matrix(prov.corrcoef(prov.upload(y))), oracle.corrcoef(y) if shape[0] <= 5000 else None
        assert np.isnan(got[0]).all() and np.isnan(got[:, 0]).all() and np.isnan(got[-1]).all()
        if want is not None:
            assert np.array_equal(np.isnan(got), np.isnan(want))
    with pytest.raises(Exception):
        prov.corrcoef(prov.upload(x), rows="pairwise")


def test_peaks(prov, oracle):
    for n in (0, 1, 2, 49, 1000):
        got, want = prov.download_matrix(prov.peaks(n)), oracle.peaks(n)
        assert got.shape == want.shape and (n == 0 or np.max(np.abs(got - want)) <= 2e-14)
    rng = np.random.default_rng(8)
    x, y = rng.uniform(-3, 3, (300, 7)), rng.uniform(-3, 3, (300, 7))
    assert np.max(np.abs(prov.download_matrix(prov.peaks_xy(prov.upload(x), prov.upload(y))) - oracle.peaks_xy(x, y))) <= 2e-14
    with pytest.raises(Exception):
        prov.peaks_xy(prov.upload(x), prov.upload(y.T))


def test_rank_cond_pinv(prov, oracle):
    """The CPU decomposes with nalgebra's SVD, the device by one-sided Jacobi: parity by tolerance - ranks equal (decisions away from the
    cutoff), cond to 1e-10 relative, pinv to 1e-11 * ||pinv|| (and the four Moore-Penrose identities to 1e-10)."""
    assert prov.download(prov.rank(prov.upload(np.array([1.0, 3.0, 2.0, 4.0]).reshape(2, 2, order="F"))))[0] == 2
    assert prov.download(prov.rank(prov.upload(np.diag([1.0, 1e-16]))))[0] == 1
    assert prov.download(prov.rank(prov.upload(np.diag([1.0, 1e-4])), 1e-3))[0] == 1 and prov.download(prov.rank(prov.upload(np.zeros((0, 0)))))[0] == 0
    assert np.allclose(prov.download_matrix(prov.pinv(prov.upload(np.diag([1.0, 1e-12])), 1e-6)), np.diag([1.0, 0.0]), atol=1e-15)
    rng = np.random.default_rng(23)
    for m, n, r in ((6, 4, 4), (4, 6, 3), (50, 50, 50), (200, 120, 60), (120, 200, 120), (7, 1, 1), (1, 7, 1)):
        a = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
        h = prov.upload(a)
        assert prov.download(prov.rank(h))[0] == oracle.rank(a) == r
        c_got, c_want = prov.download(prov.cond(h))[0], oracle.cond2(a)
        if r == min(m, n):
            assert abs(c_got - c_want) <= 1e-10 * c_want
        p_got, p_want = prov.download_matrix(prov.pinv(h)), oracle.pinv(a)
        assert p_got.shape == (n, m) and np.max(np.abs(p_got - p_want)) <= 1e-11 * np.max(np.abs(p_want))
        assert np.max(np.abs(a @ p_got @ a - a)) <= 1e-10 * np.max(np.abs(a)) and np.max(np.abs(p_got @ a @ p_got - p_got)) <= 1e-10 * np.max(np.abs(p_got))
    assert prov.download(prov.cond(prov.upload(np.diag([1.0, 0.0]))))[0] == np.inf and prov.download(prov.cond(prov.upload(np.zeros((0, 3)))))[0] == 0.0
    assert list(prov.pinv(prov.upload(np.zeros((3, 0)))).shape) == [0, 3]
    sq = rng.standard_normal((40, 40))
    sv = np.linalg.svd(sq, compute_uv=False)
    assert abs(prov.download(prov.rcond(prov.upload(sq)))[0] - sv.min() / sv.max()) <= 1e-10 * sv.min() / sv.max()      # rcond.rs:304-320
    assert prov.download(prov.rcond(prov.upload(np.zeros((3, 3)))))[0] == 0.0 and prov.download(prov.rcond(prov.upload(np.zeros((0, 0)))))[0] == np.inf
    with pytest.raises(Exception):
        prov.rcond(prov.upload(np.zeros((3, 4))))
    with pytest.raises(Exception):
        prov.cond(prov.upload(np.eye(3)), "fro")
    with pytest.raises(Exception):
        prov.pinv(prov.upload(np.eye(3)), -1.0)


def test_covariance_to_correlation(prov, oracle):
    # simple_provider.rs:8869-8900
    corr, sig = prov.covariance_to_correlation(prov.upload(np.array([[4.0, 2.0], [2.0, 9.0]])))
    assert np.allclose(prov.download_matrix(corr), [[1.0, 1.0 / 3.0], [1.0 / 3.0, 1.0]], atol=1e-15) and np.array_equal(prov.download(sig), [2.0, 3.0]) and list(sig.shape) == [2, 1]
    rng = np.random.default_rng(41)
    for n in (1, 3, 50, 700):
        x = rng.standard_normal((n + 5, n))
        cov = np.cov(x, rowvar=False).reshape(n, n)
        cov[0, 0] = 0.0 if n > 2 else cov[0, 0]
        if n > 2:
            cov[0, 1:] = cov[1:, 0] = 0.0                                             # a zero-variance variable: NaN row / column
            cov[1, 2] = cov[2, 1] = np.nan
        want_c, want_s = oracle.covariance_to_correlation(cov)
        c, s = prov.covariance_to_correlation(prov.upload(cov))
        assert bits_equal(prov.download_matrix(c), want_c) and bits_equal(prov.download_matrix(s), want_s)
    bad = {"symmetric": np.array([[1.0, 0.1], [0.2, 1.0]]), "nonnegative": np.array([[-1.0, 0.0], [0.0, 1.0]]), "finite": np.array([[1.0, np.inf], [np.inf, 1.0]]),
           "bounds": np.array([[1.0, 5.0], [5.0, 1.0]]), "square": np.zeros((2, 3))}
    for word, m in bad.items():
        with pytest.raises(Exception, match=word):
            prov.covariance_to_correlation(prov.upload(m))
        if word != "square":
            with pytest.raises(ValueError, match=word):
                oracle.covariance_to_correlation(m)
    e = prov.covariance_to_correlation(prov.upload(np.zeros((0, 0))))
    assert list(e[0].shape) == [0, 0] and list(e[1].shape) == [0, 1]
