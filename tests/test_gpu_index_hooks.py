"""GPU parity of the subscript / grid / slice-write hooks and the per-element forms of a real tensor (include/rmhip.h, index_ops.hip):
ndgrid, sub2ind, ind2sub, scatter_column / scatter_row, pow2_scale, round_digits, unary_real / imag / conj / angle, logical_isreal.
Integer and copy work, single rounded operations: bit-exact against the oracle's numpy restatements; error wording as on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits_equal(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return False
    g, w = got.copy(), want.copy()
    gn, wn = np.isnan(g), np.isnan(w)
    if not np.array_equal(gn, wn):
        return False
    g[gn], w[wn] = 0.0, 0.0
    return np.array_equal(g.view(np.uint64), w.view(np.uint64))


def strides_of(dims):
    s, cur = [], 1
    for d in dims:
        s.append(cur)
        cur *= d
    return s


def test_ndgrid(prov, oracle):
    rng = np.random.default_rng(1)
    for shape in ((3, 4), (5, 1), (1, 7), (4, 3, 5), (300, 200), (2, 3, 4, 5)):
        axes = [rng.standard_normal(n) for n in shape]
        hs = [prov.upload(a.reshape(-1, 1)) for a in axes]
        for count in range(1, len(shape) + 1):
            got = prov.ndgrid(hs, shape, count)
            want = oracle.ndgrid(axes, list(shape), count)
            assert len(got) == count
            for g, w in zip(got, want):
                assert tuple(g.shape) == tuple(shape) and bits_equal(np.asarray(prov.download(g)).ravel(order="F"), w.ravel(order="F"))
    # ndgrid.rs tests: [X, Y] = ndgrid(1:2, 1:3) -> X = [1 1 1; 2 2 2], Y = [1 2 3; 1 2 3]
    X, Y = prov.ndgrid([prov.upload(np.array([[1.0], [2.0]])), prov.upload(np.array([[1.0], [2.0], [3.0]]))], (2, 3), 2)
    assert np.array_equal(prov.download_matrix(X), [[1, 1, 1], [2, 2, 2]]) and np.array_equal(prov.download_matrix(Y), [[1, 2, 3], [1, 2, 3]])
    with pytest.raises(Exception):
        prov.ndgrid(hs[:1], (3, 4), 2)
    with pytest.raises(Exception):
        prov.ndgrid([prov.upload(np.zeros((5, 1)))], (3, 4), 1)


def test_sub2ind_and_ind2sub(prov, oracle):
    rng = np.random.default_rng(2)
    for dims, n in (((3, 4), 7), ((5,), 5), ((2, 3, 4), 1000), ((300, 200, 7), 100003)):
        st = strides_of(dims)
        subs = [rng.integers(1, d + 1, n).astype(np.float64) for d in dims]
        hs = [prov.upload(s.reshape(-1, 1)) for s in subs]
        got = prov.sub2ind(dims, st, hs, [False] * len(dims), n, (n, 1))
        want = oracle.sub2ind(dims, st, subs, [False] * len(dims), n, [n, 1])
        assert bits_equal(prov.download_matrix(got), want)
        total = int(np.prod(dims))
        back = prov.ind2sub(dims, st, got, total, n, (n, 1))
        for b, s in zip(back, subs):
            assert bits_equal(prov.download_matrix(b).ravel(), s)
        # a scalar subscript broadcasts (sub2ind.rs: scalar_mask)
        hsc = prov.upload(np.array([[float(dims[-1])]]))
        got = prov.sub2ind(dims, st, hs[:-1] + [hsc], [False] * (len(dims) - 1) + [True], n, (n, 1))
        want = oracle.sub2ind(dims, st, subs[:-1] + [np.array([float(dims[-1])])], [False] * (len(dims) - 1) + [True], n, [n, 1])
        assert bits_equal(prov.download_matrix(got), want)
    # sub2ind.rs tests: sub2ind([3 4], 2, 3) = 8; ind2sub([3 4], 8) = (2, 3)
    one = prov.sub2ind((3, 4), (1, 3), [prov.upload(np.array([[2.0]])), prov.upload(np.array([[3.0]]))], [False, False], 1, (1, 1))
    assert prov.download(one).ravel()[0] == 8.0
    r, c = prov.ind2sub((3, 4), (1, 3), prov.upload(np.array([[8.0]])), 12, 1, (1, 1))
    assert prov.download(r).ravel()[0] == 2.0 and prov.download(c).ravel()[0] == 3.0
    # refusals carry the CPU's wording for the FIRST offender in (element, dimension) order
    rows = np.array([1.0, 2.5, 9.0, np.nan])
    cols = np.array([1.0, 1.0, 1.0, 1.0])
    for bad_rows, needle in ((rows, "dimension 1 must be an integer"), (np.array([1.0, 4.0, 2.5]), "subscript 4 exceeds dimension 1 (size 3)"),
                             (np.array([np.inf, 0.0]), "dimension 1 must be finite"), (np.array([0.0, np.nan]), "subscript 0 exceeds dimension 1")):
        with pytest.raises(Exception) as e:
            prov.sub2ind((3, 4), (1, 3), [prov.upload(bad_rows.reshape(-1, 1)), prov.upload(cols[:bad_rows.size].reshape(-1, 1))], [False, False], bad_rows.size,
                         (bad_rows.size, 1))
        assert needle in str(e.value), str(e.value)
        assert oracle.sub2ind((3, 4), (1, 3), [bad_rows, cols], [False, False], bad_rows.size, [bad_rows.size, 1])[1] == 0
    for bad, needle in ((np.array([1.0, 13.0]), "Index exceeds number of array elements. Index must not exceed 12."), (np.array([0.0]), "Linear indices must be positive integers."),
                        (np.array([2.0, 1.5]), "Linear indices must be positive integers."), (np.array([np.nan]), "Linear indices must be positive integers.")):
        with pytest.raises(Exception) as e:
            prov.ind2sub((3, 4), (1, 3), prov.upload(bad.reshape(-1, 1)), 12, bad.size, (bad.size, 1))
        assert needle in str(e.value), str(e.value)
    assert prov.supports_ind2sub() is True


def test_scatter_column_and_row(prov, oracle):
    rng = np.random.default_rng(3)
    for rows, cols in ((1, 1), (3, 4), (257, 129), (4096, 300)):
        m = rng.standard_normal((rows, cols))
        hm = prov.upload(m)
        for idx in sorted({0, cols // 2, cols - 1}):
            v = rng.standard_normal(rows)
            got = prov.scatter_column(hm, idx, prov.upload(v.reshape(-1, 1)))
            assert bits_equal(prov.download_matrix(got), oracle.scatter_line(m, True, idx, v))
        for idx in sorted({0, rows // 2, rows - 1}):
            v = rng.standard_normal(cols)
            got = prov.scatter_row(hm, idx, prov.upload(v.reshape(1, -1)))
            assert bits_equal(prov.download_matrix(got), oracle.scatter_line(m, False, idx, v))
        assert bits_equal(prov.download_matrix(hm), m)                              # the operand is untouched
        with pytest.raises(Exception):
            prov.scatter_column(hm, cols, prov.upload(np.zeros((rows, 1))))
        with pytest.raises(Exception):
            prov.scatter_row(hm, 0, prov.upload(np.zeros((cols + 1, 1))))


def test_pow2_scale_round_digits_and_real_parts(prov, oracle):
    rng = np.random.default_rng(4)
    m = rng.standard_normal((300, 70))
    e_int = rng.integers(-1100, 1100, (300, 70)).astype(np.float64)
    hm = prov.upload(m)
    got = prov.download_matrix(prov.pow2_scale(hm, prov.upload(e_int)))
    with np.errstate(over="ignore", under="ignore"):
        assert bits_equal(got, oracle.pow2_scale(m, e_int))                          # integral exponents: exact powers, one rounding
        e_frac = rng.uniform(-20, 20, (300, 70))
        got = prov.download_matrix(prov.pow2_scale(hm, prov.upload(e_frac)))
        want = oracle.pow2_scale(m, e_frac)
    assert np.all(np.abs(got - want) <= 5e-16 * np.abs(want))                        # exp2 within 2 ulp of libm's
    with pytest.raises(Exception):
        prov.pow2_scale(hm, prov.upload(np.zeros((70, 300))))
    x = np.concatenate([rng.standard_normal(5000) * 10.0 ** rng.integers(-8, 8, 5000), [2.345, -2.5, 0.5, 1.5, -0.5, 0.49999999999999994, 1e300, -1e-300, np.inf, -np.inf,
                                                                                        np.nan, 0.0, -0.0, 4503599627370497.0]]).reshape(-1, 1)
    hx = prov.upload(x)
    for digits in (0, 1, 2, 3, 7, 15, 22, 300, 310, 400, -1, -2, -5, -300, -400):
        got = prov.download_matrix(prov.round_digits(hx, digits))
        assert bits_equal(got, oracle.round_decimals(x, digits)), digits
    with pytest.raises(Exception):
        prov.round_digits(hx, 3, True)
    assert bits_equal(prov.download_matrix(prov.unary_real(hx)), x) and bits_equal(prov.download_matrix(prov.unary_conj(hx)), x)
    assert bits_equal(prov.download_matrix(prov.unary_imag(hx)), np.zeros_like(x))
    assert bits_equal(prov.download_matrix(prov.unary_angle(hx)), oracle.angle_real(x))
    assert prov.logical_isreal(hx) is True


def test_full_size(prov):
    n = 8192
    h = prov.fill_uniform(9, -100.0, 100.0, (n, n))
    x = prov.download_matrix(h)
    assert np.array_equal(prov.download_matrix(prov.round_digits(h, 2)), np.sign(x) * np.floor(np.abs(x * 100.0) + 0.5) / 100.0)
    v = prov.fill_uniform(10, -1.0, 1.0, (n, 1))
    got = prov.download_matrix(prov.scatter_column(h, 17, v))
    x[:, 17] = prov.download_matrix(v).ravel()
    assert np.array_equal(got, x)
