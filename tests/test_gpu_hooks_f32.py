"""The hooks added in round 4 on a precision-32 provider (f32 storage, ProviderPrecision::F32): operands are widened, the f64 kernels run,
results are rounded to f32 once - so every hook must equal the f64 provider's result on the f32-rounded operand, rounded to f32."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prov32(built):
    from runmat_amd import HipProvider

    p = HipProvider(int(os.environ.get("RMHIP_TEST_DEVICE", "0")), precision="F32")
    yield p
    p.close()


def f32r(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(order="F"), np.asarray(b, dtype=np.float64).ravel(order="F")
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_order_and_index_hooks_round_once(prov, prov32):
    rng = np.random.default_rng(21)
    x = f32r(rng.standard_normal((257, 33)))
    x[5, 3], x[7, 7] = np.nan, 0.0
    h64, h32 = prov.upload(x), prov32.upload(x)
    assert prov32.precision() == "F32"
    for dim in (0, 1):
        a, b = prov.cummin_scan(h64, dim), prov32.cummin_scan(h32, dim)
        assert same(prov32.download(b.values), f32r(prov.download(a.values))) and same(prov32.download(b.indices), prov.download(a.indices))
        a, b = prov.sort_dim(h64, dim, "descend", "abs"), prov32.sort_dim(h32, dim, "descend", "abs")
        assert same(b.values, a.values) and same(b.indices, a.indices)
        assert same(prov32.download(prov32.reduce_median_dim(h32, dim)), f32r(prov.download(prov.reduce_median_dim(h64, dim))))
        assert same(prov32.download(prov32.diff_dim(h32, 2, dim, True)), f32r(prov.download(prov.diff_dim(h64, 2, dim, True))))
        assert same(prov32.download(prov32.gradient_dim(h32, dim, 0.5)), f32r(prov.download(prov.gradient_dim(h64, dim, 0.5))))
        t64, t32 = prov.download(prov.cumtrapz_dim(h64, dim, 0.25)), prov32.download(prov32.cumtrapz_dim(h32, dim, 0.25))
        ok = ~np.isnan(np.asarray(t64))
        assert np.allclose(np.asarray(t32)[ok], f32r(t64)[ok], rtol=3e-7, atol=1e-6)
    a, b = prov.find(h64, 9, "last"), prov32.find(h32, 9, "last")
    for f in ("linear", "rows", "cols", "values"):
        assert same(prov32.download(getattr(b, f)), f32r(prov.download(getattr(a, f))))
    y = f32r(rng.standard_normal((6, 3)))
    assert same(prov32.download(prov32.kron(prov32.upload(y), prov32.upload(y.T))), f32r(prov.download(prov.kron(prov.upload(y), prov.upload(y.T)))))
    assert same(prov32.download(prov32.cross(prov32.upload(y), prov32.upload(y[::-1]))), f32r(prov.download(prov.cross(prov.upload(y), prov.upload(y[::-1])))))
    v = f32r(rng.standard_normal(7))
    assert same(prov32.download(prov32.diag_from_vector(prov32.upload(v.reshape(-1, 1)), -2)), np.diag(v, -2))
    assert same(prov32.download(prov32.round_digits(h32, 2)), f32r(prov.download(prov.round_digits(h64, 2))))
    assert same(prov32.download(prov32.unary_angle(h32)), f32r(prov.download(prov.unary_angle(h64))))
    s = x[:33, :33].copy()
    s[np.isnan(s)] = 0.0
    s = f32r(s + s.T)
    assert prov32.issymmetric(prov32.upload(s)) is True and prov32.issymmetric(h32) is False
    subs = [prov32.upload(np.array([[1.0], [3.0], [2.0]])), prov32.upload(np.array([[4.0], [1.0], [2.0]]))]
    lin = prov32.sub2ind((3, 4), (1, 3), subs, [False, False], 3, (3, 1))
    assert same(prov32.download(lin), [10.0, 3.0, 5.0])
    r, c = prov32.ind2sub((3, 4), (1, 3), lin, 12, 3, (3, 1))
    assert same(prov32.download(r), [1.0, 3.0, 2.0]) and same(prov32.download(c), [4.0, 1.0, 2.0])
    gx, gy = prov32.ndgrid([prov32.upload(v[:3].reshape(-1, 1)), prov32.upload(v[3:].reshape(-1, 1))], (3, 4), 2)
    assert same(prov32.download(gx), np.repeat(v[:3].reshape(-1, 1), 4, axis=1)) and same(prov32.download(gy), np.repeat(v[3:].reshape(1, -1), 3, axis=0))
    m32 = prov32.scatter_row(h32, 4, prov32.upload(f32r(np.arange(33.0)).reshape(1, -1)))
    want = x.copy()
    want[4, :] = np.arange(33.0)
    assert same(prov32.download(m32), want)
    a = f32r(rng.standard_normal((65, 65)) + 8.0 * np.eye(65))
    xi = np.asarray(prov32.download(prov32.inv(prov32.upload(a)))).reshape(65, 65, order="F")
    assert np.max(np.abs(a @ xi - np.eye(65))) < 5e-6
    prov32.set_rng_state(12345)
    prov.set_rng_state(12345)
    assert same(prov32.download(prov32.random_unifrnd(2.0, 5.0, (1001, 1))), f32r(prov.download(prov.random_unifrnd(2.0, 5.0, (1001, 1)))))
    assert same(prov32.download(prov32.random_integer_range(-9, 9, (1001, 1))), prov.download(prov.random_integer_range(-9, 9, (1001, 1))))


def test_transforms_round_once(prov, prov32):
    """fft_dim / ifft_dim / complex constructors on a precision-32 provider: f32 operands widened, the f64 transform, every real and
    imaginary part rounded to f32 once (complex storage itself stays 2 x f64)."""
    rng = np.random.default_rng(21)
    for shape, dim, length in (((64, 5), 0, None), ((6, 100), 1, None), ((1000,), 0, 1024), ((9, 8192), 1, None)):
        x = f32r(rng.standard_normal(shape))
        up = lambda p: p.upload(x.ravel(order="F"), shape)
        f64, f32 = prov.fft_dim(up(prov), length, dim), prov32.fft_dim(up(prov32), length, dim)
        assert prov32.is_complex(f32) and list(f32.shape) == list(f64.shape)
        want = prov.download(f64)
        got = prov32.download(f32)
        assert np.array_equal(got.real, f32r(want.real)) and np.array_equal(got.imag, f32r(want.imag))
        back = prov32.ifft_dim(f32, None, dim)
        b = prov32.download(back)
        assert np.array_equal(b.real, f32r(b.real)) and np.array_equal(b.imag, f32r(b.imag))
        re = prov32.fft_extract_real(back)
        assert prov32.is_complex(re) is False and np.array_equal(prov32.download(re), b.real)
    a = rng.standard_normal((4, 3))
    z = prov32.complex_from_real_imag(prov32.upload(a), prov32.upload(2 * a))
    assert np.array_equal(prov32.download(z), f32r(a).ravel(order="F") + 2j * f32r(a).ravel(order="F"))


def test_filters_and_polynomials_round_once(prov, prov32):
    rng = np.random.default_rng(5)
    img, ker = f32r(rng.standard_normal((150, 70, 2))), f32r(rng.standard_normal((5, 4)))
    up = lambda p, x: p.upload(x.ravel(order="F"), x.shape)
    for padding in ("replicate", 1.5):
        a, b = prov.imfilter(up(prov, img), up(prov, ker), padding, "full"), prov32.imfilter(up(prov32, img), up(prov32, ker), padding, "full")
        assert list(a.shape) == list(b.shape) and same(prov32.download(b), f32r(prov.download(a)))
    a, b = prov.conv2d(up(prov, img[:, :, 0]), up(prov, ker), "same"), prov32.conv2d(up(prov32, img[:, :, 0]), up(prov32, ker), "same")
    assert same(prov32.download(b), f32r(prov.download(a)))
    p, q = f32r(rng.standard_normal((1, 9))), f32r(rng.standard_normal((4, 1)))
    p[0, 0] = 0.0
    a, b = prov.polyder_product(up(prov, p), up(prov, q)), prov32.polyder_product(up(prov32, p), up(prov32, q))
    assert list(a.shape) == list(b.shape) and same(prov32.download(b), f32r(prov.download(a)))
    (an, ad), (bn, bd) = prov.polyder_quotient(up(prov, p), up(prov, q)), prov32.polyder_quotient(up(prov32, p), up(prov32, q))
    assert list(ad.shape) == list(bd.shape) == [7, 1] and same(prov32.download(bn), f32r(prov.download(an))) and same(prov32.download(bd), f32r(prov.download(ad)))
    a, b = prov.polyint(up(prov, p), 0.1), prov32.polyint(up(prov32, p), 0.1)
    assert list(b.shape) == [1, 10] and same(prov32.download(b), f32r(prov.download(a)))
    assert same(prov32.download(prov32.polyder_single(up(prov32, p))), f32r(prov.download(prov.polyder_single(up(prov, p)))))
