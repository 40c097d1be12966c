"""GPU parity of fft_dim / ifft_dim / fft_extract_real / complex_from_real(_imag) (include/rmhip.h, fft.hip) against the oracle's DFT by
direct evaluation in long double, and - at sizes where that takes too long - against numpy's pocketfft.

Tolerance (the reference transforms with rustfft, so parity is by tolerance; its own tests allow 1e-12 / 1e-10 absolute on vectors of
magnitude ~10): every point within  C * eps * log2(work length) * ||line||_2  of the exact transform, C = 4 for power-of-two lengths
and 8 for the chirp-convolution path (work length = the padded convolution length), both before the 1 / n of the inverse."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = json.loads((Path(__file__).parent / "golden" / "fft_kats.json").read_text())
EPS = np.finfo(np.float64).eps


def cpx(pairs):
    return np.array([complex(a, b) for a, b in pairs])


def upload_any(prov, x):
    up = lambda v: prov.upload(np.ascontiguousarray(v).ravel(order="F"), x.shape)   # (keeps a rank-1 shape rank-1)
    if np.iscomplexobj(x):
        return prov.complex_from_real_imag(up(x.real), up(x.imag))
    return up(x)


def bound(x, n, dim, inverse):
    """Per-line error bound, broadcastable against the result."""
    xs = x.reshape(x.shape + (1,) * max(0, dim + 1 - x.ndim))
    pow2 = n & (n - 1) == 0
    work = n if pow2 else 1 << math.ceil(math.log2(2 * n - 1))
    c = 4.0 if pow2 else 8.0
    norm = np.sqrt(np.sum(np.abs(np.take(xs, range(min(n, xs.shape[dim])), axis=dim)) ** 2, axis=dim, keepdims=True))
    return c * EPS * max(1.0, math.log2(work)) * norm / (n if inverse else 1) + 1e-300


def check(prov, x, length, dim, inverse, want):
    h = upload_any(prov, x)
    out = (prov.ifft_dim if inverse else prov.fft_dim)(h, length, dim)
    assert prov.is_complex(out) and list(out.shape) == list(want.shape)
    got = prov.download(out).reshape(want.shape, order="F")
    n = want.shape[dim]
    if want.size:
        err = np.abs(got - want)
        lim = bound(np.asarray(x), n, dim, inverse)
        assert np.all(err <= lim), (x.shape, length, dim, inverse, float(np.max(err / lim)))
    prov.free(out)
    prov.free(h)


def test_reference_kats(prov):
    for k in K["fft"]:
        out = prov.fft_dim(prov.upload(np.array(k["x"], dtype=np.float64), k["shape"]), k["len"], k["dim"])
        assert list(out.shape) == k["oshape"] and prov.is_complex(out)
        assert np.max(np.abs(prov.download(out) - cpx(k["out"]))) <= k["tol"]
    for k in K["fft_first"]:
        out = prov.fft_dim(prov.upload(np.array(k["x"], dtype=np.float64), k["shape"]), k["len"], k["dim"])
        assert list(out.shape) == k["oshape"] and abs(prov.download(out)[0] - complex(*k["first"])) <= k["tol"]


@pytest.mark.parametrize("shape,dim,length", [
    ((2,), 0, None), ((4,), 0, None), ((8,), 0, None), ((16, 3), 0, None), ((64, 5), 0, None), ((512,), 0, None), ((4096, 2), 0, None), ((1024, 3), 0, 4096),
    ((8192,), 0, None), ((8192, 3), 0, None), ((16384,), 0, None), ((3000,), 0, 8192), ((5, 64), 1, None), ((300, 256), 1, None), ((7, 512), 1, None),
    ((33, 1024, 2), 1, None), ((3, 4, 16), 2, None), ((3, 4, 5), 1, 8), ((3, 4, 5), 2, 4), ((100, 6), 0, 64), ((6, 2), 3, None), ((6, 2), 2, 4),
    ((1, 1), 0, None), ((1, 1), 0, 8), ((2, 9000), 1, 8192),
], ids=str)
def test_power_of_two_lengths(prov, oracle, shape, dim, length):
    rng = np.random.default_rng(abs(hash((shape, dim, length))) % 2**32)
    for x in (rng.standard_normal(shape), rng.standard_normal(shape) + 1j * rng.standard_normal(shape)):
        for inverse in (False, True):
            n = length if length is not None else (shape[dim] if dim < len(shape) else 1)
            lines = int(np.prod(shape)) // max(1, shape[dim] if dim < len(shape) else 1)
            if n * min(n, 4096) * lines <= 5e7:
                want = oracle.fft_dim(x, length, dim, inverse)
            else:
                xs = x.reshape(x.shape + (1,) * max(0, dim + 1 - x.ndim))
                want = (np.fft.ifft if inverse else np.fft.fft)(xs, n=length, axis=dim)
            check(prov, x, length, dim, inverse, want)


@pytest.mark.parametrize("shape,dim,length", [
    ((3,), 0, None), ((5,), 0, None), ((7, 4), 0, None), ((4,), 0, 7), ((2, 3, 3), 1, 7), ((100,), 0, None), ((1000, 3), 0, None), ((4097,), 0, None),
    ((6, 100), 1, None), ((5, 300, 2), 1, None), ((44100,), 0, None), ((1000,), 0, 1500), ((1000,), 0, 999), ((17, 33), 1, 31), ((12, 3000), 1, None),
], ids=str)
def test_other_lengths(prov, oracle, shape, dim, length):
    rng = np.random.default_rng(abs(hash((shape, dim, length))) % 2**32)
    for x in (rng.standard_normal(shape), rng.standard_normal(shape) + 1j * rng.standard_normal(shape)):
        for inverse in (False, True):
            n = length if length is not None else shape[dim]
            lines = int(np.prod(shape)) // shape[dim]
            if n * n * lines <= 5e7:
                want = oracle.fft_dim(x, length, dim, inverse)
            else:
                want = (np.fft.ifft if inverse else np.fft.fft)(x, n=length, axis=dim)
            check(prov, x, length, dim, inverse, want)


def test_round_trip_linearity_and_a_long_vector(prov):
    rng = np.random.default_rng(77)
    for n in (1 << 19, 1 << 20, 1 << 22, 1 << 25, 1000003 // 7, 300007):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        hx, hy = prov.upload(x), prov.upload(y)
        fx, fy = prov.fft_dim(hx, None, 0), prov.fft_dim(hy, None, 0)
        want = np.fft.fft(x)
        got = prov.download(fx)
        c = 4.0 if n & (n - 1) == 0 else 8.0
        lim = c * EPS * math.log2(2 * n) * np.linalg.norm(x)
        assert np.max(np.abs(got - want)) <= 2 * lim                               # (numpy's own error is inside the bound as well)
        fs = prov.fft_dim(prov.elem_add(hx, hy), None, 0)
        assert np.max(np.abs(prov.download(fs) - (got + prov.download(fy)))) <= 4 * lim
        back = prov.ifft_dim(fx, None, 0)
        assert prov.is_complex(back)
        rt = prov.download(back)
        assert np.max(np.abs(rt.real - x)) <= 4 * lim / math.sqrt(n) and np.max(np.abs(rt.imag)) <= 4 * lim / math.sqrt(n)
        re = prov.fft_extract_real(back)
        assert not prov.is_complex(re) and np.array_equal(prov.download(re), rt.real)
        for h in (hx, hy, fx, fy, fs, back, re):
            prov.free(h)


def test_complex_storage_plumbing(prov):
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((5, 7)), rng.standard_normal((5, 7))
    ha, hb = prov.upload(a), prov.upload(b)
    z = prov.complex_from_real_imag(ha, hb)
    assert prov.is_complex(z) and list(z.shape) == [5, 7] and prov.logical_isreal(z) is False and prov.logical_isreal(ha) is True
    assert np.array_equal(prov.download(z).reshape((5, 7), order="F"), a + 1j * b)
    z0 = prov.complex_from_real(ha)
    assert np.array_equal(prov.download(z0), a.ravel(order="F") + 0j)
    zs = prov.complex_from_real_imag(ha, prov.upload(np.array([[2.5]])))
    assert np.array_equal(prov.download(zs), a.ravel(order="F") + 2.5j)
    zr = prov.complex_from_real_imag(prov.upload(np.array([[-1.0]])), hb)
    assert np.array_equal(prov.download(zr), -1.0 + 1j * b.ravel(order="F"))
    with pytest.raises(Exception):
        prov.complex_from_real_imag(ha, prov.upload(np.zeros((7, 5))))
    assert np.array_equal(prov.download(prov.fft_extract_real(z)), a.ravel(order="F"))
    r2 = prov.fft_extract_real(ha)                                                # a real input: a copy under a new id
    assert r2.buffer_id != ha.buffer_id and np.array_equal(prov.download(r2), a.ravel(order="F"))
    # real-valued entry points refuse complex storage (the caller gathers, as for any Err)
    for f in (lambda: prov.unary_sin(z), lambda: prov.elem_add(z, ha), lambda: prov.reduce_sum(z), lambda: prov.transpose(z), lambda: prov.matmul(z, z)):
        with pytest.raises(Exception):
            f()
    # empty and zero-length transforms
    e = prov.fft_dim(ha, 0, 1)
    assert list(e.shape) == [5, 0] and prov.is_complex(e) and prov.download(e).size == 0
    e2 = prov.fft_dim(prov.upload(np.zeros((0, 3))), 4, 0)
    assert list(e2.shape) == [4, 3] and not prov.download(e2).any()


def test_baseline_size_matrix(prov):
    """BASELINE's 8192 x 8192 operand: every column and every row transformed, compared with numpy's pocketfft over the whole result,
    then Parseval's identity and the round trip on the device results."""
    n = 8192
    h = prov.fill_uniform(5, -1.0, 1.0, (n, n))
    x = prov.download_matrix(h)
    for dim in (0, 1):
        f = prov.fft_dim(h, None, dim)
        got = prov.download(f).reshape((n, n), order="F")
        want = np.fft.fft(x, axis=dim)
        lim = 2 * 4.0 * EPS * 13 * np.sqrt(np.sum(x * x, axis=dim, keepdims=True))
        assert np.all(np.abs(got - want) <= lim)
        assert abs(np.sum(np.abs(got) ** 2) / n - np.sum(x * x)) <= 1e-12 * np.sum(x * x)
        back = prov.fft_extract_real(prov.ifft_dim(f, None, dim))
        assert np.max(np.abs(prov.download_matrix(back) - x)) <= 64 * EPS
        del got, want


@pytest.mark.parametrize("shape,dim,length", [((16,), 0, None), ((15,), 0, None), ((4096, 3), 0, None), ((7, 1000), 1, None), ((100,), 0, 128), ((300,), 0, 100),
                                              ((8192, 4), 0, None)], ids=str)
def test_signal_hilbert(prov, oracle, shape, dim, length):
    """hilbert = fft -> one-sided mask -> ifft: two transforms' worth of the transform tolerance, checked against the oracle (direct DFTs)
    or scipy for long lines; the real part returns the (padded / truncated) input."""
    from scipy.signal import hilbert
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape)
    n = length if length is not None else shape[dim]
    want = oracle.hilbert(x, length, dim) if n <= 1024 else hilbert(x, N=length, axis=dim)
    h = prov.signal_hilbert(prov.upload(x.ravel(order="F"), shape), length, dim)
    assert prov.is_complex(h) and list(h.shape) == list(want.shape)
    got = prov.download(h).reshape(want.shape, order="F")
    lim = 2 * bound(np.asarray(x), n, dim, False) * 2.0 / np.sqrt(n)                 # forward error through the 1 / n of the inverse, mask factor 2
    assert np.all(np.abs(got - want) <= lim + 1e-300), float(np.max(np.abs(got - want) / lim))
    with pytest.raises(Exception):
        prov.signal_hilbert(prov.upload(x.ravel(order="F"), shape), 0, dim)
    with pytest.raises(Exception):
        prov.signal_hilbert(prov.upload(x.ravel(order="F"), shape), None, len(shape))
