"""The oracle's DFT (direct evaluation in long double, oracle.c `orc_dft_dim`) against the reference's own unit-test vectors
(tests/golden/fft_kats.json) and against numpy's pocketfft on every framing case: dimension, padding, truncation, inverse, complex
input."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle

K = json.loads((Path(__file__).parent / "golden" / "fft_kats.json").read_text())


def cpx(pairs):
    return np.array([complex(a, b) for a, b in pairs])


def test_reference_kats():
    for k in K["fft"]:
        y = oracle.fft_dim(np.array(k["x"], dtype=np.float64).reshape(k["shape"], order="F"), k["len"], k["dim"])
        assert list(y.shape) == k["oshape"]
        assert np.max(np.abs(y.ravel(order="F") - cpx(k["out"]))) <= k["tol"]
    for k in K["fft_first"]:
        y = oracle.fft_dim(np.array(k["x"], dtype=np.float64).reshape(k["shape"], order="F"), k["len"], k["dim"])
        assert list(y.shape) == k["oshape"] and abs(y.ravel()[0] - complex(*k["first"])) <= k["tol"]


@pytest.mark.parametrize("shape,dim,length", [((8,), 0, None), ((7,), 0, None), ((5, 6), 0, None), ((5, 6), 1, None), ((3, 4, 5), 1, 7), ((3, 4, 5), 2, 3),
                                              ((16, 3), 0, 32), ((6, 2), 3, None), ((6, 2), 2, 4), ((1, 1), 0, None), ((97,), 0, None)], ids=str)
def test_against_numpy(shape, dim, length):
    rng = np.random.default_rng(len(shape) * 100 + dim)
    for x in (rng.standard_normal(shape), rng.standard_normal(shape) + 1j * rng.standard_normal(shape)):
        xs = x.reshape(x.shape + (1,) * max(0, dim + 1 - x.ndim))
        for inverse in (False, True):
            want = (np.fft.ifft if inverse else np.fft.fft)(xs, n=length, axis=dim)
            got = oracle.fft_dim(x, length, dim, inverse)
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 1e-13 * max(1.0, np.max(np.abs(want)))


def test_round_trip_and_empty():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((12, 5))
    assert np.max(np.abs(oracle.fft_dim(oracle.fft_dim(x, None, 0), None, 0, True) - x)) <= 1e-15 * 12
    assert oracle.fft_dim(np.zeros((0, 3)), 4, 0).shape == (4, 3) and not oracle.fft_dim(np.zeros((0, 3)), 4, 0).any()
    assert oracle.fft_dim(x, 0, 1).shape == (12, 0)


def test_hilbert_against_scipy():
    from scipy.signal import hilbert
    rng = np.random.default_rng(9)
    for shape, dim, n in (((16,), 0, None), ((15,), 0, None), ((12, 3), 0, None), ((4, 10), 1, None), ((9,), 0, 16), ((20,), 0, 7)):
        x = rng.standard_normal(shape)
        want = hilbert(x, N=n, axis=dim)
        got = oracle.hilbert(x, n, dim)
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-13
