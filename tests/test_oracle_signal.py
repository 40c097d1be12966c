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
