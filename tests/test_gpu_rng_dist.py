"""GPU parity of the scaled / transformed draws of the CPU-parity random stream (rmhip_random_unifrnd / _exponential / _normrnd /
_integer_range): the uniforms are the CPU generator's bit for bit, so unifrnd and the integers are bit-exact, the exponential and the
scaled normals differ from the CPU's libm by its own rounding only; the stream state advances exactly as the CPU's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prov32(built):
    import os
    from runmat_amd import HipProvider

    p = HipProvider(int(os.environ.get("RMHIP_TEST_DEVICE", "0")), precision="F32")
    yield p
    p.close()


def bits_equal(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 300001])
def test_unifrnd_and_integers_are_bit_exact(prov, oracle, n):
    seed = oracle.rng_default_seed()
    for a, b in ((2.0, 5.0), (-1e-3, 1e300), (0.1, 0.1), (3.0, -4.0)):
        prov.set_rng_state(seed)
        got = prov.download(prov.random_unifrnd(a, b, (n, 1)))
        want, state = oracle.rng_unifrnd(seed, a, b, n)
        assert bits_equal(got, want), (a, b)
        assert prov.get_rng_state() == state
    for lo, hi in ((1, 6), (-3, 3), (0, 2**53 - 1), (-2**62, -2**62 + 2**53 - 1), (2**62, 2**62 + 999), (5, 5)):
        prov.set_rng_state(seed)
        got = prov.download(prov.random_integer_range(lo, hi, (n, 1)))
        want, state = oracle.rng_integer_range(seed, lo, hi, n)
        assert bits_equal(got, want), (lo, hi)
        assert prov.get_rng_state() == state                                     # a one-value range consumes nothing
        assert got.min() >= lo and got.max() <= hi
    h = prov.upload(np.zeros((3, 4)))
    prov.set_rng_state(seed)
    like = prov.random_integer_like(h, 1, 10)
    assert like.shape == (3, 4) and bits_equal(prov.download(like), oracle.rng_integer_range(seed, 1, 10, 12)[0])
    prov.set_rng_state(seed)
    assert bits_equal(prov.download(prov.random_uniform_like(h)), oracle.rng_uniform(seed, 12)[0])
    prov.set_rng_state(seed)
    assert np.max(np.abs(prov.download(prov.random_normal_like(h)).ravel() - oracle.rng_normal(seed, 12)[0])) <= 8e-14


def test_integer_range_refusals(prov, oracle):
    for lo, hi in ((2, 1), (0, 2**53), (-2**63, 2**63 - 1)):
        assert oracle.rng_integer_range(1, lo, hi, 4) is None                      # simple_provider.rs:3689-3698
        with pytest.raises(Exception):
            prov.random_integer_range(lo, hi, (4, 1))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 300001])
def test_exponential_and_scaled_normals(prov, oracle, n):
    seed = oracle.rng_default_seed()
    for mu in (1.0, 2.5, 1e-300, -3.0):
        prov.set_rng_state(seed)
        got = prov.download(prov.random_exponential(mu, (n, 1))).ravel()
        want, state = oracle.rng_exponential(seed, mu, n)
        assert np.all(np.abs(got - want) <= 8e-16 * np.abs(want) + 5e-324), mu      # the device logarithm within 2 ulp of the truth, libm within 1, one multiply each
        assert prov.get_rng_state() == state
    for mu, sigma in ((0.0, 1.0), (10.0, 0.5), (-1e6, 3.0)):
        prov.set_rng_state(seed)
        got = prov.download(prov.random_normrnd(mu, sigma, (n, 1))).ravel()
        want, state = oracle.rng_normrnd(seed, mu, sigma, n)
        assert np.max(np.abs(got - want)) <= abs(sigma) * 8e-14 + abs(mu) * 2.3e-16, (mu, sigma)
        assert prov.get_rng_state() == state                                     # whole pairs, as randn


def test_exponential_edge_uniforms_and_accuracy(prov, oracle):
    """u = 0 stands for f64::MIN_POSITIVE (random.rs:296): -mu ln(2^-1022); u just below one keeps the logarithm's RELATIVE accuracy;
    against an 80-bit reference over a long stream the draws stay within 2 ulp."""
    a, mask = 6364136223846793005, (1 << 64) - 1
    ainv = pow(a, -1, 1 << 64)
    before = lambda x: ((x - 1) * ainv) & mask  # noqa: E731
    for first in (0, 5, 1 << 11, ((1 << 53) - 1) << 11, 1 << 63, ((1 << 52) | (54 << 45)) << 11):
        s = before(first)
        prov.set_rng_state(s)
        got = prov.download(prov.random_exponential(2.0, (4, 1))).ravel()
        want, _ = oracle.rng_exponential(s, 2.0, 4)
        assert np.all(np.abs(got - want) <= 8e-16 * np.abs(want)), (first, got, want)
    seed, n = oracle.rng_default_seed(), 400000
    prov.set_rng_state(seed)
    got = prov.download(prov.random_exponential(1.0, (n, 1))).ravel()
    u, _ = oracle.rng_uniform(seed, n)
    ref = -np.log(np.maximum(u, 2.2250738585072014e-308).astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    assert float(np.max(np.abs(got.astype(np.longdouble) - ref) / ulp)) <= 2.0


def test_f32_provider_rounds_the_same_stream(prov32, oracle):
    seed = oracle.rng_default_seed()
    prov32.set_rng_state(seed)
    got = prov32.download(prov32.random_unifrnd(2.0, 5.0, (1001, 1))).ravel()
    want, state = oracle.rng_unifrnd(seed, 2.0, 5.0, 1001)
    assert np.array_equal(got, want.astype(np.float32).astype(np.float64)) and prov32.get_rng_state() == state
    prov32.set_rng_state(seed)
    got = prov32.download(prov32.random_integer_range(-7, 7, (1001, 1))).ravel()
    assert np.array_equal(got, oracle.rng_integer_range(seed, -7, 7, 1001)[0])


def test_moments_at_full_size(prov):
    """1e8 draws (BASELINE's Monte-Carlo size): the sample moments of every distribution within 6 sigma of their values."""
    n = 10**8
    prov.rng_seed(7)
    for make, mean, var in ((lambda: prov.random_unifrnd(2.0, 5.0, (n, 1)), 3.5, 0.75), (lambda: prov.random_exponential(2.0, (n, 1)), 2.0, 4.0),
                            (lambda: prov.random_normrnd(1.0, 3.0, (n, 1)), 1.0, 9.0), (lambda: prov.random_integer_range(1, 6, (n, 1)), 3.5, 35.0 / 12.0)):
        h = make()
        m = float(prov.download(prov.reduce_mean(h)).ravel()[0])
        assert abs(m - mean) <= 6.0 * np.sqrt(var / n), (mean, m)
        prov.free(h)
