"""Oracle restatements of the shape / indexing hooks (oracle/oracle.c, "Shape / indexing hooks") pinned by the KATs the
reference's own tests hold for them, plus numpy cross-checks (independent implementations of the same tiling / permutation).
CPU only."""
import numpy as np
import pytest


def cm(data, shape):
    """column-major data + shape -> ndarray (how the reference's `Tensor::new(data, shape)` stores it)"""
    return np.asarray(data, dtype=np.float64).reshape(shape, order="F")


def test_repmat_reference_kats(oracle):
    # crates/runmat-runtime/src/builtins/array/shape/repmat.rs:753-782 repeats_matrix_with_vector_reps
    t = oracle.repmat(cm([1, 3, 2, 4], (2, 2)), [2, 3])
    assert t.shape == (4, 6)
    for col in range(6):
        assert list(t[:, col]) == ([1, 3, 1, 3] if col % 2 == 0 else [2, 4, 2, 4])
    # :797-809 scalar_replication_factor_expands_all_dims
    assert oracle.repmat(cm([1, 2, 3], (1, 3)), [2]).shape == (2, 6)
    # :811-845 repmat_high_dim_numeric
    base = np.arange(6.0)
    out = oracle.repmat(cm(base, (1, 3, 2)), [2, 1, 3])
    assert out.shape == (2, 3, 6)
    flat = out.reshape(-1, order="F")
    for k in range(6):
        for j in range(3):
            for i in range(2):
                assert flat[i + 2 * (j + 3 * k)] == base[(j % 3) + 3 * (k % 2)]
    # :1044-1060 repmat_gpu_roundtrip
    g = oracle.repmat(cm([1, 2], (2, 1)), [2])
    assert g.shape == (4, 2) and list(g.reshape(-1, order="F")) == [1, 2, 1, 2, 1, 2, 1, 2]
    # :913-929: a zero factor gives an empty tensor of the tiled shape
    assert oracle.repmat(cm([1, 2], (1, 2)), [0, 1]).shape == (0, 2)
    with pytest.raises(ValueError):
        oracle.repmat(cm([1.0], (1, 1)), [])


@pytest.mark.parametrize("shape,reps", [((2, 2), (2, 3)), ((3, 1), (1, 4)), ((1, 5), (3, 1)), ((2, 3, 4), (2, 1, 2)), ((4,), (3,)),
                                        ((2, 3), (2, 2, 2)), ((1, 1), (5, 7)), ((3, 2), (1, 1))])
def test_repmat_matches_numpy_tile(oracle, shape, reps):
    rng = np.random.default_rng(5)
    X = rng.standard_normal(shape)
    got = oracle.repmat(X, reps)
    # rank rule (simple_provider.rs:2179-2184): one factor = every dimension of max(rank, 2) dims
    rank = max(len(shape), 2) if len(reps) == 1 else max(len(shape), len(reps))
    full_reps = tuple(reps[0] for _ in range(rank)) if len(reps) == 1 else tuple(reps) + (1,) * (rank - len(reps))
    Xp = X.reshape(tuple(shape) + (1,) * (rank - len(shape)))
    want = np.tile(Xp, full_reps)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_permute_reference_kats(oracle):
    # crates/runmat-runtime/src/builtins/array/shape/permute.rs:618-630 permute_swaps_dims (order [2 1 3] one-based)
    t = cm(np.arange(1.0, 25.0), (2, 3, 4))
    out = oracle.permute(t, [1, 0, 2])
    assert out.shape == (3, 2, 4) and np.array_equal(out, np.transpose(t, (1, 0, 2)))
    # :634-645 permute_adds_trailing_dimension
    assert oracle.permute(cm([1, 2, 3], (1, 3)), [1, 0, 2]).shape == (3, 1, 1)
    with pytest.raises(ValueError, match="duplicate dimension index"):  # :649-655
        oracle.permute(cm(np.arange(6.0), (2, 3)), [1, 1])
    with pytest.raises(ValueError, match="at least the number of dimensions"):  # :690-700
        oracle.permute(cm(np.arange(8.0), (2, 2, 2)), [1, 0])
    # integer KAT :784-792: [1 MAX;3 4] column-major [1, MAX, 3, 4] permuted [2 1] -> [1, 3, MAX, 4]
    assert list(oracle.permute(cm([1, 99, 3, 4], (2, 2)), [1, 0]).reshape(-1, order="F")) == [1, 3, 99, 4]


@pytest.mark.parametrize("shape,order", [((2, 3, 4), (2, 0, 1)), ((5, 7), (1, 0)), ((3, 4, 5, 2), (3, 1, 0, 2)), ((6, 1, 4), (0, 2, 1)),
                                         ((2, 3), (0, 1, 2)), ((2, 3), (2, 0, 1))])
def test_permute_matches_numpy_transpose(oracle, shape, order):
    X = np.random.default_rng(6).standard_normal(shape)
    Xp = X.reshape(tuple(shape) + (1,) * (len(order) - len(shape)))
    assert np.array_equal(oracle.permute(X, order), np.transpose(Xp, order))


def test_linspace_reference_kats(oracle):
    # crates/runmat-runtime/src/builtins/array/creation/linspace.rs:495-512 linspace_basic
    t = oracle.linspace(0.0, 1.0, 5)
    assert t.shape == (1, 5) and np.max(np.abs(t[0] - [0.0, 0.25, 0.5, 0.75, 1.0])) < 1e-12
    d = oracle.linspace(-1.0, 1.0, 100)  # :526-538
    assert d.shape == (1, 100) and abs(d[0, 0] + 1.0) < 1e-12 and d[0, -1] == 1.0
    assert oracle.linspace(0.0, 10.0, 0).shape == (1, 0)  # :541-556
    assert oracle.linspace(5.0, 9.0, 1)[0, 0] == 9.0       # :559-572: a single point is `stop`
    # the benchmark's first statement (benchmarks/elementwise-math/runmat.m:10): the last element is stop exactly
    x = oracle.linspace(0.0, 4 * np.pi, 1001)
    assert x[0, -1] == 4 * np.pi and x[0, 0] == 0.0
    step = (4 * np.pi - 0.0) / 1000.0
    assert x[0, 7] == 0.0 + 7.0 * step


def test_gather_scatter_and_nan_maps(oracle):
    X = np.random.default_rng(8).standard_normal((4, 5))
    flat = X.reshape(-1, order="F")
    idx = [0, 19, 7, 7, 3]
    g = oracle.gather_linear(X, idx, (5, 1))
    assert g.shape == (5, 1) and list(g[:, 0]) == [flat[i] for i in idx]
    with pytest.raises(IndexError):
        oracle.gather_linear(X, [20], (1, 1))
    s = oracle.scatter_linear(X, [2, 5, 2], [10.0, 20.0, 30.0])  # the later duplicate wins (sequential loop)
    sf = s.reshape(-1, order="F")
    assert sf[2] == 30.0 and sf[5] == 20.0 and sf[0] == flat[0]
    v = np.array([1.0, np.nan, -0.0, np.inf, -np.nan])
    z = oracle.unary("nan_to_zero", v)
    assert list(z[[0, 1, 3, 4]]) == [1.0, 0.0, np.inf, 0.0] and np.signbit(z[2])
    assert list(oracle.unary("not_nan", v)) == [1.0, 0.0, 1.0, 1.0, 0.0]


def test_flip_circshift_tri_eye_reference_kats(oracle):
    m32 = cm([1, 4, 2, 5, 3, 6], (3, 2))
    # array/shape/flip.rs:740-761: vertical (dim 1) and horizontal (dim 2)
    assert list(oracle.flip(m32, [0]).reshape(-1, order="F")) == [2, 4, 1, 6, 3, 5]
    assert list(oracle.flip(m32, [1]).reshape(-1, order="F")) == [5, 3, 6, 1, 4, 2]
    assert np.array_equal(oracle.flip(m32, [0, 0]), m32)  # named twice: flipped back (simple_provider.rs:1757-1762)
    # array/shape/circshift.rs:971-1005
    assert list(oracle.circshift(cm([1, 2, 3, 4, 5], (5, 1)), [2]).reshape(-1)) == [4, 5, 1, 2, 3]
    m23 = cm([1, 4, 2, 5, 3, 6], (2, 3))
    assert list(oracle.circshift(m23, [0, -1]).reshape(-1, order="F")) == [2, 5, 3, 6, 1, 4]
    # array/shape/tril.rs:432-470, triu.rs:431-470
    assert list(oracle.tri(m23, False, 0).reshape(-1, order="F")) == [1, 4, 0, 5, 0, 0]
    assert list(oracle.tri(m23, False, 1).reshape(-1, order="F")) == [1, 4, 2, 5, 0, 6]
    assert list(oracle.tri(m23, False, -1).reshape(-1, order="F")) == [0, 4, 0, 0, 0, 0]
    assert list(oracle.tri(m23, True, 0).reshape(-1, order="F")) == [1, 0, 2, 5, 3, 6]
    assert list(oracle.tri(m23, True, 1).reshape(-1, order="F")) == [0, 0, 2, 0, 3, 6]
    assert list(oracle.tri(m23, True, -1).reshape(-1, order="F")) == [1, 4, 2, 5, 3, 6]
    # identity_data (simple_provider.rs:2293-2336): [n] is n x n, pages repeat the identity
    assert np.array_equal(oracle.eye([3]), np.eye(3)) and np.array_equal(oracle.eye([2, 3]), np.eye(2, 3))
    e = oracle.eye([2, 3, 2])
    assert e.shape == (2, 3, 2) and np.array_equal(e[:, :, 0], np.eye(2, 3)) and np.array_equal(e[:, :, 1], np.eye(2, 3))


@pytest.mark.parametrize("shape", [(4, 5), (3, 4, 5), (7,), (2, 1, 3)])
def test_flip_circshift_tri_match_numpy(oracle, shape):
    X = np.random.default_rng(9).standard_normal(shape)
    for axes in ([0], [len(shape) - 1], list(range(len(shape)))):
        assert np.array_equal(oracle.flip(X, axes), np.flip(X, axes))
    shifts = [2, -1, 5][:len(shape)]
    assert np.array_equal(oracle.circshift(X, shifts), np.roll(X, shifts, axis=tuple(range(len(shifts)))))
    if len(shape) >= 2:
        for off in (-2, 0, 1):
            want_l, want_u = X.copy(), X.copy()
            r, c = np.indices(shape[:2])
            ml = (r - c < -off)
            mu = (c - r < off)
            want_l[ml] = 0.0
            want_u[mu] = 0.0
            assert np.array_equal(oracle.tri(X, False, off), want_l) and np.array_equal(oracle.tri(X, True, off), want_u)
