"""The oracle's restatements of diag_from_vector, kron, cross, gradient and issymmetric against the reference's own unit-test vectors
(tests/golden/linear_hooks_kats.json) and against numpy where numpy computes the same thing."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle

K = json.loads((Path(__file__).parent / "golden" / "linear_hooks_kats.json").read_text())


def arr(v, shape):
    return np.array(v, dtype=np.float64).reshape(shape, order="F")


def test_reference_kats():
    for k in K["kron"]:
        out = oracle.kron(arr(k["a"], k["sa"]), arr(k["b"], k["sb"]))
        assert list(out.shape) == k["so"] and np.array_equal(out.ravel(order="F"), k["out"])
    for k in K["cross"]:
        out = oracle.cross(arr(k["a"], k["shape"]), arr(k["b"], k["shape"]), k["dim"])
        assert np.array_equal(out.ravel(order="F"), k["out"])
    for k in K["gradient"]:
        out = oracle.gradient(arr(k["x"], k["shape"]), k["dim"], k["spacing"], k["coords"])
        assert np.array_equal(out.ravel(order="F"), np.array(k["out"])), k
    for k in K["issymmetric"]:
        assert oracle.issymmetric(arr(k["a"], k["shape"]), k["skew"], k["tol"]) == k["out"]
    for k in K["ishermitian"]:
        assert oracle.ishermitian(arr(k["a"], k["shape"]), k["skew"], k["tol"]) == k["out"], k
    for k in K["bandwidth"]:
        assert list(oracle.bandwidth(arr(k["a"], k["shape"]))) == k["out"], k


def test_against_numpy():
    rng = np.random.default_rng(2)
    a, b = rng.standard_normal((3, 4)), rng.standard_normal((2, 5))
    assert np.array_equal(oracle.kron(a, b), np.kron(a, b))
    a3, b3 = rng.standard_normal((2, 3, 2)), rng.standard_normal((3, 1, 2))
    assert np.array_equal(oracle.kron(a3, b3), np.kron(a3, b3))
    assert oracle.kron(np.zeros((0, 2)), b).shape == (0, 10)                      # kron.rs:704-712
    u, v = rng.standard_normal((5, 3, 4)), rng.standard_normal((5, 3, 4))
    assert np.allclose(oracle.cross(u, v), np.cross(u, v, axis=1), rtol=0, atol=1e-15)
    x = rng.standard_normal((7, 6, 3))
    for d in range(3):
        assert np.allclose(oracle.gradient(x, d, 0.5), np.gradient(x, 0.5, axis=d), rtol=0, atol=1e-14)
    assert np.array_equal(oracle.gradient(x, 3), np.zeros_like(x))                 # a dimension beyond the rank: extent one
    v = rng.standard_normal(5)
    for off in (-2, 0, 3):
        assert np.array_equal(oracle.diag_from_vector(v, off), np.diag(v, off))
    assert np.array_equal(oracle.diag_from_vector(v, 1, 3, 4), np.diag(v, 1)[:3, :4])
    s = a @ a.T
    assert oracle.issymmetric(s) and not oracle.issymmetric(a) and oracle.issymmetric(s - s.T, True)
    s2 = s.copy()
    s2[0, 1] += 1e-9
    assert not oracle.issymmetric(s2) and oracle.issymmetric(s2, tol=1e-8)
    s2[1, 2] = s2[2, 1] = np.nan
    assert not oracle.issymmetric(s2, tol=1.0)                                      # NaN never equals, never within (issymmetric.rs:517-526)
    s3 = s.copy()
    s3[0, 1] = s3[1, 0] = np.inf
    assert oracle.issymmetric(s3)                                                   # equal infinities pass the == test
    s5 = s.copy()
    s5[2, 2] = np.nan                                                               # the one place the two predicates differ: a NaN
    assert oracle.issymmetric(s5) and not oracle.ishermitian(s5)                    # diagonal (ishermitian.rs:462-465)
    assert oracle.ishermitian(s) and oracle.ishermitian(s - s.T, True) and not oracle.ishermitian(a)
    t = np.triu(rng.standard_normal((6, 9)), -2)
    t = np.tril(t, 3)
    t[2, 0], t[0, 3] = 1.0, 1.0
    assert oracle.bandwidth(t) == (2, 3) and oracle.bandwidth(np.zeros((4, 4))) == (0, 0)
    assert oracle.bandwidth(np.array([0.0, 0.0, 5.0])) == (0, 2)                    # a rank-1 shape is a row (bandwidth.rs:306)


def test_inv_restatement_on_the_reference_vectors_and_against_numpy():
    """inv.rs:365-383 (A = [4 -2; 1 3]: A * inv(A) = I to 1e-12), :402-412 (diag(4, 2) -> diag(0.25, 0.5) exactly), :484-488 (singular)."""
    a = np.array([4.0, 1.0, -2.0, 3.0]).reshape(2, 2, order="F")
    x = oracle.inv(a)
    assert np.max(np.abs(a @ x - np.eye(2))) < 1e-12
    assert np.array_equal(oracle.inv(np.diag([4.0, 2.0])), np.diag([0.25, 0.5]))
    assert oracle.inv(np.array([[1.0, 2.0], [2.0, 4.0]])) is None
    rng = np.random.default_rng(4)
    for n in (1, 3, 50, 200):
        m = rng.standard_normal((n, n)) + n * np.eye(n)
        assert np.max(np.abs(oracle.inv(m) - np.linalg.inv(m))) <= 1e-13


def test_index_hook_restatements_on_the_reference_vectors():
    """sub2ind.rs / ind2sub.rs unit tests (sub2ind([3 4], 2, 3) = 8 and back; out-of-range and non-integer subscripts refused),
    ndgrid.rs ([X, Y] = ndgrid(1:2, 1:3)), round.rs (round(2.345, 2) = 2.35 on doubles; halves away from zero), pow2 (ldexp on integers)."""
    assert oracle.sub2ind([3, 4], [1, 3], [np.array([2.0]), np.array([3.0])], [False, False], 1, [1, 1]).ravel()[0] == 8.0
    r, c = oracle.ind2sub([3, 4], [1, 3], np.array([[8.0]]), 12)
    assert r.ravel()[0] == 2.0 and c.ravel()[0] == 3.0
    assert oracle.sub2ind([3, 4], [1, 3], [np.array([4.0]), np.array([1.0])], [False, False], 1, [1, 1]) == (0, 0)
    assert oracle.sub2ind([3, 4], [1, 3], [np.array([1.0, 2.0]), np.array([1.0, 1.5])], [False, False], 2, [2, 1]) == (1, 1)
    assert oracle.ind2sub([3, 4], [1, 3], np.array([[1.0, 13.0]]), 12) == 1
    X, Y = oracle.ndgrid([np.array([1.0, 2.0]), np.array([1.0, 2.0, 3.0])], [2, 3], 2)
    assert np.array_equal(X, [[1, 1, 1], [2, 2, 2]]) and np.array_equal(Y, [[1, 2, 3], [1, 2, 3]])
    assert list(oracle.round_decimals(np.array([2.5, -2.5, 0.5, 0.49999999999999994, -0.2]), 0)) == [3.0, -3.0, 1.0, 0.0, -0.0]
    assert oracle.round_decimals(np.array([2.345]), 2)[0] == 2.35 and oracle.round_decimals(np.array([1234.0]), -2)[0] == 1200.0
    assert oracle.powi10(3) == 1000.0 and oracle.powi10(-2) == 0.01 and oracle.powi10(22) == 1e22 and oracle.powi10(400) == np.inf
    assert np.array_equal(oracle.pow2_scale(np.array([3.0, -1.5]), np.array([4.0, -1.0])), [48.0, -0.75])
    a = oracle.angle_real(np.array([1.0, -1.0, 0.0, -0.0]))
    assert list(a) == [0.0, np.pi, 0.0, np.pi]


def test_chol_restatement():
    """chol.rs unit tests: the SPD 3 x 3 [[4 12 -16]; [12 37 -43]; [-16 -43 98]] -> R = [[2 6 -8]; [0 1 5]; [0 0 3]]; a non-positive pivot
    reports its column and zeroes the rest; an asymmetric pair reports the column it is found in."""
    a = np.array([[4.0, 12, -16], [12, 37, -43], [-16, -43, 98]])
    r, info = oracle.chol(a)
    assert info == 0 and np.array_equal(r, [[2, 6, -8], [0, 1, 5], [0, 0, 3]])
    r, info = oracle.chol(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert info == 2 and np.array_equal(r, [[1.0, 2.0], [0.0, 0.0]])
    assert oracle.chol(np.array([[4.0, 2.0], [2.000001, 3.0]]))[1] == 2 and oracle.chol(np.array([[-1.0]]))[1] == 1
    rng = np.random.default_rng(5)
    b = rng.standard_normal((80, 80))
    a = b @ b.T + 80 * np.eye(80)
    r, info = oracle.chol(a)
    assert info == 0 and np.max(np.abs(r - np.linalg.cholesky(a).T)) < 1e-12 and np.array_equal(np.tril(r, -1), np.zeros((80, 80)))


def test_norm_restatement_on_the_reference_vectors():
    """norm.rs:795-960: norm([3 4]) = 5, norm([2 -7 4], Inf) = 7 / -Inf = 2, norm([0 0 5 0], 0) = 1, fro of diag(2, 1) = sqrt 5, the
    fractional p-norm; what needs singular values is refused here (the GPU hook returns UNSUPPORTED for it)."""
    assert oracle.norm(np.array([[3.0], [4.0]])) == 5.0 and oracle.norm(np.array([[2.0], [-7.0], [4.0]]), "inf") == 7.0
    assert oracle.norm(np.array([[2.0], [-7.0], [4.0]]), "-inf") == 2.0 and oracle.norm(np.array([[0.0], [0.0], [5.0], [0.0]]), "zero") == 1.0
    assert abs(oracle.norm(np.array([[2.0, 0.0], [0.0, 1.0]]), "fro") - np.sqrt(5.0)) < 1e-15
    assert abs(oracle.norm(np.array([[1.0], [2.0], [3.0]]), "p", 1.5) - (1 + 2 ** 1.5 + 3 ** 1.5) ** (1 / 1.5)) < 1e-14
    assert oracle.norm(np.array([[3.0, 0.0], [0.0, 1.0]]), "two") is None and oracle.norm(np.array([[1.0], [2.0]]), "p", 0.5) is None
    rng = np.random.default_rng(8)
    x = rng.standard_normal((50, 30))
    assert abs(oracle.norm(x, "fro") - np.linalg.norm(x)) < 1e-12 and abs(oracle.norm(x, "one") - np.linalg.norm(x, 1)) < 1e-12
    assert abs(oracle.norm(x, "inf") - np.linalg.norm(x, np.inf)) < 1e-12 and np.isnan(oracle.norm(np.array([[1.0], [np.nan]]), "inf"))
    assert oracle.norm(np.array([[1e200], [1e200]])) == np.sqrt(2.0) * 1e200                      # the running rescale: no overflow


def test_corrcoef_kat_and_numpy():
    # corrcoef.rs:1005-1034 (`corrcoef_matrix_basic`, tolerance 1e-10)
    m = np.array([1.0, 2.0, 3.0, 4.0, 2.0, 4.0, 6.0, 8.0, 4.0, 1.0, -1.0, 0.0]).reshape((4, 3), order="F")
    want = np.array([1.0, 1.0, -0.836660026534, 1.0, 1.0, -0.836660026534, -0.836660026534, -0.836660026534, 1.0]).reshape(3, 3)
    assert np.allclose(oracle.corrcoef(m), want, rtol=0, atol=1e-10)
    rng = np.random.default_rng(17)
    x = rng.standard_normal((50, 6))
    assert np.allclose(oracle.corrcoef(x), np.corrcoef(x, rowvar=False), rtol=0, atol=1e-14)
    assert np.allclose(oracle.corrcoef(x, "biased"), np.corrcoef(x, rowvar=False), rtol=0, atol=1e-14)   # the denominators cancel
    x[:, 2] = 3.0                                                                   # a constant column: NaN row / column, NaN diagonal
    x[7, 4] = np.nan
    r = oracle.corrcoef(x)
    assert np.isnan(r[2]).all() and np.isnan(r[:, 4]).all() and r[0, 0] == 1.0 and np.isfinite(r[0, 1])
    assert np.isnan(oracle.corrcoef(np.ones((1, 3)))).all() and oracle.corrcoef(np.zeros((4, 0))).shape == (0, 0)


def test_peaks_known_values():
    # peaks.rs:735-746 (`peaks_formula_known_value`): Z(0, 0) = exp(-1) * 8 / 3 within 1e-12
    assert abs(oracle.peaks_xy(0.0, 0.0) - np.exp(-1.0) * 8.0 / 3.0) < 1e-12
    z = oracle.peaks(49)
    assert z.shape == (49, 49) and abs(z.max() - 8.1) < 0.1 and abs(z.min() + 6.55) < 0.1      # MATLAB's surface: peak ~8.1, trough ~-6.55
    assert oracle.peaks(1).shape == (1, 1) and oracle.peaks(0).shape == (0, 0)


def test_rank_cond_pinv_reference_vectors():
    # rank.rs:337-341, 389-391, 397-401, 407-410
    assert oracle.rank(np.array([1.0, 3.0, 2.0, 4.0]).reshape(2, 2, order="F")) == 2
    assert oracle.rank(np.diag([1.0, 1e-16])) == 1
    assert oracle.rank(np.diag([1.0, 1e-4])) == 2 and oracle.rank(np.diag([1.0, 1e-4]), 1e-3) == 1 and oracle.rank(np.zeros((0, 0))) == 0
    # pinv.rs:404-440: a rank-one matrix, a tall selection matrix, a custom tolerance that drops 1e-12
    a = np.array([1.0, 2.0, 2.0, 4.0]).reshape(2, 2, order="F")
    assert np.allclose(oracle.pinv(a), np.linalg.pinv(a), atol=1e-14) and oracle.pinv(np.zeros((3, 0))).shape == (0, 3)
    t = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 1.0]).reshape(3, 2, order="F")
    assert np.allclose(oracle.pinv(t), t.T, atol=1e-15)
    assert np.allclose(oracle.pinv(np.diag([1.0, 1e-12]), 1e-6), np.diag([1.0, 0.0]), atol=1e-15)
    assert oracle.cond2(np.diag([4.0, 2.0])) == 2.0 and oracle.cond2(np.diag([1.0, 0.0])) == float("inf") and oracle.cond2(np.zeros((0, 3))) == 0.0
