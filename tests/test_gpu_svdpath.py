"""GPU tests of the Jacobi-SVD path (svdsolve.hip): what the LU / Gram solves refuse - singular, rank-deficient and ill-conditioned
systems with min(rows, cols) <= 4096 - gets the reference's own answer on the device: the minimum-norm least-squares solution of an SVD
with the tolerance eps * max(m, n) * max(s_max, 1) (crates/runmat-runtime/src/builtins/math/linalg/ops/mldivide.rs:380-404), checked against
the oracle's restatement (oracle.c `orc_mldivide_svd`, the same one-sided Jacobi) to 1e-10 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(prov, oracle, A, B, tol=1e-10):
    s0 = prov.lu_stats()["svd_solves"]
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
    want = oracle.mldivide_svd(A, B)
    assert x.shape == want.shape
    assert np.max(np.abs(x - want)) <= tol * max(1.0, float(np.max(np.abs(want)))), float(np.max(np.abs(x - want)))
    return prov.lu_stats()["svd_solves"] - s0


def test_reference_kats_through_the_svd_path(prov, oracle):
    """mldivide.rs:662-696: the 2 x 2 system and the 3 x 2 least-squares case (full rank: the LU / Gram paths answer them, the SVD path is
    not needed), then the same shapes made singular."""
    assert _check(prov, oracle, np.array([[1.0, 2.0], [3.0, 4.0]]), np.array([[5.0], [6.0]])) == 0
    assert _check(prov, oracle, np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]), np.array([[7.0], [8.0], [9.0]])) == 0
    assert _check(prov, oracle, np.array([[1.0, 2.0], [2.0, 4.0]]), np.array([[1.0], [1.0]])) == 1          # singular square: min-norm solution
    assert _check(prov, oracle, np.ones((3, 2)), np.ones((3, 1))) == 1                                      # rank 1 least squares
    assert _check(prov, oracle, np.ones((2, 3)), np.array([[1.0], [2.0]])) == 1                             # rank 1, wide: min norm of an inconsistent system


@pytest.mark.parametrize("m,n,rank", [(300, 40, 25), (40, 300, 25), (257, 129, 129), (500, 500, 499), (500, 500, 100), (1000, 64, 1), (64, 1000, 63)])
def test_rank_deficient_systems_match_the_svd_oracle(prov, oracle, m, n, rank):
    rng = np.random.default_rng(m + 3 * n + rank)
    A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))  # exact rank `rank` up to rounding
    B = rng.standard_normal((m, 2))
    took = _check(prov, oracle, A, B, tol=1e-9)
    assert took == (1 if rank < min(m, n) else 0) or took == 1  # a full-rank case may still be refused by the Gram guard


def test_ill_conditioned_least_squares(prov, oracle):
    """cond(A) = 1e10: beyond the Gram route's guard (it squares the condition number), well inside the SVD's; singular values down to
    1e-10 are kept by the tolerance rule (tol ~ 300 * eps), so this is a full-rank solve that needs relative accuracy in the small ones."""
    rng = np.random.default_rng(8)
    q1, _ = np.linalg.qr(rng.standard_normal((300, 60)))
    q2, _ = np.linalg.qr(rng.standard_normal((60, 60)))
    A = (q1 * np.logspace(0, -10, 60)) @ q2
    x_true = rng.standard_normal((60, 1))
    B = A @ x_true
    s0 = prov.lu_stats()["svd_solves"]
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
    assert prov.lu_stats()["svd_solves"] == s0 + 1
    want = oracle.mldivide_svd(A, B)
    assert np.linalg.norm(x - want) <= 2e-5 * np.linalg.norm(want)          # both carry cond * eps = 1e-6 of forward error, times a constant
    assert np.linalg.norm(A @ x - B) <= 1e-12 * np.linalg.norm(B) * 60      # the residual is what the reference's tests pin


def test_linsolve_and_mrdivide_take_it_too(prov, oracle):
    from runmat_amd import ProviderLinsolveOptions

    rng = np.random.default_rng(2)
    A = rng.standard_normal((120, 7)) @ rng.standard_normal((7, 120))  # singular square
    B = rng.standard_normal((120, 3))
    s0 = prov.lu_stats()["svd_solves"]
    r = prov.linsolve(prov.upload(A), prov.upload(B), ProviderLinsolveOptions())
    x = prov.download_matrix(r.solution)
    assert np.max(np.abs(x - oracle.mldivide_svd(A, B))) <= 1e-9 * np.max(np.abs(x))
    y = prov.download_matrix(prov.mrdivide(prov.upload(B.T.copy()), prov.upload(A)))  # B' / A = (A' \\ B)'
    assert np.max(np.abs(y - oracle.mldivide_svd(A.T.copy(), B).T)) <= 1e-9 * np.max(np.abs(y))
    assert prov.lu_stats()["svd_solves"] == s0 + 2


def test_beyond_the_cap_is_still_handed_back(prov):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(3)
    n = 4224
    A = rng.standard_normal((n, 10)) @ rng.standard_normal((10, n))
    with pytest.raises(ProviderError) as e:
        prov.mldivide(prov.upload(A), prov.upload(np.ones((n, 1))))
    assert e.value.code == 7


@pytest.mark.parametrize("m,n,rank", [(1300, 1300, 1250), (2600, 1100, 1000), (1100, 2600, 1000)])
def test_rank_deficient_systems_beyond_1024_columns(prov, m, n, rank):
    """orders the oracle's scalar Jacobi would take minutes for: against LAPACK's minimum-norm least squares (numpy.linalg.lstsq, the same
    definition with a relative cut-off far from any singular value of these matrices)"""
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n)) / np.sqrt(rank)
    B = rng.standard_normal((m, 2))
    s0 = prov.lu_stats()["svd_solves"]
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
    assert prov.lu_stats()["svd_solves"] == s0 + 1
    want = np.linalg.lstsq(A, B, rcond=1e-12)[0]
    assert x.shape == want.shape and np.max(np.abs(x - want)) <= 1e-9 * np.max(np.abs(want))
