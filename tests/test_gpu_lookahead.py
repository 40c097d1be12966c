"""GPU tests of the blocked look-ahead LU driver (lu.hip getrf_blocked): the control flow every solve with
n >= 5120 takes, including BASELINE.json configs[4] (x = A\\b at 16384 x 16384).

The reference pins `A\\b` by residual and forward error (crates/runmat-runtime/src/builtins/math/linalg/ops/
mldivide.rs:662-696: residual < 1e-12 on its 2x2, < 1e-10 on its least-squares case); SURVEY.md 8(d) config 5
states the bounds at size: ||A x - b|| / (||A|| ||x||) <= 1e-12 n and ||x - 1||_inf <= 1e-9 on A = U(-1,1) + n I.
Pivot vectors are integer work and must be identical between every driver / panel variant and the oracle
(crates/runmat-accelerate/src/host_lu.rs:37-59)."""
import contextlib
import json
import math
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"


@contextlib.contextmanager
def env(**kv):
    """Developer knobs of lu.hip are read with getenv on every factorisation."""
    old = {k: os.environ.get(k) for k in kv}
    try:
        for k, v in kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _system(prov, n, seed, dominant):
    """Device matrix + host copy.  dominant: SURVEY.md 8(d) config 5 generator U(-1,1) + n I; else bench.py's U(-1,1)."""
    hu = prov.fill_uniform(seed, -1.0, 1.0, (n, n))
    A = prov.download_matrix(hu)
    if dominant:
        A[np.diag_indices(n)] += float(n)
        prov.free(hu)
        hu = prov.upload(A)
    b = A @ np.ones((n, 1))
    return hu, A, b


def _check_solution(A, b, x, n, fwd):
    x = x.reshape(-1, 1)
    assert np.all(np.isfinite(x))
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-12 * n, f"relative residual {res:.3e}"
    err = float(np.max(np.abs(x - 1.0)))
    assert err <= fwd, f"forward error {err:.3e}"


@pytest.mark.parametrize("n", [5250, 6144, 8192])
def test_lookahead_driver_solve_and_pivots(prov, n):
    """Default knobs take the look-ahead driver at these sizes (threshold 5120; 5250 = 82 base panels + 2 columns: ragged
    last panel, trailing blocks that are no multiples of the dgemm tiles).  Solution bounds, run-to-run
    determinism, and the pivot vector against the single-stream recursive driver and the per-column panels."""
    ha, A, b = _system(prov, n, 31 + n, dominant=True)
    hb = prov.upload(b)
    x1 = prov.download(prov.mldivide(ha, hb))
    _check_solution(A, b, x1, n, 1e-9)
    x2 = prov.download(prov.mldivide(ha, hb))
    assert np.array_equal(x1, x2), "two solves of the same system must agree bit for bit"
    prov.free(ha)
    # pivots on a matrix that really pivots (no dominant diagonal)
    hg, G, _ = _system(prov, n, 77 + n, dominant=False)
    r = prov.lu(hg)
    piv_la = prov.download(r.perm_vector).astype(np.int64)
    L = prov.download_matrix(r.lower)
    hlu = prov.matmul(r.lower, r.upper)  # the product on the device (fp64 MFMA): 2 n^3 flop would take the host minutes
    LU = prov.download_matrix(hlu)
    assert sorted(piv_la.tolist()) == list(range(1, n + 1))
    assert np.count_nonzero(piv_la != np.arange(1, n + 1)) > n // 2  # the generator does exercise the interchanges
    assert np.max(np.abs(L)) <= 1.0  # partial pivoting: |l_ij| <= 1, exactly (multipliers are quotients by the column maximum)
    assert np.max(np.abs(G[piv_la - 1, :] - LU)) <= 1e-12 * n
    for h in (r.combined, r.lower, r.upper, r.perm_matrix, r.perm_vector, hlu):
        prov.free(h)
    del L, LU
    with env(RMHIP_LU_LOOKAHEAD="0"):
        r0 = prov.lu(hg)
        piv_rec = prov.download(r0.perm_vector).astype(np.int64)
        for h in (r0.combined, r0.lower, r0.upper, r0.perm_matrix, r0.perm_vector):
            prov.free(h)
    assert np.array_equal(piv_la, piv_rec), "look-ahead and recursive drivers must pick the same pivots"
    with env(RMHIP_LU_PANEL="columns"):
        rc = prov.lu(hg)
        piv_col = prov.download(rc.perm_vector).astype(np.int64)
        for h in (rc.combined, rc.lower, rc.upper, rc.perm_matrix, rc.perm_vector):
            prov.free(h)
    assert np.array_equal(piv_la, piv_col), "persistent and per-column panels must pick the same pivots"
    prov.free(hg)


def test_lookahead_forced_at_2048_vs_oracle(prov, oracle):
    """The look-ahead driver forced at a size the oracle's host LU finishes in seconds: identical pivots, factors to
    rounding (host_lu.rs:19-119 restated in oracle.c)."""
    n = 2048
    A = np.random.default_rng(2048).uniform(-1, 1, (n, n))
    comb, L, U, P, piv = oracle.lu(A)
    for nb in (128, 256):
        with env(RMHIP_LU_LOOKAHEAD="1", RMHIP_LU_NB=nb):
            r = prov.lu(prov.upload(A))
            g_piv = prov.download(r.perm_vector)
            g_comb = prov.download_matrix(r.combined)
        assert np.array_equal(g_piv, piv.reshape(-1)), f"nb={nb}: pivot vector differs from the host LU"
        assert np.max(np.abs(g_comb - comb)) <= 1e-11 * max(1.0, np.abs(comb).max())
        for h in (r.combined, r.lower, r.upper, r.perm_matrix, r.perm_vector):
            prov.free(h)


@pytest.mark.parametrize("dominant", [True, False], ids=["U+nI", "U"])
def test_mldivide_16384(prov, dominant):
    """BASELINE.json configs[4] at full size on one GPU, on SURVEY.md 8(d)'s generator and on bench.py's."""
    n = 16384
    ha, A, b = _system(prov, n, 31 if dominant else 41, dominant)
    hb = prov.upload(b)
    hx = prov.mldivide(ha, hb)
    x = prov.download(hx)
    # U(-1,1) without the diagonal shift has a condition number of ~1e5-1e6 at this size: the forward error is
    # cond * eps, the residual bound is the same
    _check_solution(A, b, x, n, 1e-9 if dominant else 1e-7)
    hx2 = prov.mldivide(ha, hb)
    assert np.array_equal(x, prov.download(hx2)), "run-to-run determinism at 16384"
    for h in (ha, hb, hx, hx2):
        prov.free(h)


def _oracle_prices():
    return json.loads((GOLDEN / "monte_carlo_rng_oracle.json").read_text())["cases"]


@pytest.mark.parametrize("case", _oracle_prices(), ids=lambda c: f"M{c['M']}_T{c['T']}")
def test_monte_carlo_price_full_size(prov, case):
    """BASELINE.json configs[3]: the 1e8-sample price (and the benchmark-shaped 1e6 x 256) against the number the CPU
    oracle computed here (tests/golden/make_oracle_numbers.py); rel 1e-10 per SURVEY.md 8(d) config 4; the RNG end
    state is integer work and must match bit for bit.  All three request shapes of the workload are checked.
    NOTE on what pins this golden: it is the ORACLE's own output (the C restatement of random.rs / the builtins), not a number the
    reference produced - the reference-script pin (tests/golden/monte_carlo_lcg.json, reference_f64.json, from the imported Python
    comparators) stops at M = 200 000; the oracle is pinned on those and on the reference's sequence definitions (test_oracle_kats.py)."""
    from runmat_amd import sharding as sh

    g = sh.Group()
    M, T, want = case["M"], case["T"], case["price"]
    s0 = case["seed_state"]
    from planner_requests import monte_carlo_shaders

    shaders = monte_carlo_shaders(100.0)

    def evolved(prov, g, M, T, rng_state):
        return sh.monte_carlo_price_evolved(prov, g, M, T, rng_state=rng_state, payoff_shader=shaders[1])

    def fused(prov, g, M, T, rng_state):
        return sh.monte_carlo_price_fused(prov, g, M, T, shaders, rng_state=rng_state)

    evolved.__name__, fused.__name__ = "monte_carlo_price_evolved", "monte_carlo_price_fused"
    forms = [evolved]
    if T <= 4:
        forms += [fused, sh.monte_carlo_price_sharded]
    for f in forms:
        price, state = f(prov, g, M, T, rng_state=s0)
        assert state == case["final_state"], f.__name__
        assert abs(price - want) <= 1e-10 * want, f"{f.__name__}: {price!r} vs {want!r}"
        assert math.isfinite(price)


@pytest.mark.parametrize("n,nrhs", [(1024, 1), (1000, 3), (1536, 4), (1111, 5), (2048, 8)])
def test_substitution_chain_kernel_vs_launch_per_block(prov, oracle, n, nrhs):
    """x = A\\B with few right-hand sides runs its forward and backward substitution as one launch per direction
    (k_subst_chain: one workgroup per 128 rows, flags between them).  Same answer as the launch-per-block form
    (RMHIP_LU_SUBST=pair) up to the summation order, ragged last block and 4 + 1 / 4 + 4 column splits included."""
    rng = np.random.default_rng(900 + n + nrhs)
    A = rng.uniform(-1.0, 1.0, (n, n)) + n * np.eye(n)
    B = rng.uniform(-1.0, 1.0, (n, nrhs))
    ha, hb = prov.upload(A), prov.upload(B)
    x_chain = prov.download_matrix(prov.mldivide(ha, hb))
    with env(RMHIP_LU_SUBST="pair"):
        x_pair = prov.download_matrix(prov.mldivide(ha, hb))
    scale = np.max(np.abs(x_pair))
    assert np.max(np.abs(x_chain - x_pair)) <= 1e-13 * scale
    res = np.linalg.norm(A @ x_chain - B) / (np.linalg.norm(A) * np.linalg.norm(x_chain))
    assert res <= 1e-14 * n
    assert np.array_equal(x_chain, prov.download_matrix(prov.mldivide(ha, hb)))  # run-to-run deterministic
    prov.free(ha)
    prov.free(hb)


def test_substitution_chain_timeout_falls_back(built):
    """A spin of the chain kernel that times out (never seen; forced by RMHIP_LU_TEST_SUBST_RETRY) clobbers X: the
    right-hand side is gathered again and solved by the launch-per-block form, and the context keeps that form."""
    from runmat_amd import HipProvider

    p2 = HipProvider(0)
    rng = np.random.default_rng(6)
    n = 1280
    A = rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    B = rng.uniform(-1, 1, (n, 3))
    ha, hb = p2.upload(A), p2.upload(B)
    with env(RMHIP_LU_SUBST="pair"):
        want = p2.download_matrix(p2.mldivide(ha, hb))
    with env(RMHIP_LU_TEST_SUBST_RETRY="1"):
        got = p2.download_matrix(p2.mldivide(ha, hb))
    assert np.array_equal(got, want)  # the fallback IS the pair form
    again = p2.download_matrix(p2.mldivide(ha, hb))  # the context stays on it
    assert np.array_equal(again, want)
    p2.close()


@pytest.mark.parametrize("rows,cols", [(5632, 6400), (6400, 5376)])
def test_lookahead_driver_rectangular_lu(prov, rows, cols):
    """`lu` of wide and tall matrices through the look-ahead driver (min(rows, cols) >= 5120): pivots identical to the single-stream
    recursive driver, |L| <= 1, P A = L U."""
    hg = prov.fill_uniform(500 + rows, -1.0, 1.0, (rows, cols))
    G = prov.download_matrix(hg)
    r = prov.lu(hg)
    piv = prov.download(r.perm_vector).astype(np.int64)
    hlu = prov.matmul(r.lower, r.upper)
    LU = prov.download_matrix(hlu)
    L = prov.download_matrix(r.lower)
    assert sorted(piv.tolist()) == list(range(1, rows + 1))
    assert np.max(np.abs(L)) <= 1.0
    assert np.max(np.abs(G[piv - 1, :] - LU)) <= 1e-12 * max(rows, cols)
    for h in (r.combined, r.lower, r.upper, r.perm_matrix, r.perm_vector, hlu):
        prov.free(h)
    del L, LU
    with env(RMHIP_LU_LOOKAHEAD="0"):
        r0 = prov.lu(hg)
        piv0 = prov.download(r0.perm_vector).astype(np.int64)
        for h in (r0.combined, r0.lower, r0.upper, r0.perm_matrix, r0.perm_vector):
            prov.free(h)
    assert np.array_equal(piv, piv0)
    prov.free(hg)


@pytest.mark.parametrize("n,reps", [(12288, 12)])
def test_lookahead_driver_repeated_solves_are_identical(prov, n, reps):
    """Three streams run the factorisation (panels, dgemm, interchange + solve of the other column half), and once the panels
    sit on one XCD the update kernels are persistent workgroups that leave that XCD - deciding on a word the panel kernel
    writes while they start.  Every repeat must give the bits of the first and a small residual (a workgroup that split
    over that decision once left half of some 128 x 128 tiles un-updated in one solve out of four at this size)."""
    hu = prov.fill_uniform(41, -1.0, 1.0, (n, n))
    hb = prov.fill_uniform(42, -1.0, 1.0, (n, 1))
    first = None
    for rep in range(reps):
        hx = prov.mldivide(hu, hb)
        hr = prov.elem_sub(prov.matmul(hu, hx), hb)
        res = float(np.abs(prov.download(hr)).max())
        x = prov.download(hx).ravel()
        prov.free(hx)
        prov.free(hr)
        assert res < 1e-7, f"solve {rep}: max |A x - b| = {res:.3e}"
        if first is None:
            first = x
        assert np.array_equal(x, first), f"solve {rep} differs from solve 0"
    prov.free(hu)
    prov.free(hb)
