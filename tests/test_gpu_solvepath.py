"""GPU tests of the SOLVE PATH of the LU (lu.hip: k_rp_top / k_rp_below): `mldivide` / `linsolve` / `mrdivide` factor with
pivoting restricted to each panel's top block and accept the factorisation only if every multiplier below the block
stays under tau; otherwise they refactor with the grid-wide rule `lu` uses.

What the reference pins for a solve is the solution, by residual (crates/runmat-runtime/src/builtins/math/linalg/ops/
mldivide.rs:662-696) - the pivot sequence never leaves the provider, unlike `lu`'s (host_lu.rs:37-59, covered by
tests/test_gpu_lookahead.py).  So the bar here is: backward error of partial-pivoting quality on every matrix class,
the fallback taken (and counted) where the restricted choice is not good enough, bit-reproducible results, and the
grid-wide path still reachable and correct."""
import contextlib
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def env(**kv):
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


def _backward_error(A, x, b):
    x = x.reshape(b.shape)
    return float(np.linalg.norm(A @ x - b, np.inf) / (np.linalg.norm(A, np.inf) * np.linalg.norm(x, np.inf) + np.linalg.norm(b, np.inf)))


def _matrix(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "U":  # bench.py's generator
        return rng.uniform(-1, 1, (n, n))
    if kind == "U+nI":
        return rng.uniform(-1, 1, (n, n)) + n * np.eye(n)
    if kind == "graded":  # singular values 1 .. 1e-10 between random orthogonal factors
        q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
        q2, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (q1 * np.logspace(0, -10, n)) @ q2
    raise ValueError(kind)


def _solve_both(prov, A, b):
    """(x on the solve path, x with the grid-wide rule forced, stats before / after the first)."""
    ha, hb = prov.upload(A), prov.upload(b)
    s0 = prov.lu_stats()
    hx = prov.mldivide(ha, hb)
    s1 = prov.lu_stats()
    with env(RMHIP_LU_FAST="0"):
        hy = prov.mldivide(ha, hb)
    x, y = prov.download(hx).ravel(), prov.download(hy).ravel()
    for h in (ha, hb, hx, hy):
        prov.free(h)
    return x, y, s0, s1


@pytest.mark.parametrize("kind,n", [("U", 4096), ("U", 8192), ("U+nI", 4096), ("U+nI", 8192), ("graded", 4096), ("U", 1000), ("U", 257),
                                    ("U+nI", 5250)])
def test_solve_path_matches_partial_pivoting_quality(prov, kind, n):
    """Accepted without a fallback, multipliers a small constant, backward error of the order partial pivoting gives on the same
    system (both are a few n * eps; the bound asserts 'within 8x and below 1e-13'), forward error bounded through it."""
    A = _matrix(kind, n, 100 + n)
    b = A @ np.ones((n, 1))
    x, y, s0, s1 = _solve_both(prov, A, b)
    assert s1["solve_path_factorizations"] == s0["solve_path_factorizations"] + 1
    assert s1["pivot_growth_fallbacks"] == s0["pivot_growth_fallbacks"]
    assert 0.0 <= s1["last_max_multiplier"] <= (4.0 if kind != "U+nI" else 0.01), s1
    bx, by = _backward_error(A, x, b), _backward_error(A, y, b)
    assert np.all(np.isfinite(x)) and bx <= 1e-13 and bx <= 8.0 * by + 1e-16, (bx, by)
    if kind != "graded":  # cond(graded) = 1e10: the forward error is cond * eps for either pivot rule
        assert np.max(np.abs(x - 1.0)) <= (1e-9 if kind == "U+nI" else 1e-7)
    else:
        assert np.max(np.abs(x - 1.0)) <= 50.0 * max(np.max(np.abs(y - 1.0)), 1e-7)


@pytest.mark.parametrize("n,nrhs", [(2100, 1), (4100, 3), (5001, 2)])
def test_ragged_orders_factor_at_the_padded_order(prov, n, nrhs):
    """n >= 2048 that is not a multiple of 128: the solves factor [A 0; 0 I] at the next multiple (every trailing update a whole
    tile; rmhip_ops.cpp: lu_pad_rows).  The padded rows hold zeros in A's columns and are never chosen as pivots; the solution agrees
    with the unpadded factorisation's (RMHIP_LU_PAD=0) to rounding - the panel boundaries, hence the grouping of the updates, depend
    on the order - and has the same backward error; linsolve and mrdivide take the same route."""
    rng = np.random.default_rng(n)
    A = rng.uniform(-1, 1, (n, n))
    B = rng.uniform(-1, 1, (n, nrhs))
    hA, hB = prov.upload(A), prov.upload(B)
    x1 = prov.download_matrix(prov.mldivide(hA, hB))
    with env(RMHIP_LU_PAD="0"):
        x0 = prov.download_matrix(prov.mldivide(hA, hB))
    assert x1.shape == (n, nrhs) and np.max(np.abs(x0 - x1)) <= 1e-9 * max(1.0, np.abs(x1).max())
    for j in range(nrhs):
        assert _backward_error(A, x1[:, j], B[:, j]) <= 64 * n * 2.3e-16 and _backward_error(A, x0[:, j], B[:, j]) <= 64 * n * 2.3e-16
    xl = prov.download_matrix(prov.linsolve(hA, hB).solution)
    assert np.array_equal(xl, x1)
    Bt = np.ascontiguousarray(B.T)
    xr = prov.download_matrix(prov.mrdivide(prov.upload(Bt), prov.upload(np.ascontiguousarray(A.T))))  # B' / A' = (A \ B)'
    assert xr.shape == (nrhs, n) and np.max(np.abs(xr.T - x1)) <= 1e-9 * max(1.0, np.abs(x1).max())


def test_solve_random_orders_fuzz(prov):
    """Orders across every driver switch - the single-panel sizes, the recursion, the look-ahead driver's threshold (1152), the padded
    orders (>= 2048, not a multiple of 128) - with 1 .. 4 right-hand sides: backward error of partial-pivoting quality for mldivide,
    the same solution from linsolve, and mrdivide on the transposed system."""
    rng = np.random.default_rng(77)
    orders = [1, 2, 3, 17, 63, 64, 65, 129, 255, 256, 257, 300, 511, 513, 700, 1023, 1025, 1151, 1152, 1153, 1280, 1500, 2047, 2049, 2100, 2176, 2500]
    orders += [int(v) for v in rng.integers(4, 900, 12)]
    for n in orders:
        nrhs = int(rng.integers(1, 5))
        A = rng.uniform(-1, 1, (n, n))
        B = rng.uniform(-1, 1, (n, nrhs))
        hA, hB = prov.upload(A), prov.upload(B)
        x = prov.download_matrix(prov.mldivide(hA, hB))
        assert x.shape == (n, nrhs)
        for j in range(nrhs):
            assert _backward_error(A, x[:, j], B[:, j]) <= 64 * max(n, 8) * 2.3e-16, (n, j)
        if n > 1:  # (scalar operands: linsolve hands back to the CPU path, as the reference's provider does)
            xl = prov.download_matrix(prov.linsolve(hA, hB).solution)
            assert np.array_equal(xl, x), n
        xr = prov.download_matrix(prov.mrdivide(prov.upload(np.ascontiguousarray(B.T)), prov.upload(np.ascontiguousarray(A.T))))
        assert np.max(np.abs(xr.T - x)) <= 1e-8 * max(1.0, np.abs(x).max()), n
        for h in (hA, hB):
            prov.free(h)


def test_wilkinson_growth_matrix(prov):
    """Wilkinson's matrix (1 on the diagonal and in the last column, -1 below the diagonal) doubles the last column at every
    step under partial pivoting: growth 2^(n-1).  Every multiplier is exactly 1, so the solve path accepts it - and must then
    be exactly as good (or bad) as the grid-wide rule: no pivoting happens under either, the factors are the same integers."""
    for n in (24, 40, 300):
        A = np.eye(n) - np.tril(np.ones((n, n)), -1)
        A[:, -1] = 1.0
        b = A @ np.ones((n, 1))
        x, y, s0, s1 = _solve_both(prov, A, b)
        # (the statistic covers the rows below the 256-row top block; inside it partial pivoting bounds them by 1)
        assert s1["pivot_growth_fallbacks"] == s0["pivot_growth_fallbacks"] and s1["last_max_multiplier"] == (1.0 if n > 256 else 0.0)
        assert np.array_equal(x, y) or np.allclose(x, y, rtol=0, atol=0, equal_nan=True)
        if n <= 40:  # 2^(n-1) eps is still small
            assert np.max(np.abs(x - 1.0)) <= 2.0 ** (n - 1) * 2.3e-16 * n


def test_permuted_dominant_matrix_falls_back_at_the_first_panel(prov):
    """Rows of U + nI shuffled: the large entries are spread over all rows and the top block holds one of them with probability
    256 / n, so the multipliers explode (~ n) - the first panel's check (one small read) hands the matrix to the grid-wide rule.
    The answer is then THAT rule's answer, bit for bit, and the refactorisation is counted."""
    n = 3000
    rng = np.random.default_rng(5)
    A = (rng.uniform(-1, 1, (n, n)) + n * np.eye(n))[rng.permutation(n)]
    b = A @ np.ones((n, 1))
    prov.reset_telemetry()
    x, y, s0, s1 = _solve_both(prov, A, b)
    assert s1["pivot_growth_fallbacks"] == s0["pivot_growth_fallbacks"] + 1
    assert s1["solve_path_factorizations"] == s0["solve_path_factorizations"]
    assert s1["last_max_multiplier"] > s1["tau"]
    assert np.array_equal(x, y)
    assert _backward_error(A, x, b) <= 1e-14
    assert dict(prov.telemetry_snapshot()["solve_fallbacks"]) == {"lu:pivot_growth": 1}


def test_forced_fallback_hook_and_tau_knob(prov):
    n = 2048
    A = _matrix("U", n, 77)
    b = A @ np.ones((n, 1))
    ha, hb = prov.upload(A), prov.upload(b)
    with env(RMHIP_LU_FAST="0"):
        want = prov.download(prov.mldivide(ha, hb))
    s0 = prov.lu_stats()
    with env(RMHIP_LU_TEST_GROWTH="1"):  # the check "fails" at the end of a complete solve-path factorisation
        got = prov.download(prov.mldivide(ha, hb))
    s1 = prov.lu_stats()
    assert np.array_equal(got, want) and s1["pivot_growth_fallbacks"] == s0["pivot_growth_fallbacks"] + 1
    with env(RMHIP_LU_TAU="1.25"):  # a bound this matrix's multipliers (~2) do not meet
        got = prov.download(prov.mldivide(ha, hb))
    s2 = prov.lu_stats()
    assert np.array_equal(got, want) and s2["pivot_growth_fallbacks"] == s1["pivot_growth_fallbacks"] + 1 and s2["tau"] == 1.25
    got = prov.download(prov.mldivide(ha, hb))  # back on the solve path
    s3 = prov.lu_stats()
    assert s3["solve_path_factorizations"] == s2["solve_path_factorizations"] + 1 and s3["tau"] == 8.0
    assert _backward_error(A, got, b) <= 1e-13


def test_singular_and_nan_inputs_still_end_in_the_reference_errors(prov):
    """A pivot at the cut-off inside a top block, or a NaN, makes the solve path hand over; the grid-wide rule then reports what the
    CPU's caller expects (SINGULAR -> its SVD path)."""
    from runmat_amd import ProviderError

    n = 1500
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, (n, n))
    A[:, 17] = A[:, 3]  # two equal columns: exactly singular
    os.environ["RMHIP_NO_SVD_PATH"] = "1"  # (without it the Jacobi-SVD path answers on the device: tests/test_gpu_svdpath.py)
    try:
        with pytest.raises(ProviderError) as e:
            prov.mldivide(prov.upload(A), prov.upload(np.ones((n, 1))))
    finally:
        del os.environ["RMHIP_NO_SVD_PATH"]
    assert e.value.code == 7
    B = rng.uniform(-1, 1, (n, n))
    B[400, 5] = np.nan
    try:
        x = prov.download(prov.mldivide(prov.upload(B), prov.upload(np.ones((n, 1)))))
        assert not np.all(np.isfinite(x))  # garbage in, NaN out - never a silently finite answer
    except ProviderError as err:
        assert err.code in (2, 7)


def test_least_squares_uses_the_solve_path_on_its_gram_matrix(prov, oracle):
    rng = np.random.default_rng(12)
    A = rng.uniform(-1, 1, (900, 300))
    b = rng.uniform(-1, 1, (900, 2))
    s0 = prov.lu_stats()
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(b)))
    s1 = prov.lu_stats()
    assert s1["solve_path_factorizations"] == s0["solve_path_factorizations"] + 1
    assert np.max(np.abs(x - oracle.mldivide_svd(A, b))) <= 1e-9 * np.max(np.abs(x))


@pytest.mark.parametrize("fast", ["1", "0"], ids=["solve-path", "grid-wide"])
@pytest.mark.parametrize("n,reps", [(8192, 8), (10240, 3), (13001, 3), (14336, 3), (16384, 3)])
def test_repeated_solves_are_bit_identical(prov, n, reps, fast):
    """Bit determinism and an on-device residual for every repeat, on both pivot rules, across the phase boundaries of the look-ahead
    driver (panel widths switch at 10240 / 8192 remaining rows, the third stream stops at 6144) and a size (13001) that is a
    multiple of nothing.  The grid-wide rule is the fallback of every solve and runs persistent workgroups that meet through
    flags: it stays under test although no default solve takes it any more."""
    hu = prov.fill_uniform(1000 + n, -1.0, 1.0, (n, n))
    hb = prov.fill_uniform(2000 + n, -1.0, 1.0, (n, 1))
    first = None
    with env(RMHIP_LU_FAST=fast):
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


@pytest.mark.parametrize("fast", ["1", "0"], ids=["solve-path", "grid-wide"])
def test_solve_under_real_contention(prov, built, fast):
    """A second context on the same GPU keeps every CU busy with 4096^3 dgemms from another thread while this one solves at
    n = 6144.  The solve path needs no co-residency and must simply give the uncontended bits.  The grid-wide rule's persistent
    panels may find their workgroups not co-resident: then the bounded spins expire, the context refactors on its next, more
    conservative path (counted in rmhip_lu_stats.panel_exchange_timeouts) - and the answer must still be right."""
    from runmat_amd import HipProvider

    n = 6144
    hu = prov.fill_uniform(71, -1.0, 1.0, (n, n))
    hb = prov.fill_uniform(72, -1.0, 1.0, (n, 1))
    with env(RMHIP_LU_FAST=fast):
        calm = prov.download(prov.mldivide(hu, hb))
        stop = threading.Event()
        started = threading.Event()

        def hog():
            other = HipProvider(0)
            try:
                a = other.fill_uniform(5, -1.0, 1.0, (4096, 4096))
                k = 0
                while not stop.is_set():
                    other.free(other.matmul(a, a))
                    started.set()
                    k += 1
                    if k % 8 == 0:  # a bounded queue: the hog must stop when told to, not seconds of queued products later
                        other.synchronize()
                other.synchronize()
            finally:
                other.close()

        th = threading.Thread(target=hog)
        th.start()
        try:
            assert started.wait(60)
            s0 = prov.lu_stats()
            outs = [prov.download(prov.mldivide(hu, hb)) for _ in range(2)]
            s1 = prov.lu_stats()
        finally:
            stop.set()
            th.join()
    timeouts = s1["panel_exchange_timeouts"] - s0["panel_exchange_timeouts"]
    if fast == "1":
        assert timeouts == 0
    A = prov.download_matrix(hu)
    b = prov.download_matrix(hb)
    for x in outs:
        if timeouts == 0:
            assert np.array_equal(x, calm)
        assert _backward_error(A, x, b) <= 1e-13  # a retried solve took another panel path: same quality, other bits
    prov.free(hu)
    prov.free(hb)


@pytest.mark.parametrize("n", [2, 3, 5, 8, 17, 31, 33, 63, 64, 65, 100, 127, 128])
def test_small_systems_in_one_launch(prov, oracle, n):
    """n <= 64 with up to 16 right-hand sides: one workgroup eliminates [A | B] in LDS with partial pivoting and substitutes back
    (small_solve.hip; the orders above 64 in the list take the blocked kernels and must meet the same assertions).  Same answer quality
    as the blocked path, same decisions on singular and nearly singular input."""
    rng = np.random.default_rng(1000 + n)
    for nrhs in (1, 3, 16):
        A = rng.uniform(-1, 1, (n, n))
        B = rng.uniform(-1, 1, (n, nrhs))
        ha, hb = prov.upload(A), prov.upload(B)
        s0 = prov.lu_stats()
        x = prov.download_matrix(prov.mldivide(ha, hb))
        s1 = prov.lu_stats()
        assert s1["solve_path_factorizations"] == s0["solve_path_factorizations"] + 1 and s1["last_max_multiplier"] == 0.0
        assert x.shape == (n, nrhs) and _backward_error(A, x, B) <= 1e-14
        with env(RMHIP_NO_SMALL_SOLVE="1"):
            y = prov.download_matrix(prov.mldivide(ha, hb))
        assert np.max(np.abs(x - y)) <= 1e-9 * max(1.0, float(np.abs(y).max()))  # the same system through the blocked kernels
        assert np.max(np.abs(x - oracle.mldivide_lu(A, B))) <= 1e-9 * max(1.0, float(np.abs(x).max()))
    # exact arithmetic: integers, unit lower triangular times upper triangular with a permutation -> the exact solution
    L = np.tril(rng.integers(-2, 3, (n, n)).astype(float), -1) + np.eye(n)
    U = np.triu(rng.integers(-2, 3, (n, n)).astype(float), 1) + np.diag(rng.choice([1.0, -1.0, 2.0], n))
    P = np.eye(n)[rng.permutation(n)]
    A = P @ L @ U
    xs = rng.integers(-3, 4, (n, 2)).astype(float)
    if n <= 17:  # entries stay small integers: every step of the elimination is exact
        x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(A @ xs)))
        assert np.max(np.abs(x - xs)) <= 1e-9


def test_small_systems_singular_nan_and_limits(prov, oracle):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(77)
    A = rng.uniform(-1, 1, (12, 12))
    A[:, 7] = A[:, 2]  # exactly singular: the SVD path answers with the reference's minimum-norm solution
    b = rng.uniform(-1, 1, (12, 2))
    s0 = prov.lu_stats()["svd_solves"]
    x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(b)))
    assert prov.lu_stats()["svd_solves"] == s0 + 1
    assert np.max(np.abs(x - oracle.mldivide_svd(A, b))) <= 1e-9 * max(1.0, float(np.abs(x).max()))
    with env(RMHIP_NO_SVD_PATH="1"):
        with pytest.raises(ProviderError) as e:
            prov.mldivide(prov.upload(A), prov.upload(b))
    assert e.value.code == 7
    H = 1.0 / (np.arange(1, 13)[:, None] + np.arange(12)[None, :])  # Hilbert 12: cond ~ 1e16 -> the reference drops singular values
    s0 = prov.lu_stats()["svd_solves"]
    xh = prov.download_matrix(prov.mldivide(prov.upload(H), prov.upload(np.ones((12, 1)))))
    assert prov.lu_stats()["svd_solves"] == s0 + 1  # no pivot below the cut-off, but a pivot ratio ~1e-16: the SVD decides
    assert np.all(np.isfinite(xh)) and np.max(np.abs(H @ xh - 1.0)) <= 1e-6
    N = rng.uniform(-1, 1, (20, 20))
    N[3, 4] = np.nan
    try:
        xn = prov.download(prov.mldivide(prov.upload(N), prov.upload(np.ones((20, 1)))))
        assert not np.all(np.isfinite(xn))
    except ProviderError as err:
        assert err.code in (2, 7)
    # one past the limits: the blocked path, same answers
    for n, nrhs in ((129, 1), (64, 17)):
        A = rng.uniform(-1, 1, (n, n)) + 2 * np.eye(n)
        B = rng.uniform(-1, 1, (n, nrhs))
        x = prov.download_matrix(prov.mldivide(prov.upload(A), prov.upload(B)))
        assert _backward_error(A, x, B) <= 1e-14
    # mrdivide and linsolve reach it through the same entry
    A = rng.uniform(-1, 1, (40, 40))
    B = rng.uniform(-1, 1, (5, 40))
    y = prov.download_matrix(prov.mrdivide(prov.upload(B), prov.upload(A)))
    assert np.max(np.abs(y @ A - B)) <= 1e-12 * 40 * np.linalg.norm(A) * max(1.0, np.linalg.norm(y))
