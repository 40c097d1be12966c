"""GPU parity of the shape / indexing hooks (include/rmhip.h "shape / indexing hooks") through the C ABI: repmat (lazy view),
permute, fill_like, read_scalar, gather / scatter_linear, linspace, map_nan_to_zero / not_nan_mask.  All of it is data
movement or integer work: bit-exact against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits_equal(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64))


def broadcast_reps(a, b):
    """times.rs:568-598 `broadcast_reps`: trailing-padded shapes, factors that expand each operand to the output shape."""
    rank = max(len(a), len(b), 1)
    aa = [a[i] if i < len(a) else 1 for i in range(rank)]
    bb = [b[i] if i < len(b) else 1 for i in range(rank)]
    out = []
    for ad, bd in zip(aa, bb):
        if ad == bd or bd == 1:
            out.append(ad)
        elif ad == 1:
            out.append(bd)
        else:
            return None
    return tuple(out), [1 if aa[i] == out[i] else out[i] for i in range(rank)], [1 if bb[i] == out[i] else out[i] for i in range(rank)]


CALLER_SHAPES = [((4, 1), (1, 3)), ((2, 3), (2, 1)), ((1, 1), (40, 30)), ((7, 1, 5), (1, 6, 1)), ((1000, 1), (1, 1000)),
                 ((32, 5000), (1, 5000)), ((32, 1), (32, 5000)), ((3, 1, 70), (1, 700, 1)), ((127, 64), (127, 1)), ((2, 1), (1, 64)),
                 ((5, 7, 11, 13), (5, 1, 11, 1))]  # the shapes of test_gpu_parity.py::test_binary_broadcast_and_mismatch


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "pow"])
def test_callers_sequence_repmat_elem_free(prov, oracle, op):
    """The exact call sequence of plus / minus / times / rdivide / power on two resident operands of different shapes
    (math/elementwise/times.rs:501-543): broadcast_reps -> repmat each operand that needs it -> elem_* -> free the
    expansions.  Bit-exact against the CPU builtin's broadcast (oracle.binary)."""
    rng = np.random.default_rng(31)
    for sa, sb in CALLER_SHAPES:
        A, B = rng.uniform(0.5, 2.0, sa), rng.uniform(0.5, 2.0, sb)
        ha, hb = prov.upload(A), prov.upload(B)
        out_shape, reps_l, reps_r = broadcast_reps(sa, sb)
        made_l, made_r = any(r != 1 for r in reps_l), any(r != 1 for r in reps_r)
        le = prov.repmat(ha, reps_l) if made_l else ha
        re = prov.repmat(hb, reps_r) if made_r else hb
        assert le.shape == out_shape and re.shape == out_shape
        h = getattr(prov, "elem_" + op)(le, re)
        if made_l:
            prov.free(le)
        if made_r:
            prov.free(re)
        assert h.shape == out_shape
        want = oracle.binary(op, A, B)
        got = prov.download_matrix(h)
        if op == "pow":  # libm pow vs ocml pow: tolerance, as for the same-shape hook (test_gpu_parity.py)
            assert np.max(np.abs(got - want) / np.abs(want)) <= 4 * 2.3e-16, (sa, sb)
        else:
            assert bits_equal(got, want), (op, sa, sb)
        # the operands themselves are untouched and still usable
        assert bits_equal(prov.download_matrix(ha), A)
        for x in (ha, hb, h):
            prov.free(x)


def test_repmat_reference_kats_and_materialisation(prov, oracle):
    # repmat.rs:1044-1060 repmat_gpu_roundtrip and :1085-1115 repmat_wgpu_matches_cpu
    h = prov.upload(np.array([1.0, 2.0]), (2, 1))
    t = prov.repmat(h, [2])
    assert t.shape == (4, 2) and list(prov.download(t)) == [1, 2, 1, 2, 1, 2, 1, 2]
    prov.free(h)  # the base may go first: the view keeps the storage alive
    assert list(prov.download(t)) == [1, 2, 1, 2, 1, 2, 1, 2]
    prov.free(t)
    base = np.array([1.0, 4.0, 2.0, 5.0]).reshape((2, 2), order="F")
    hb = prov.upload(base)
    t = prov.repmat(hb, [2, 3])
    assert bits_equal(prov.download_matrix(t), oracle.repmat(base, [2, 3]))
    # a view of a view, a transpose of a view, a reshape of a view, a reduction over a view: every consumer sees the tiled tensor
    t2 = prov.repmat(t, [1, 2, 2])
    assert bits_equal(prov.download_matrix(t2), oracle.repmat(oracle.repmat(base, [2, 3]), [1, 2, 2]))
    tt = prov.transpose(t)
    assert bits_equal(prov.download_matrix(tt), oracle.repmat(base, [2, 3]).T)
    s = prov.reduce_sum_dim(t, 0)
    assert bits_equal(prov.download_matrix(s), oracle.reduce_sum(oracle.repmat(base, [2, 3]), [0]))
    for x in (hb, t, t2, tt, s):
        prov.free(x)


@pytest.mark.parametrize("shape,reps", [((2, 2), (2, 3)), ((3, 1), (1, 4)), ((1, 5), (3, 1)), ((2, 3, 4), (2, 1, 2)), ((4, 1), (3,)),
                                        ((2, 3), (2, 2, 2)), ((1, 1), (5, 7)), ((3, 2), (1, 1)), ((300, 7), (3, 2)), ((1, 5000), (32, 1)),
                                        ((1024, 1), (1, 700)), ((2, 3, 2, 3), (2, 2, 2, 2)), ((5, 4), (0, 2)), ((8192, 1), (1, 512))])
def test_repmat_materialised_vs_oracle(prov, oracle, shape, reps):
    X = np.random.default_rng(32).standard_normal(shape)
    h = prov.upload(X)
    t = prov.repmat(h, reps)
    want = oracle.repmat(X, reps)
    assert t.shape == want.shape
    assert bits_equal(prov.download_matrix(t), want)  # download materialises the view
    assert bits_equal(prov.download_matrix(t), want)  # ... once: the second read sees the settled buffer
    # a tiled view read in place against a dense operand of the tiled shape (true tilings: the dimension split)
    if want.size:
        D = np.random.default_rng(33).standard_normal(want.shape)
        hd, t2 = prov.upload(D), prov.repmat(h, reps)
        r = prov.elem_sub(t2, hd)
        assert bits_equal(prov.download_matrix(r), want - D)
        r2 = prov.elem_sub(hd, t2)
        assert bits_equal(prov.download_matrix(r2), D - want)
        for x in (hd, t2, r, r2):
            prov.free(x)
    prov.free(h)
    prov.free(t)


def test_repmat_views_in_fused_elementwise(prov, oracle):
    """A fusion group whose inputs are repmat views reads them in place (stride 0 over the base)."""
    from planner_requests import sin_mul_add_plan

    rng = np.random.default_rng(34)
    m, n = 257, 131
    a, b, cc = rng.uniform(-3, 3, (m, 1)), rng.uniform(-1, 1, (1, n)), rng.uniform(-1, 1, (m, n))
    plan, out_id = sin_mul_add_plan()
    shader = plan.generate_wgsl_for_output(out_id, "f64")
    ha, hb, hc = prov.upload(a), prov.upload(b), prov.upload(cc)
    va, vb = prov.repmat(ha, [1, n]), prov.repmat(hb, [m, 1])
    hd = prov.fused_elementwise(shader, [va, vb, hc], (m, n), m * n)
    want = oracle.sin_mul_add(np.tile(a, (1, n)), np.tile(b, (m, 1)), cc)
    got = prov.download_matrix(hd)
    assert np.max(np.abs(got - want)) <= 4e-16 * 2.0
    # conflicting tilings of one dimension: one of the views is materialised, the result is the same
    x = rng.standard_normal((6, 1))
    h2, h3 = prov.upload(x[:2]), prov.upload(x[:3])
    v2, v3 = prov.repmat(h2, [3, 1]), prov.repmat(h3, [2, 1])
    r = prov.elem_add(v2, v3)
    assert bits_equal(prov.download_matrix(r), np.tile(x[:2], (3, 1)) + np.tile(x[:3], (2, 1)))
    for h in (ha, hb, hc, va, vb, hd, h2, h3, v2, v3, r):
        prov.free(h)


def test_repmat_errors(prov):
    from runmat_amd import ProviderError

    h = prov.upload(np.ones((2, 2)))
    with pytest.raises(ProviderError):
        prov.repmat(h, [])
    prov.free(h)
    with pytest.raises(ProviderError) as e:
        prov.repmat(h, [2, 2])
    assert e.value.code == 5  # buffer not found


@pytest.mark.parametrize("shape,order", [((2, 3, 4), (1, 0, 2)), ((2, 3, 4), (2, 0, 1)), ((5, 7), (1, 0)), ((3, 4, 5, 2), (3, 1, 0, 2)),
                                         ((6, 1, 4), (0, 2, 1)), ((2, 3), (0, 1, 2)), ((2, 3), (2, 0, 1)), ((1, 3), (1, 0, 2)),
                                         ((300, 200), (1, 0)), ((65, 130, 9), (1, 0, 2)), ((65, 9, 130), (2, 1, 0)), ((70, 80, 6), (2, 0, 1)),
                                         ((4, 1000, 3), (1, 2, 0)), ((129, 3, 67), (0, 2, 1)), ((16, 8, 4, 8, 6), (4, 3, 2, 1, 0))])
def test_permute_vs_oracle(prov, oracle, shape, order):
    X = np.random.default_rng(35).standard_normal(shape)
    h = prov.upload(X)
    p = prov.permute(h, order)
    want = oracle.permute(X, order)
    assert p.shape == want.shape and bits_equal(prov.download_matrix(p), want)
    prov.free(h)
    prov.free(p)


def test_permute_mean_vecdim_sequence(prov, oracle):
    """mean(x, vecdim) on a resident tensor (reduction/mean.rs:975-1030): permute reduced dims first -> reshape [reduce_len,
    num_slices] -> reduce_mean_dim(0) -> reshape -> permute back."""
    X = np.random.default_rng(36).standard_normal((6, 5, 4))
    h = prov.upload(X)
    reduce_dims, kept = [0, 2], [1]
    order = reduce_dims + kept
    permuted = prov.permute(h, order)
    reduce_len = 6 * 4
    r2 = prov.reshape(permuted, (reduce_len, 5))
    red = prov.reduce_mean_dim(r2, 0)
    kept_shape = prov.reshape(red, (5,))
    expanded = prov.reshape(kept_shape, (1, 1, 5))
    inv = [0] * 3
    for dst, src in enumerate(order):
        inv[src] = dst
    out = prov.permute(expanded, inv)
    assert out.shape == (1, 5, 1)
    want = oracle.reduce_sum(oracle.permute(X, order).reshape((reduce_len, 5), order="F"), [0], mean=True)
    assert np.allclose(prov.download(out), want.reshape(-1), rtol=1e-14, atol=1e-15)  # summation order: tolerance, as test_gpu_parity's reductions
    from runmat_amd import ProviderError

    with pytest.raises(ProviderError, match="duplicate dimension index"):
        prov.permute(h, [1, 1, 0])
    with pytest.raises(ProviderError, match="at least the number of dimensions"):
        prov.permute(h, [1, 0])


def test_fill_like_read_scalar_linspace(prov, oracle):
    from runmat_amd import ProviderError

    X = np.random.default_rng(37).standard_normal((7, 9))
    h = prov.upload(X)
    for make, v in ((prov.zeros_like, 0.0), (prov.ones_like, 1.0), (lambda p: prov.fill_like(p, -2.5), -2.5)):
        f = make(h)
        assert f.shape == (7, 9) and np.all(prov.download(f) == v)
        prov.free(f)
    flat = X.reshape(-1, order="F")
    for i in (0, 1, 8, 62):
        assert prov.read_scalar(h, i) == flat[i]
    with pytest.raises(ProviderError, match="out of bounds"):
        prov.read_scalar(h, 63)
    # views are indexed in place
    t = prov.transpose(h)
    tf = X.T.reshape(-1, order="F")
    assert all(prov.read_scalar(t, i) == tf[i] for i in (0, 5, 10, 62))
    v = prov.repmat(h, [2, 3])
    vf = np.tile(X, (2, 3)).reshape(-1, order="F")
    assert all(prov.read_scalar(v, i) == vf[i] for i in (0, 7, 13, 14 * 9, 14 * 27 - 1))
    for x in (h, t, v):
        prov.free(x)
    # linspace: the first statement of benchmarks/elementwise-math/runmat.m:10
    for start, stop, count in ((0.0, 4 * np.pi, 1_048_576), (0.0, 1.0, 5), (-1.0, 1.0, 100), (5.0, 9.0, 1), (0.0, 10.0, 0), (3.0, -7.0, 12345)):
        l = prov.linspace(start, stop, count)
        assert l.shape == (1, count)
        assert bits_equal(prov.download(l), oracle.linspace(start, stop, count).reshape(-1)), (start, stop, count)
        prov.free(l)


def test_gather_scatter_linear(prov, oracle):
    from runmat_amd import ProviderError

    rng = np.random.default_rng(38)
    X = rng.standard_normal((50, 40))
    h = prov.upload(X)
    idx = rng.integers(0, X.size, 777)
    g = prov.gather_linear(h, idx, (777, 1))
    assert bits_equal(prov.download_matrix(g), oracle.gather_linear(X, idx, (777, 1)))
    with pytest.raises(ProviderError, match="out of bounds"):
        prov.gather_linear(h, [X.size], (1, 1))
    # gather from views
    t = prov.transpose(h)
    gt = prov.gather_linear(t, idx, (1, 777))
    assert bits_equal(prov.download_matrix(gt), oracle.gather_linear(X.T, idx, (1, 777)))
    # scatter with duplicates: the last occurrence wins
    sidx = np.concatenate([rng.integers(0, X.size, 300), [5, 5, 5, 17]])
    vals = rng.standard_normal(sidx.size)
    hv = prov.upload(vals)
    prov.scatter_linear(h, sidx, hv)
    assert bits_equal(prov.download_matrix(h), oracle.scatter_linear(X, sidx, vals))
    with pytest.raises(ProviderError):
        prov.scatter_linear(h, [0, 1], hv)  # values / index count mismatch
    for x in (h, g, t, gt, hv):
        prov.free(x)


def test_scatter_linear_large_index_sets_resolve_on_the_device(prov, oracle):
    """From 4096 indices on, bounds and duplicates are resolved by two kernels (owner table: the last occurrence wins, as in the
    reference's sequential loop, simple_provider.rs:2698-2711); an out-of-bounds index fails before anything is stored."""
    from runmat_amd import ProviderError

    rng = np.random.default_rng(381)
    X = rng.standard_normal((300, 200))
    h = prov.upload(X)
    n = 50000  # heavy duplication: 60000 cells, 50000 draws + forced repeats at both ends of the position range
    sidx = rng.integers(0, X.size, n).astype(np.uint32)
    sidx[-3:] = sidx[:3]
    sidx[100:200] = 4242
    vals = rng.standard_normal(n)
    hv = prov.upload(vals)
    prov.scatter_linear(h, sidx, hv)
    want = oracle.scatter_linear(X, sidx, vals)
    assert bits_equal(prov.download_matrix(h), want)
    # out of bounds somewhere in the middle: error names the first offending position, the target keeps its values
    bad = sidx.copy()
    bad[31000] = X.size
    bad[45000] = X.size + 7
    with pytest.raises(ProviderError, match=r"position 31000\) out of bounds"):
        prov.scatter_linear(h, bad, hv)
    assert bits_equal(prov.download_matrix(h), want)
    # gather of a large index set: the bounds check rides in the kernel
    g = prov.gather_linear(h, sidx, (n, 1))
    assert bits_equal(prov.download_matrix(g), oracle.gather_linear(want, sidx, (n, 1)))
    prov.free(g)
    with pytest.raises(ProviderError, match=r"position 31000\) out of bounds"):
        prov.gather_linear(h, bad, (n, 1))
    # exactly at the threshold, and one below it (host path): same result
    for m in (4096, 4095):
        h2 = prov.upload(X)
        hv2 = prov.upload(vals[:m])
        prov.scatter_linear(h2, sidx[:m], hv2)
        assert bits_equal(prov.download_matrix(h2), oracle.scatter_linear(X, sidx[:m], vals[:m])), m
        prov.free(h2)
        prov.free(hv2)
    prov.free(h)
    prov.free(hv)


def test_nan_maps_and_omitnan_sum_sequence(prov, oracle):
    """sum(x, 'omitnan') on a resident tensor (reduction/sum.rs:795): map_nan_to_zero, then the plain reduction."""
    X = np.random.default_rng(39).standard_normal((33, 17))
    X[3, 4] = np.nan
    X[0, 0] = -np.nan
    X[5, 5] = -0.0
    h = prov.upload(X)
    z = prov.map_nan_to_zero(h)
    m = prov.not_nan_mask(h)
    assert bits_equal(prov.download_matrix(z), oracle.unary("nan_to_zero", X))
    assert bits_equal(prov.download_matrix(m), oracle.unary("not_nan", X))
    s = prov.reduce_sum_dim(z, 0)
    assert np.allclose(prov.download_matrix(s), oracle.reduce_sum(X, [0], omitnan=True), rtol=1e-14, atol=1e-15)
    for x in (h, z, m, s):
        prov.free(x)


def test_hooks_on_f32_provider(oracle):
    """Precision-32 provider: the hooks move f32 storage as it is (no widen / narrow round trip)."""
    from runmat_amd import HipProvider

    p = HipProvider(0, precision="F32")
    try:
        X = np.random.default_rng(40).standard_normal((9, 6)).astype(np.float32).astype(np.float64)
        h = p.upload(X)
        t = p.repmat(h, [2, 3])
        assert p.buffer_bits(t) == 32 and bits_equal(p.download_matrix(t), oracle.repmat(X, [2, 3]))
        r = p.elem_mul(p.repmat(h, [2, 3]), t)
        assert bits_equal(p.download_matrix(r), (oracle.repmat(X, [2, 3]) ** 2).astype(np.float32).astype(np.float64))
        q = p.permute(h, [1, 0])
        assert p.buffer_bits(q) == 32 and bits_equal(p.download_matrix(q), X.T)
        g = p.gather_linear(h, [0, 53, 7], (3, 1))
        assert bits_equal(p.download(g), X.reshape(-1, order="F")[[0, 53, 7]])
        assert p.read_scalar(h, 11) == X.reshape(-1, order="F")[11]
        l = p.linspace(0.0, 1.0, 7)
        assert bits_equal(p.download(l), oracle.linspace(0.0, 1.0, 7).reshape(-1).astype(np.float32).astype(np.float64))
        z = p.map_nan_to_zero(h)
        assert p.buffer_bits(z) == 32 and bits_equal(p.download_matrix(z), X)
    finally:
        p.close()


@pytest.mark.parametrize("shape", [(4, 5), (3, 4, 5), (7, 1), (2, 1, 3), (300, 257), (1025, 3, 2), (5, 6, 7, 3)])
def test_flip_circshift_tri_vs_oracle(prov, oracle, shape):
    X = np.random.default_rng(41).standard_normal(shape)
    h = prov.upload(X)
    for axes in ([0], [len(shape) - 1], list(range(len(shape))), [0, 0], [len(shape) + 1]):
        f = prov.flip(h, axes)
        assert f.shape == tuple(shape) and bits_equal(prov.download_matrix(f), oracle.flip(X, axes)), axes
        prov.free(f)
    for shifts in ([2], [0, -1], [2, -1, 5][:len(shape)], [-7], [1] * (len(shape) + 1)):
        s = prov.circshift(h, shifts)
        assert bits_equal(prov.download_matrix(s), oracle.circshift(X, shifts)), shifts
        prov.free(s)
    for off in (-2, 0, 1, 1000):
        lo, up = prov.tril(h, off), prov.triu(h, off)
        assert bits_equal(prov.download_matrix(lo), oracle.tri(X, False, off)) and bits_equal(prov.download_matrix(up), oracle.tri(X, True, off))
        prov.free(lo)
        prov.free(up)
    prov.free(h)


def test_eye_and_cat(prov, oracle):
    from runmat_amd import ProviderError

    for shape in ([3], [2, 3], [5, 2], [2, 3, 2], [1025, 1030], []):
        e = prov.eye(shape)
        want = oracle.eye(shape)
        assert e.shape == want.shape and bits_equal(prov.download_matrix(e), want)
        prov.free(e)
    rng = np.random.default_rng(42)
    A, B, C3 = rng.standard_normal((4, 3)), rng.standard_normal((2, 3)), rng.standard_normal((4, 5))
    ha, hb, hc = prov.upload(A), prov.upload(B), prov.upload(C3)
    v = prov.cat(1, [ha, hb])            # vertical
    assert v.shape == (6, 3) and bits_equal(prov.download_matrix(v), np.vstack([A, B]))
    hcat = prov.cat(2, [ha, hc, ha])     # horizontal, three inputs
    assert hcat.shape == (4, 11) and bits_equal(prov.download_matrix(hcat), np.hstack([A, C3, A]))
    d3 = prov.cat(3, [ha, ha])           # a new trailing dimension
    assert d3.shape == (4, 3, 2) and bits_equal(prov.download_matrix(d3), np.stack([A, A], axis=2))
    X3, Y3 = rng.standard_normal((3, 2, 4)), rng.standard_normal((3, 5, 4))
    h3 = prov.cat(2, [prov.upload(X3), prov.upload(Y3)])
    assert h3.shape == (3, 7, 4) and bits_equal(prov.download_matrix(h3), np.concatenate([X3, Y3], axis=1))
    with pytest.raises(ProviderError, match="mismatch"):
        prov.cat(1, [ha, hc])
    with pytest.raises(ProviderError, match="at least two"):
        prov.cat(1, [ha])
    big = rng.standard_normal((2048, 700))
    hbig = prov.upload(big)
    assert bits_equal(prov.download_matrix(prov.cat(2, [hbig, hbig])), np.hstack([big, big]))


def test_views_keep_their_values_when_the_base_is_written_in_place(prov, oracle):
    """`B = repmat(A, 2, 2); A(idx) = v` (write_slice.rs:723 -> scatter_linear): the reference's repmat result is a buffer of its own,
    so B must keep the tiling of the OLD A whether or not it was materialised before the write; the same for a transpose view, and
    for the in-place block updates (rmhip_blk_assign) and a raw device pointer handed out for the base."""
    rng = np.random.default_rng(11)
    A = rng.uniform(-1, 1, (5, 3))
    old_tiled = np.tile(A, (2, 2))
    ha = prov.upload(A)
    view = prov.repmat(ha, [2, 2])          # lazy: shares A's storage
    tview = prov.transpose(ha)              # lazy as well
    vals = prov.upload(np.array([[9.0], [8.0]]))
    prov.scatter_linear(ha, np.array([0, 7], dtype=np.uint32), vals)
    A2 = A.copy(order="F")
    A2.reshape(-1, order="F")[[0, 7]] = [9.0, 8.0]
    assert np.array_equal(prov.download_matrix(ha), A2)
    assert np.array_equal(prov.download_matrix(view), old_tiled)
    assert np.array_equal(prov.download_matrix(tview), A.T)
    # a view made AFTER the write sees the new values, and reading it through the stride-0 elementwise path does too
    view2 = prov.repmat(ha, [1, 2])
    z = prov.upload(np.zeros((5, 6)))
    assert np.array_equal(prov.download_matrix(prov.elem_add(view2, z)), np.tile(A2, (1, 2)))
    # block-level in-place update of the base: the earlier view is untouched again
    view3 = prov.repmat(ha, [2, 1])
    blk = prov.upload(np.full((2, 2), -4.0))
    prov.blk_assign((ha, 1, 0, 2, 2), blk)
    assert np.array_equal(prov.download_matrix(view3), np.tile(A2, (2, 1)))
    A3 = A2.copy()
    A3[1:3, 0:2] = -4.0
    assert np.array_equal(prov.download_matrix(ha), A3)
    for h in (ha, view, tview, vals, view2, z, view3, blk):
        prov.free(h)
