"""`AccelProvider: Send + Sync` (crates/runmat-accelerate-api/src/lib.rs:1386): several host threads may call one
provider.  The library serialises calls per context (Context::call_mu); this test overlaps calls that share per-context
state - reductions (shared scratch), fused kernels (kernel cache), randn (RNG state), and a look-ahead LU solve (which
retargets the context stream) - from two threads and checks every result against a serial run of the same calls."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _work_reduce(prov, x_h, out, reps):
    for _ in range(reps):
        out.append((float(prov.download(prov.reduce_sum(x_h))[0]), prov.download(prov.reduce_sum_dim(x_h, 1)).copy()))


def _work_fused(prov, shader, hs, shape, out, reps):
    for _ in range(reps):
        h = prov.fused_elementwise(shader, hs, shape, shape[0] * shape[1])
        out.append(prov.download(h).copy())
        prov.free(h)


def _work_solve(prov, ha, hb, out, reps):
    for _ in range(reps):
        hx = prov.mldivide(ha, hb)
        out.append(prov.download(hx).copy())
        prov.free(hx)


def test_two_threads_one_context(prov):
    from planner_requests import sin_mul_add_plan

    rng = np.random.default_rng(3)
    n = 1536
    X = rng.uniform(-1, 1, (n, n))
    hx = prov.upload(X)
    plan, out_id = sin_mul_add_plan()
    shader = plan.generate_wgsl_for_output(out_id, "f64")
    A, B, C = (rng.uniform(-1, 1, (n, n)) for _ in range(3))
    hs = [prov.upload(A), prov.upload(B), prov.upload(C)]
    m = 5632  # >= 5120: the look-ahead driver with its second stream
    ha = prov.fill_uniform(5, -1.0, 1.0, (m, m))
    hb = prov.fill_uniform(6, -1.0, 1.0, (m, 1))
    # serial references
    ref_red, ref_fused, ref_solve = [], [], []
    _work_reduce(prov, hx, ref_red, 1)
    _work_fused(prov, shader, hs, (n, n), ref_fused, 1)
    _work_solve(prov, ha, hb, ref_solve, 1)
    reps = 6
    for other in ("reduce", "fused"):
        got_a, got_b = [], []
        ta = threading.Thread(target=_work_solve, args=(prov, ha, hb, got_a, 3))
        if other == "reduce":
            tb = threading.Thread(target=_work_reduce, args=(prov, hx, got_b, reps * 4))
        else:
            tb = threading.Thread(target=_work_fused, args=(prov, shader, hs, (n, n), got_b, reps * 4))
        ta.start(); tb.start(); ta.join(); tb.join()
        assert len(got_a) == 3 and all(np.array_equal(x, ref_solve[0]) for x in got_a), "LU solve disturbed by a concurrent caller"
        if other == "reduce":
            assert all(t == ref_red[0][0] and np.array_equal(v, ref_red[0][1]) for t, v in got_b)
        else:
            assert all(np.array_equal(v, ref_fused[0]) for v in got_b)
    # two threads drawing from the one RNG stream: the union of their draws is the serial stream (order of chunks may differ)
    prov.rng_seed(0)
    serial = prov.download(prov.random_normal((8 * 1000, 1))).reshape(8, 1000)
    prov.rng_seed(0)
    chunks = []

    def draw(k):
        for _ in range(k):
            chunks.append(prov.download(prov.random_normal((1000, 1))).copy())

    t1, t2 = threading.Thread(target=draw, args=(4,)), threading.Thread(target=draw, args=(4,))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert sorted(c.tobytes() for c in chunks) == sorted(serial[i].tobytes() for i in range(8))
