"""Developer tool (GPU box): rates of the order-statistics hooks (order_ops.hip) at 8192^2 f64 and on long vectors - ms per call and GB/s on
the algorithmic bytes (cummin / cummax: 8 read + 16 written per element; diff: 8 + 8; sort: 8 read + 16 written; median: 8 read).
Usage: order_rates.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)


def timed(label, fn, nbytes, reps=5):
    def free(r):
        for h in (r if isinstance(r, (list, tuple)) else [r]):
            if hasattr(h, "values") and hasattr(h.values, "buffer_id"):
                prov.free(h.values); prov.free(h.indices)
            elif hasattr(h, "buffer_id"):
                prov.free(h)
    free(fn())
    prov.synchronize()
    prov.timer_begin()
    for _ in range(reps):
        free(fn())
    t = prov.timer_end() / reps  # ms
    print(f"{label:52s} {t:9.3f} ms  {nbytes/(t*1e-3)/1e9:8.1f} GB/s", flush=True)


n = 8192
h = prov.fill_uniform(5, -1.0, 1.0, (n, n))
e = n * n
for dim in (0, 1):
    timed(f"cummin_scan 8192^2 dim {dim}", lambda: prov.cummin_scan(h, dim), 24 * e)
    timed(f"cummax_scan 8192^2 dim {dim} reverse omitnan", lambda: prov.cummax_scan(h, dim, True, True), 24 * e)
    timed(f"diff_dim 8192^2 dim {dim} (column-major)", lambda: prov.diff_dim(h, 1, dim, True), 16 * e)
    timed(f"diff_dim 8192^2 dim {dim} (reference order)", lambda: prov.diff_dim(h, 1, dim, False), 16 * e)
    timed(f"reduce_median_dim 8192^2 dim {dim}", lambda: prov.reduce_median_dim(h, dim), 8 * e, reps=2)
timed("reduce_median 8192^2 (all)", lambda: prov.reduce_median(h), 8 * e, reps=2)
lib, C = prov._lib, __import__("ctypes")
for dim in (0, 1):
    def sort_device():
        sv, si = C.c_uint64(), C.c_uint64()
        prov._check(lib.rmhip_sort_dim(prov._ctx, h.buffer_id, dim, 0, 0, C.byref(sv), C.byref(si)))
        prov._check(lib.rmhip_free(prov._ctx, sv)); prov._check(lib.rmhip_free(prov._ctx, si))
        return []
    timed(f"sort_dim 8192^2 dim {dim} (device part)", sort_device, 24 * e, reps=2)
def find_device(hx, limit=-1, last=0):
    outs = [C.c_uint64() for _ in range(4)]
    prov._check(lib.rmhip_find(prov._ctx, hx.buffer_id, limit, last, *[C.byref(o) for o in outs]))
    for o in outs: prov._check(lib.rmhip_free(prov._ctx, o))
    return []
dense = prov.elem_gt(h, prov.fill((n, n), 0.0))
sparse = prov.elem_gt(h, prov.fill((n, n), 0.999))
timed("find 8192^2, half of the elements nonzero", lambda: find_device(dense), 8 * e + 32 * e // 2)
timed("find 8192^2, 0.05 % nonzero", lambda: find_device(sparse), 8 * e)
timed("find 8192^2, first 10 (limit)", lambda: find_device(dense, 10), 8 * e)
prov.free(h)
for m in (10**6, 10**7, 10**8):
    v = prov.fill_uniform(6, -1.0, 1.0, (m, 1))
    timed(f"cummin_scan vector {m:.0e}", lambda: prov.cummin_scan(v, 0), 24 * m)
    timed(f"reduce_median vector {m:.0e}", lambda: prov.reduce_median(v), 8 * m, reps=2)
    prov.free(v)
