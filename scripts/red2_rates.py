"""Developer tool (GPU box): rates of the round-3 reduction hooks at 8192 x 8192 f64 (512 MiB in; scans also write 512 MiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider

prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = prov.fill_uniform(3, -1.0, 1.0, (n, n))
N = n * n * 8.0


def rate(tag, f, nbytes, reps=12):
    def free(r):
        for h in (r if isinstance(r, tuple) else (r,)):
            prov.free(h)
    for _ in range(3):
        free(f())
    prov.timer_begin()
    for _ in range(reps):
        free(f())
    ms = prov.timer_end() / reps
    print(f"{tag:34s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:7.0f} GB/s", flush=True)


def mm(fn, dim):
    r = fn(a, dim)
    return (r.values, r.indices)


for dim in (0, 1):
    rate(f"reduce_min_dim dim{dim}", lambda: mm(prov.reduce_min_dim, dim), N)
    rate(f"reduce_max_dim dim{dim}", lambda: mm(prov.reduce_max_dim, dim), N)
    rate(f"reduce_sum_dim dim{dim} (reference)", lambda: prov.reduce_sum_dim(a, dim), N)
    rate(f"reduce_std_dim dim{dim}", lambda: prov.reduce_std_dim(a, dim), N)
    rate(f"reduce_nnz_dim dim{dim}", lambda: prov.reduce_nnz_dim(a, dim), N)
    rate(f"cumsum_scan dim{dim}", lambda: prov.cumsum_scan(a, dim), 2 * N)
rate("reduce_std all", lambda: prov.reduce_std(a), N)
rate("reduce_any all", lambda: prov.reduce_any(a), N)
v = prov.fill_uniform(4, -1.0, 1.0, (n * n, 1))
rate("cumsum_scan 6.7e7 vector", lambda: prov.cumsum_scan(v, 0), 2 * N)
rate("reduce_min_dim 6.7e7 vector", lambda: mm(prov.reduce_min_dim, 0) if False else (lambda r: (r.values, r.indices))(prov.reduce_min_dim(v, 0)), N)
prov.free(a); prov.free(v)
# ragged shapes: rows just above a multiple of 512 (the window count along the rows is rounded to a multiple of the XCD count)
for shape in ((8200, 8192), (5000, 13000)):
    a = prov.fill_uniform(5, -1.0, 1.0, shape)
    Ns = shape[0] * shape[1] * 8.0
    rate(f"reduce_sum_dim dim1 {shape}", lambda: prov.reduce_sum_dim(a, 1), Ns)
    rate(f"reduce_min_dim dim1 {shape}", lambda: mm(prov.reduce_min_dim, 1), Ns)
    rate(f"reduce_std_dim dim1 {shape}", lambda: prov.reduce_std_dim(a, 1), Ns)
    prov.free(a)
