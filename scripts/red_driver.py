"""Developer tool (GPU box): a fixed sequence of reduction calls at 8192 x 8192 f64 for rocprofv3 passes (kernel stats, FETCH_SIZE /
WRITE_SIZE): sum(x,1), sum(x,2), sum(x,'all'), min / max with indices along both dims, std, nnz, cumsum - `reps` times each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider

prov = HipProvider(0)
n = 8192
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
a = prov.fill_uniform(3, -1.0, 1.0, (n, n))
for _ in range(reps):
    for f in (lambda: prov.reduce_sum_dim(a, 0), lambda: prov.reduce_sum_dim(a, 1), lambda: prov.reduce_sum(a), lambda: prov.reduce_std_dim(a, 0),
              lambda: prov.reduce_nnz_dim(a, 1), lambda: prov.cumsum_scan(a, 0)):
        prov.free(f())
    for dim in (0, 1):
        for fn in (prov.reduce_min_dim, prov.reduce_max_dim):
            r = fn(a, dim)
            prov.free(r.values)
            prov.free(r.indices)
prov.synchronize()
print("ok")
