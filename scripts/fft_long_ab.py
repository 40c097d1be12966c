"""Developer tool (GPU box): time fft_dim on one long vector per power of two - used to place the two-pass / three-pass crossover (fft.hip)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from runmat_amd.provider import HipProvider
p = HipProvider()
sync = p.upload(np.zeros((1, 1)))
for lg in (19, 20, 21, 22, 23, 24):
    v = p.fill_uniform(6, -1.0, 1.0, (1 << lg, 1))
    out = p.fft_dim(v, None, 0); p.download(sync); p.free(out)
    t = time.perf_counter()
    for _ in range(20):
        p.free(p.fft_dim(v, None, 0))
    p.download(sync)
    print(lg, f"{(time.perf_counter() - t) / 20 * 1e3:.4f} ms")
    p.free(v)
