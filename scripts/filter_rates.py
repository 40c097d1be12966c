"""ms per call of imfilter / conv2d on an 8192 x 8192 operand (the numbers DESIGN §3.13 quotes); run on the GPU box:
    python scripts/filter_rates.py            (or under `rocprofv3 --kernel-trace --stats` for the per-kernel durations in profiles/)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from runmat_amd.provider import HipProvider  # noqa: E402

p = HipProvider()
n = 8192
h = p.fill_uniform(3, -1.0, 1.0, (n, n))
sync = p.upload(np.zeros((1, 1)))
rng = np.random.default_rng(1)
print(f"{'call':40s} {'ms':>8s} {'GB/s (one read + one write of the operand)':>44s}")
for ks in ((3, 3), (5, 5), (15, 15)):
    k = p.upload(rng.standard_normal(ks))
    for name, call in ((f"imfilter {ks} replicate same", lambda: p.imfilter(h, k, "replicate")), (f"imfilter {ks} zero same", lambda: p.imfilter(h, k, 0.0)),
                       (f"conv2d {ks} same", lambda: p.conv2d(h, k, "same"))):
        p.free(call())
        p.download(sync)
        t = time.perf_counter()
        for _ in range(10):
            p.free(call())
        p.download(sync)
        ms = (time.perf_counter() - t) / 10 * 1e3
        print(f"{name:40s} {ms:8.3f} {2 * n * n * 8 / ms / 1e6:44.0f}")
