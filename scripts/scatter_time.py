"""Developer tool (GPU box): wall clock of rmhip_scatter_linear at three sizes (set RMHIP_SCATTER_DEVICE_MIN=0 for the host path)."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from runmat_amd import HipProvider
prov = HipProvider(0)
for numel, n in ((10**8, 10**7), (10**7, 10**6), (10**6, 10**5)):
    t = prov.zeros((numel, 1))
    rng = np.random.default_rng(1)
    idx = rng.integers(0, numel, n).astype(np.uint32)
    v = prov.upload(rng.standard_normal(n))
    prov.scatter_linear(t, idx, v)
    t0 = time.perf_counter()
    for _ in range(3): prov.scatter_linear(t, idx, v)
    print(f"numel {numel} n {n}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per call", flush=True)
    prov.free(t); prov.free(v)
