"""Developer tool: host<->device rate of rmhip_upload / rmhip_download (pageable numpy memory, as RunMat hands it over)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
for n in (1024, 4096, 8192):
    a = np.random.default_rng(1).uniform(-1, 1, (n, n))  # row-major: the mirror uploads it as-is and transposes on the device
    nbytes = a.nbytes
    af = np.asfortranarray(a)  # column-major, as RunMat's HostTensorView
    h = prov.upload(a); prov.free(h)
    t0 = time.perf_counter(); hf = prov.upload(af); prov.synchronize(); tf = time.perf_counter() - t0
    assert np.array_equal(prov.download(hf), af.reshape(-1, order="F")); prov.free(hf)
    print(f"  column-major upload {tf*1e3:.2f} ms = {nbytes/tf/1e9:.1f} GB/s")
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); h = prov.upload(a); prov.synchronize(); ts.append(time.perf_counter() - t0); hs = h
        if _ < 3: prov.free(h)
    up = min(ts)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); b = prov.download(hs); ts.append(time.perf_counter() - t0)
    down = min(ts)
    print(f"{n}x{n} f64 ({nbytes/2**20:.0f} MiB): upload {up*1e3:.2f} ms = {nbytes/up/1e9:.1f} GB/s, download {down*1e3:.2f} ms = {nbytes/down/1e9:.1f} GB/s", flush=True)
    prov.free(hs)
