import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
for n in (2048, 4096):
    A = np.random.default_rng(0).uniform(-1, 1, (n, n)) + n*np.eye(n)
    ha = prov.upload(A); hb = prov.upload(A @ np.ones((n, 1)))
    ts = []
    for rep in range(8):
        prov.synchronize(); t0 = time.perf_counter()
        hx = prov.mldivide(ha, hb); prov.synchronize()
        ts.append(round((time.perf_counter()-t0)*1e3, 1)); prov.free(hx)
    print(n, "mldivide ms:", ts, prov.telemetry_snapshot()["bytes_allocated"] >> 20, "MiB allocated")
    ts = []
    for rep in range(6):
        prov.synchronize(); t0 = time.perf_counter()
        r = prov.lu(ha); prov.synchronize()
        ts.append(round((time.perf_counter()-t0)*1e3, 1))
        for h in (r.combined, r.lower, r.upper, r.perm_matrix, r.perm_vector): prov.free(h)
    print(n, "lu ms:", ts)
