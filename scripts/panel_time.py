"""Developer tool: time the base-panel factorisation alone (blk_lu on a rows x 64 view)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
for rows in [int(a) for a in (sys.argv[1:] or ["256", "1024", "2048", "4096", "8192", "16384"])]:
    ts = []
    for rep in range(6):
        buf = prov.fill_uniform(5 + rep, -1, 1, (rows, 64))
        prov.synchronize(); t0 = time.perf_counter()
        perm, info = prov.blk_lu((buf, 0, 0, rows, 64))
        prov.synchronize(); ts.append(time.perf_counter() - t0)
        prov.free(buf); prov.free(perm)
    print(f"rows={rows}: best {min(ts)*1e6:.0f} us  median {sorted(ts)[3]*1e6:.0f} us per 64-column panel ({min(ts)*1e6/64:.2f} us/col)", flush=True)
