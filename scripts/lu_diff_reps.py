"""Developer tool (GPU box): factor the same matrix repeatedly and locate where a run's factors differ from the majority.
Usage: lu_diff_reps.py [n] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a = prov.fill_uniform(41, -1, 1, (n, n))
ref = None; refp = None
for rep in range(reps):
    r = prov.lu(a)
    comb = prov.download(r.combined).reshape((n, n), order="F")
    piv = prov.download(r.perm_vector).ravel()
    for h in (r.combined, r.lower, r.upper, r.perm_matrix, r.perm_vector): prov.free(h)
    if ref is None:
        ref, refp = comb, piv
        continue
    if np.array_equal(comb, ref):
        print(f"rep {rep}: identical", flush=True); continue
    d = comb != ref
    cols = np.nonzero(d.any(axis=0))[0]; rows = np.nonzero(d.any(axis=1))[0]
    pd = np.nonzero(piv != refp)[0]
    c0 = cols[0]
    rr = np.nonzero(d[:, c0])[0]
    print(f"rep {rep}: DIFF cols {cols[0]}..{cols[-1]} ({cols.size}), rows {rows[0]}..{rows[-1]} ({rows.size}); first pivot diff at {pd[0] if pd.size else -1};"
          f" in first differing col {c0}: rows {rr[0]}..{rr[-1]} ({rr.size}); max |d| {np.abs(comb-ref).max():.3e}", flush=True)
    for rix in rows[:6]:
        cc = np.nonzero(d[rix])[0]
        runs = []
        start = prev = cc[0]
        for x in cc[1:]:
            if x != prev + 1: runs.append((start, prev)); start = x
            prev = x
        runs.append((start, prev))
        print(f"   row {rix}: {cc.size} cols, runs {runs[:8]}{'...' if len(runs) > 8 else ''}; max |d| {np.abs(comb[rix]-ref[rix]).max():.3e}", flush=True)
    continue
    # per 128-column block: number of differing entries, first 40 blocks with any
    blk = [(cb, int(d[:, cb:cb+128].sum())) for cb in range(0, n, 128)]
    print("   blocks(col0:count) " + " ".join(f"{cb}:{cnt}" for cb, cnt in blk if cnt)[:600], flush=True)
