"""Developer tool: dump the packed LU factors (and pivots) of a seeded matrix to a .npy; compare two dumps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "dump":
    from runmat_amd import HipProvider
    prov = HipProvider(0)
    n = int(sys.argv[2])
    a = prov.fill_uniform(7, -1, 1, (n, n))
    r = prov.lu(a)
    np.save(sys.argv[3], prov.download(r.combined).reshape(n, n, order="F"))
    np.save(sys.argv[3] + ".piv", prov.download(r.perm_vector))
else:
    x, y = np.load(sys.argv[2]), np.load(sys.argv[3])
    px, py = np.load(sys.argv[2] + ".piv.npy"), np.load(sys.argv[3] + ".piv.npy")
    print("pivots equal:", np.array_equal(px, py), "first pivot diff:", int(np.argmax(px != py)) if not np.array_equal(px, py) else -1)
    neq = x.view(np.uint64) != y.view(np.uint64)
    cols = np.where(neq.any(axis=0))[0]
    print("differing columns:", len(cols), "first:", cols[:8], "max abs diff", float(np.max(np.abs(x - y))))
    if len(cols):
        c = cols[0]; rows = np.where(neq[:, c])[0]
        print("in first differing column", c, ": rows", rows[:10], "count", len(rows))
        print("values", x[rows[:4], c], y[rows[:4], c])
