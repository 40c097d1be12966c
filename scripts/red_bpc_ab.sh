#!/bin/bash
# Developer tool (GPU box): sum(x,2) against kernel B's target blocks per CU (RMHIP_RED_B_BPC) on several shapes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
cat > /tmp/red_bpc.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from runmat_amd import HipProvider
prov = HipProvider(0)
out = []
for shape in ((8192, 8192), (16384, 4096), (65536, 1024), (4096, 16384), (8192, 8000), (7936, 8192), (2048, 32768), (32768, 2048)):
    a = prov.fill_uniform(1, -1, 1, shape)
    N = shape[0] * shape[1] * 8.0
    for _ in range(3): prov.free(prov.reduce_sum_dim(a, 1))
    best = 1e9
    for _ in range(4):
        prov.timer_begin()
        for _ in range(20): prov.free(prov.reduce_sum_dim(a, 1))
        best = min(best, prov.timer_end() / 20)
    out.append(f"{best*1e3:6.1f}")
    prov.free(a)
print(" ".join(out))
PY
echo "shapes: 8192^2 16384x4096 65536x1024 4096x16384 8192x8000 7936x8192 2048x32768 32768x2048 (us)"
for b in 1 2 3 4 5 6; do echo -n "bpc=$b: "; RMHIP_RED_B_BPC=$b timeout 100 python /tmp/red_bpc.py 2>&1 | tail -1; done
