#!/bin/bash
# Developer tool (GPU box): A/B of the strided reduction (sum(x,2), 8192^2) over RMHIP_RED_B_MODE / RMHIP_RED_B_BPC.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
cat > /tmp/red_b.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
for shape in ((8192, 8192), (4096, 16384), (65536, 1024)):
    a = prov.fill_uniform(1, -1, 1, shape)
    N = shape[0] * shape[1] * 8.0
    for _ in range(3): prov.free(prov.reduce_sum_dim(a, 1))
    best = 1e9
    for _ in range(4):
        prov.timer_begin()
        for _ in range(20): prov.free(prov.reduce_sum_dim(a, 1))
        best = min(best, prov.timer_end() / 20)
    ref = prov.download(prov.reduce_sum_dim(a, 1))
    print(f"mode={os.environ.get('RMHIP_RED_B_MODE','0')} bpc={os.environ.get('RMHIP_RED_B_BPC','8')} {shape}: {best*1e3:.1f} us  {N/best/1e6:.0f} GB/s  checksum {float(np.sum(ref)):.12e}", flush=True)
    prov.free(a)
PY
for cfg in "0 8" "1 8" "1 4" "1 16" "1 32" "2 8" "2 16"; do
  set -- $cfg
  RMHIP_RED_B_MODE=$1 RMHIP_RED_B_BPC=$2 timeout 120 python /tmp/red_b.py 2>&1 | grep mode=
done
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  RMHIP_RED_B_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/redprof_$m -o t -- python /tmp/red_b.py > /dev/null 2>&1
  echo "== mode $m kernel stats"; cut -d, -f1-4,6,7 $(find /tmp/redprof_$m -name "*kernel_stats.csv") | head -6 | cut -c1-200
done
