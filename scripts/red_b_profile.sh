#!/bin/bash
# Developer tool (GPU box): kernel stats of sum(x,2) at 8192^2 (kernel B + finalize), under `timeout`.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cat > /tmp/red_b_once.py <<PY
import sys
sys.path.insert(0, "$ROOT")
from runmat_amd import HipProvider
prov = HipProvider(0)
a = prov.fill_uniform(1, -1, 1, (8192, 8192))
for _ in range(30): prov.free(prov.reduce_sum_dim(a, 1))
prov.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/redprof -o t -- python /tmp/red_b_once.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/redprof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], "avg", float(r["AverageNs"]) / 1e3, "us min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
