#!/bin/bash
# Developer tool (GPU box): phase ticks of the solve path's top-block kernel + per-kernel averages of one solve size.
# Usage: scripts/sp_dbg.sh <tag> [n]
set -u
TAG=${1:-dbg}
N=${2:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RMHIP_LU_PANEL_DEBUG=1 timeout 300 python $ROOT/scripts/lu_trace.py $N 2 > "$OUT/panel_debug.log" 2>&1
tail -14 "$OUT/panel_debug.log"
timeout 300 python $ROOT/scripts/lu_trace.py $N 3 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py $N 2 > "$OUT/trace.log" 2> "$OUT/trace.err"
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
find "$OUT/trace" -name "*kernel_trace.csv" -delete
