#!/bin/bash
# Developer tool (GPU box): first look at the solve path.  Usage: scripts/solvepath_run.sh <tag>
set -u
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/scripts/solvepath_check.py quick > "$OUT/quick.log" 2>&1
tail -20 "$OUT/quick.log"
timeout 600 python $ROOT/scripts/solvepath_check.py time > "$OUT/time.log" 2>&1
cat "$OUT/time.log"
RMHIP_LU_PANEL_DEBUG=1 timeout 300 python $ROOT/scripts/lu_trace.py 16384 2 > "$OUT/panel_debug.log" 2>&1
tail -16 "$OUT/panel_debug.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py 16384 3 > "$OUT/trace.log" 2> "$OUT/trace.err"
python $ROOT/scripts/lu_timeline.py "$OUT" > "$OUT/timeline.txt" 2>&1
head -70 "$OUT/timeline.txt"
find "$OUT/trace" -name "*kernel_trace.csv" -size +20M -delete
