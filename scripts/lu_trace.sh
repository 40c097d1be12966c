#!/bin/bash
# Developer tool (GPU box): kernel timeline of the 16384 solve.  Usage: scripts/lu_trace.sh <tag> [n]
set -u
TAG=${1:-r02}
N=${2:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lu_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $ROOT/scripts/lu_trace.py $N 4 > "$OUT/plain.log" 2>&1
RMHIP_LU_PANEL_DEBUG=1 python $ROOT/scripts/lu_trace.py $N 2 > "$OUT/panel_debug.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py $N 3 > "$OUT/trace.log" 2> "$OUT/trace.err"
python $ROOT/scripts/lu_timeline.py "$OUT" > "$OUT/timeline.txt" 2>&1
cat "$OUT/plain.log" "$OUT/panel_debug.log" "$OUT/trace.log"
cat "$OUT/timeline.txt" | head -80
# keep the merge-back small: stats + the summarised timeline, not the raw trace of every run
find "$OUT/trace" -name "*kernel_trace.csv" -size +20M -delete
