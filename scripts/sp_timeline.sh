#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace timeline of the solve at one size.  Usage: scripts/sp_timeline.sh <tag> [n]
set -u
TAG=${1:-tl}
N=${2:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/scripts/lu_trace.py $N 4 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py $N 3 > "$OUT/trace.log" 2> "$OUT/trace.err"
python $ROOT/scripts/lu_timeline.py "$OUT" > "$OUT/timeline.txt" 2>&1
grep -n "== solve" "$OUT/timeline.txt" | head
awk '/== solve 3/,/== solve 4/' "$OUT/timeline.txt" | head -60
awk '/== last solve/,0' "$OUT/timeline.txt" | head -80
find "$OUT/trace" -name "*kernel_trace.csv" -size +20M -delete
