#!/bin/bash
# Developer tool (GPU box): one rocprofv3 kernel trace of two solves -> per-stream timeline (lu_timeline.py) and per-launch update rates
# (lu_gemm_rates.py).  Usage: scripts/lu_super_trace.sh <tag> [n] [K=V ...]
TAG=${1:-st}; N=${2:-16384}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lu_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
env "$@" RMHIP_LU_GEMM_LOG=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py $N 2 > "$OUT/log.txt" 2> "$OUT/shapes.txt"
python $ROOT/scripts/lu_gemm_rates.py "$OUT" $N > "$OUT/gemm_rates.txt" 2>&1
python $ROOT/scripts/lu_timeline.py "$OUT" > "$OUT/timeline.txt" 2>&1
find "$OUT/trace" -name "*kernel_trace.csv" -size +30M -delete
