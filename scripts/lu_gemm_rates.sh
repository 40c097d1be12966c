#!/bin/bash
# Developer tool (GPU box): per-launch rate of every trailing update inside ONE x = A\b solve: the shapes from RMHIP_LU_GEMM_LOG (launch
# order, by stream) matched against the rocprofv3 kernel trace of the same run.  Usage: scripts/lu_gemm_rates.sh [n] [extra env ...]
N=${1:-16384}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lu_gemm_rates
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
env "$@" RMHIP_LU_GEMM_LOG=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/lu_trace.py $N 2 > "$OUT/log.txt" 2> "$OUT/shapes.txt"
python $ROOT/scripts/lu_gemm_rates.py "$OUT" $N
