#!/bin/bash
# Developer tool (GPU box): A/B of the look-ahead driver's knobs on the solve path.  Usage: scripts/sp_sweep.sh [n]
N=${1:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
run() { echo -n "$* : "; env "$@" python $ROOT/scripts/lu_trace.py $N 4 2>&1 | grep "rep=" | tail -2 | awk '{printf "%s ", $3}'; echo; }
run X=1
run RMHIP_LU_NB_LATE=64
run RMHIP_LU_NB_LATE=256
run RMHIP_LU_NB_LATE=512
run RMHIP_LU_LA_PAD=0
run RMHIP_LU_PANEL_PAD_KB=0
run RMHIP_LU_SPLIT_ROWS=2048
run RMHIP_LU_SPLIT=0
run RMHIP_LU_CHAIN_PRIO=0
run X=1
