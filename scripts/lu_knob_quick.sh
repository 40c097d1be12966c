#!/bin/bash
# Developer tool (GPU box): one-line A/B of LU knobs at n = 16384 (ms of the last of 3 solves).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { echo -n "$* : "; env "$@" python $ROOT/scripts/lu_trace.py ${N:-16384} 3 | tail -1; }
run A=0
run RMHIP_LU_NB=512
run RMHIP_LU_NB=128
run RMHIP_LU_LOOKAHEAD=0
run RMHIP_LU_SKIP=1
run RMHIP_LU_SKIP=2
