# Developer tool (GPU box): wall-clock A/B of look-ahead LU knobs and phase-dropping masks (RMHIP_LU_SKIP) at n = 16384, round 4.
cd /tmp
R=$GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python $R/scripts/lu_trace.py 16384 4 2>&1 | grep "rep=" | tail -2 | awk '{printf "%s ", $3}'; echo; }
run X=1
run RMHIP_LU_LATE_XCD=1
run RMHIP_LU_LATE_XCD=1 RMHIP_GEMM_W8P_GRID=224
run RMHIP_LU_LATE_XCD=1 RMHIP_GEMM_W8P_GRID=192
run RMHIP_LU_LATE_XCD=1 RMHIP_GEMM_W8P_GRID=160
run RMHIP_LU_LATE_XCD=1 RMHIP_GEMM_W8P_GRID=128
run RMHIP_LU_LATE_XCD=1 RMHIP_GEMM_W8P_GRID=192 RMHIP_LU_NB_LATE=256
run RMHIP_LU_LATE_XCD=2 RMHIP_GEMM_W8P_GRID=224
run X=1
