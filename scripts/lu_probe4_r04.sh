# Developer tool (GPU box): wall-clock A/B of the n = 16384 solve with the update stream's kernel variants (round 4, second pass):
# two eight-wave blocks per CU without the C preload, the sixteen-wave block, with and without the LDS pad.  Usage: lu_probe4_r04.sh
cd /tmp
R=$GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python $R/scripts/lu_trace.py 16384 4 2>&1 | grep "rep=" | tail -2 | awk '{printf "%s ", $3}'; echo; }
run X=1
run RMHIP_GEMM_PRELOAD=0
run RMHIP_GEMM_PRELOAD=0 RMHIP_LU_LA_PAD=0
run RMHIP_GEMM_W16=1
run RMHIP_GEMM_W16=1 RMHIP_LU_LA_PAD=0
run X=1
