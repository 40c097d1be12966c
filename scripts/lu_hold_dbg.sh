cd /tmp
R=$GRAFT_REPO_ROOT
for spec in "RMHIP_LU_HOLD=0 RMHIP_LU_SKIP=2" "RMHIP_LU_HOLD=8 RMHIP_LU_SKIP=2" "RMHIP_LU_HOLD=8 RMHIP_LU_SKIP=2 RMHIP_LU_HOLD_TOP_PAD_KB=0" "RMHIP_LU_HOLD=1 RMHIP_LU_SKIP=2" "RMHIP_LU_HOLD=0 RMHIP_LU_SKIP=2 RMHIP_LU_SUPER=0"; do
  echo "== $spec"
  env $spec RMHIP_LU_VERBOSE=1 python $R/scripts/lu_trace.py 16384 3 2>&1 | grep -E "rep=|seat" | sort | uniq -c
done
