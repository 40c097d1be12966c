#!/bin/bash
# Developer tool (GPU box): wall-clock A/B of the two-level LU driver's knobs at n = 16384 (or $N); each argument is "K=V,K=V" ("-" = defaults).
cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "-" ] && envs="A=0"
  r=$(env $envs python $R/scripts/lu_trace.py ${N:-16384} ${REPS:-4} 2>&1 | grep -E "rep=|max" | sed 's/n=.*rep=.: //' | awk '{printf "%s ", $0}')
  echo "$spec : $r"
done
