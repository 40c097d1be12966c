#!/bin/bash
# Developer tool (GPU box): the LU's rank-k update shapes and a deep product (scripts/lu_update_case.py) under sets of RMHIP_GEMM_* knobs.
# Usage: scripts/lu_update_ab.sh "ENV1=.. ENV2=.." "ENV=.." ...      (each argument one environment; "X=1" = defaults)
for E in "$@"; do
  for C in rank512 rank256 rank128 deep "deep 0"; do echo "$E: $(env $E python scripts/lu_update_case.py $C | tail -1)"; done
done
