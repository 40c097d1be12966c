#!/bin/bash
# Developer tool (GPU box): A/B of environment knob sets on the 16384 (or $N) solve; each argument is "K=V,K=V" (or "-" for none).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "-" ] && envs="A=0"
  r=$(env $envs python $ROOT/scripts/lu_trace.py ${N:-16384} 3 | tail -2 | sed 's/n=.*rep=.: //' | tr '\n' ' ')
  echo "$spec : $r"
done; done
