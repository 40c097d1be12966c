#!/bin/bash
# Developer tool (GPU box): A/B of library variants (ab_old/<name>) on small and full-size LU.  Usage: lu_ab.sh name...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for v in "$@"; do
  LIB=$ROOT/ab_old/$v/librmhip.so; [ "$v" = base ] && LIB=$ROOT/runmat_amd/csrc/librmhip.so
  a=$(RMHIP_LIBRARY=$LIB python $ROOT/scripts/lu_small_debug.py 256 4 | tail -1 | sed 's/.*= //')
  b=$(RMHIP_LIBRARY=$LIB python $ROOT/scripts/lu_small_debug.py 1024 4 | tail -1 | sed 's/.*= //')
  c=$(RMHIP_LIBRARY=$LIB python $ROOT/scripts/lu_trace.py 8192 3 | tail -1 | sed 's/.*: //')
  d=$(RMHIP_LIBRARY=$LIB python $ROOT/scripts/lu_trace.py 16384 3 | tail -1 | sed 's/.*: //')
  echo "$v: lu256 $a | lu1024 $b | solve8192 $c | solve16384 $d"
done; done
