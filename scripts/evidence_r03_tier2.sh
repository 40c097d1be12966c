#!/bin/bash
# Developer tool (GPU box): the grid / rate scripts behind the second-tier numbers of DESIGN.md section 7 and 8, in one call
# (outputs under gpurun_out/prof_r03_tier2/, copied to profiles/r03_*.txt by hand).  Usage: scripts/evidence_r03_tier2.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r03_tier2
mkdir -p "$OUT"
cd $ROOT
timeout 120 python scripts/misc_grid.py > "$OUT/misc_grid.txt" 2>&1
timeout 60 python scripts/cov_rates.py > "$OUT/cov_rates.txt" 2>&1
timeout 120 python scripts/lsq_rates.py > "$OUT/lsq_rates.txt" 2>&1
timeout 120 python scripts/solve_small.py > "$OUT/solve_small.txt" 2>&1
timeout 120 python scripts/ew_grid.py > "$OUT/ew_grid.txt" 2>&1
timeout 200 python scripts/red_grid.py > "$OUT/red_grid.txt" 2>&1
timeout 120 python scripts/svd_sizes.py 512 1024 2048 > "$OUT/svd_sizes.txt" 2>&1
./scripts/micro/pow_accuracy 3000000 > "$OUT/pow_accuracy.txt" 2>&1
tail -n 3 "$OUT/cov_rates.txt"; tail -n 3 "$OUT/solve_small.txt"
