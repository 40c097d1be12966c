#!/bin/bash
# Developer tool (GPU box): everything profiles/r03_* is made from, in one call.  Usage: scripts/evidence_r03.sh
# (then, in the build container: python scripts/collect_profiles.py r03)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p "$OUT"
bash $ROOT/scripts/profile_r03.sh r03 > "$OUT/profile.log" 2>&1
bash $ROOT/scripts/sp_timeline.sh r03 > "$OUT/timeline.log" 2>&1
cp $ROOT/gpurun_out/sp_r03/timeline.txt "$OUT/mldivide_timeline.txt"
cd /tmp
python $ROOT/scripts/tier2_rates.py > "$OUT/tier2_rates.txt" 2>&1
python $ROOT/scripts/red2_rates.py > "$OUT/red2_rates.txt" 2>&1
python $ROOT/scripts/red_shapes.py > "$OUT/red_shapes.txt" 2>&1
python $ROOT/scripts/gemm_variants.py > "$OUT/gemm_variants.txt" 2>&1
python $ROOT/scripts/rng_accuracy.py 2000000 > "$OUT/rng_accuracy.txt" 2>&1
for n in 1024 2048 4096 6144 8192 12288 16384; do python $ROOT/scripts/lu_trace.py $n 4 2>&1 | grep "rep=" | tail -1; done > "$OUT/solve_sizes.txt"
cd $ROOT
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 600 tests/tools/offload_calibrate > "$OUT/offload_calibration.json" 2> "$OUT/calib.err"
tail -3 "$OUT/solve_sizes.txt"; tail -2 "$OUT/tier2_rates.txt"; head -c 300 "$OUT/bench_default.json"
