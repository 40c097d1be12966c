#!/bin/bash
# Developer tool (GPU box): rocprofv3 evidence for the repmat view - kernel stats and SEPARATE FETCH_SIZE / WRITE_SIZE passes of
# scripts/hooks_driver.py, once with the lazy view (default) and once with RMHIP_EAGER_REPMAT=1.  Usage: scripts/profile_r04_hooks.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r04_hooks
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $ROOT/scripts/hooks_driver.py 10 > "$OUT/rates_lazy.txt" 2>&1
RMHIP_EAGER_REPMAT=1 python $ROOT/scripts/hooks_driver.py 10 > "$OUT/rates_eager.txt" 2>&1
for MODE in lazy eager; do
  E=0; [ $MODE = eager ] && E=1
  CMD="env RMHIP_EAGER_REPMAT=$E python $ROOT/scripts/hooks_driver.py 4"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$MODE" -o trace -- $CMD > /dev/null 2> "$OUT/trace_$MODE.err"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$MODE" -o fetch -- $CMD > /dev/null 2> "$OUT/fetch_$MODE.err"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$MODE" -o write -- $CMD > /dev/null 2> "$OUT/write_$MODE.err"
  find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
done
python $ROOT/scripts/pmc_hooks_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/rates_lazy.txt" "$OUT/rates_eager.txt" "$OUT/summary.txt"
