#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: scripts/profile_bench.sh <tag>      (writes under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/write_bench.json" 2> "$OUT/write.err"
find "$OUT" -name "*.csv" | head -30
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                agg[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"{counter} {k}: n={len(v)} mean={sum(v)/len(v):.1f} max={max(v):.1f}")
PY
