#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats + HBM traffic counters for the precision-32 workloads of bench.py.
# Usage: scripts/profile_f32.sh <tag>      (writes under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01f32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also"
for W in fused_f32 sgemm; do
  STEPS=20; [ $W = fused_f32 ] && STEPS=200   # 200 back-to-back dispatches: the per-dispatch trace below shows drift, if any
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$W" -o trace -- ${BENCH/--steps 20/--steps $STEPS} --workload $W > "$OUT/trace_${W}_bench.json" 2> "$OUT/trace_$W.err"
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_fused_f32" -o fetch -- $BENCH --workload fused_f32 > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_fused_f32" -o write -- $BENCH --workload fused_f32 > "$OUT/write_bench.json" 2> "$OUT/write.err"
for f in $(find "$OUT" -name "*kernel_stats.csv"); do echo "== $f"; head -5 "$f" | cut -c1-170; done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summary = []
for tag, counter in (("pmc_fetch_fused_f32", "FETCH_SIZE"), ("pmc_write_fused_f32", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:3]:
        summary.append({"workload": "fused_f32", "kernel": k, "counter": counter, "launches": len(v), "mean": sum(v)/len(v), "min": min(v), "max": max(v)})
        print(f"fused_f32 {counter} {k[:50]}: n={len(v)} mean={sum(v)/len(v):.1f}")
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
# per-dispatch durations of the f32 fused kernel, in launch order
rows = []
for f in glob.glob(f"{out}/trace_fused_f32/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("rm_ew_fast"):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
rows.sort()
with open(f"{out}/fused_f32_dispatches.csv", "w") as fh:
    fh.write("dispatch,start_us_since_first,duration_us\n")
    for i, (t, d) in enumerate(rows):
        fh.write(f"{i},{(t - rows[0][0]) / 1e3:.1f},{d / 1e3:.2f}\n")
if rows:
    ds = [d / 1e3 for _, d in rows]
    print(f"fused_f32 dispatches: n={len(ds)} first={ds[0]:.1f} min={min(ds):.1f} max={max(ds):.1f} mean={sum(ds)/len(ds):.1f} last={ds[-1]:.1f} us")
PY
