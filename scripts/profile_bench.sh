#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: scripts/profile_bench.sh <tag>      (writes under gpurun_out/prof_<tag>/)
# Every rocprofv3 run sits under `timeout` (one hung for ten minutes once and took the GPU budget with it).
# PMC passes use --no-also (one workload per pass): rocprofv3 --pmc crashed on the full default run,
# whose LU workload issues ~50k dispatches.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
for W in fused dgemm; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$W" -o trace -- $BENCH --no-also --workload $W > "$OUT/trace_${W}_bench.json" 2> "$OUT/trace_$W.err"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$W" -o fetch -- $BENCH --no-also --workload $W > "$OUT/fetch_${W}_bench.json" 2> "$OUT/fetch_$W.err"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$W" -o write -- $BENCH --no-also --workload $W > "$OUT/write_${W}_bench.json" 2> "$OUT/write_$W.err"
done
# matrix-pipe evidence for the dgemm kernel (own pass: SQ + GRBM counters only)
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma_dgemm" -o mfma -- $BENCH --no-also --workload dgemm > "$OUT/mfma_dgemm_bench.json" 2> "$OUT/mfma_dgemm.err"
# kernel mix of the LU solve (kernel trace only)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mldivide" -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload mldivide > "$OUT/trace_mldivide_bench.json" 2> "$OUT/trace_mldivide.err"
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -14 "$f" | cut -c1-170; done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
summary = []
for w in ("fused", "dgemm"):
    for tag, counter in ((f"pmc_fetch_{w}", "FETCH_SIZE"), (f"pmc_write_{w}", "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == counter:
                    agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:4]:
            summary.append({"workload": w, "kernel": k, "counter": counter, "launches": len(v), "mean": sum(v)/len(v), "min": min(v), "max": max(v)})
            print(f"{w} {counter} {k[:50]}: n={len(v)} mean={sum(v)/len(v):.1f}")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_mfma_dgemm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_dgemm" in row["Kernel_Name"]:
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    rec = {"workload": "dgemm", "kernel": k, "launches": len(next(iter(d.values())))}
    for cn, v in d.items():
        rec[cn + "_mean"] = sum(v) / len(v)
    if "SQ_INSTS_VALU_MFMA_MOPS_F64_mean" in rec:
        rec["mfma_flops_per_launch"] = rec["SQ_INSTS_VALU_MFMA_MOPS_F64_mean"] * 512
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in rec and "GRBM_GUI_ACTIVE_mean" in rec:
        # GRBM_GUI_ACTIVE comes back either per XCD or summed over the 8 XCDs depending on the rocprofv3 build: decide
        # from the kernel time the same run's bench line reports (cycles ~ 2.4 GHz * t)
        gui = rec["GRBM_GUI_ACTIVE_mean"]
        try:
            t_ms = json.load(open(f"{out}/mfma_dgemm_bench.json"))["roofline"]["kernel_ms"]
            if gui > 4.0 * 2.4e6 * t_ms:
                gui /= 8.0
        except Exception:
            pass
        rec["GRBM_GUI_ACTIVE_per_xcd"] = gui
        rec["mfma_util_pct(busy/(gui_active_per_xcd*1024 SIMDs))"] = 100.0 * rec["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (gui * 1024)
    summary.append(rec)
    print("MFMA", {k2: v2 for k2, v2 in rec.items() if k2 != "kernel"})
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
PY
