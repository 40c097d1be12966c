#!/bin/bash
# Developer tool (GPU box): kernel-trace one look-ahead factorisation and report how much of the panel
# kernels' time overlaps dgemm kernels.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/overlap
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RMHIP_LU_LOOKAHEAD=1 RMHIP_LU_NB=${NB:-512} timeout 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/scripts/lu_time.py 16384 > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
# keep the second factorisation only (rep=1): take the last 45% of events by time
t0, t1 = ev[0][0], ev[-1][1]
queues = {}
for s, e, n, q, st in ev:
    key = (q, st)
    queues.setdefault(key, [0, 0.0])
    queues[key][0] += 1; queues[key][1] += (e - s) / 1e6
print("queues/streams:", {k: (v[0], round(v[1], 1)) for k, v in queues.items()})
panels = [(s, e) for s, e, n, q, st in ev if "k_lu_panel" in n]
gemms = [(s, e) for s, e, n, q, st in ev if "k_dgemm" in n and (e - s) > 300000]  # big updates only (>0.3 ms)
tot = sum(e - s for s, e in panels) / 1e6
ov = 0
for ps, pe in panels:
    for gs, ge in gemms:
        lo, hi = max(ps, gs), min(pe, ge)
        if hi > lo: ov += hi - lo
print(f"panel kernels: {len(panels)} total {tot:.1f} ms; overlapped by big dgemm: {ov/1e6:.1f} ms; big dgemms: {len(gemms)} total {sum(e-s for s,e in gemms)/1e6:.1f} ms")
print(f"trace span {(t1-t0)/1e6:.1f} ms")
PY
tail -3 "$OUT/run.log"
