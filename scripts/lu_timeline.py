"""Summarise a rocprofv3 kernel trace of lu_trace.py: per solve, per queue (stream) busy time by kernel, the
overlap between the two streams and the idle gaps of the main stream.  Usage: lu_timeline.py <dir with trace/>"""
import collections, csv, glob, sys
out = sys.argv[1]
files = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?"),
                     r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    print("no trace rows"); sys.exit(0)
# split into solves at k_gather_rows (first kernel of lu_solve_device follows the factorisation): use big idle gaps instead
solves, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(r[1] for r in cur[-50:]) > 3_000_000:  # > 3 ms idle: host work between solves
        solves.append(cur); cur = []
    cur.append(b)
solves.append(cur)
def short(n):
    for k in ("k_dgemm", "k_lu_panel", "k_rp_below", "k_laswp_lists", "k_trsm_fused", "k_trsm_lower_2p", "k_subst_chain", "k_build_plist", "k_rhs_update", "k_lu_col", "k_gather_rows"):
        if k in n: return k
    return n[:30]
for si, s in enumerate(solves):
    if len(s) < 100: continue
    t0, t1 = s[0][0], max(r[1] for r in s)
    print(f"== solve {si}: {len(s)} kernels, span {(t1-t0)/1e6:.2f} ms")
    by_q = collections.defaultdict(list)
    for r in s: by_q[(r[3], r[4])].append(r)
    for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[1] - r[0] for r in rs)
        agg = collections.defaultdict(lambda: [0, 0])
        for r in rs:
            agg[short(r[2])][0] += 1; agg[short(r[2])][1] += r[1] - r[0]
        gaps = sum(max(0, b[0] - a[1]) for a, b in zip(rs, rs[1:]))
        print(f"  queue/stream {q}: {len(rs)} kernels, busy {busy/1e6:.2f} ms, gaps {gaps/1e6:.2f} ms, first {(rs[0][0]-t0)/1e6:.2f} last {(rs[-1][1]-t0)/1e6:.2f}")
        for k, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"      {k:18s} n={cnt:5d} total {ns/1e6:8.3f} ms  avg {ns/cnt/1e3:8.1f} us")
    # dgemm by k-size is not in the trace; bucket dgemm by duration instead
    dg = sorted(r[1] - r[0] for r in s if "k_dgemm" in r[2])
    if dg:
        print("  dgemm duration buckets (us): " + ", ".join(f"<{b}:{sum(1 for d in dg if lo*1e3 <= d < b*1e3)}" for lo, b in ((0, 30), (30, 60), (60, 120), (120, 250), (250, 500), (500, 1000), (1000, 2000), (2000, 1e9))))
    # timeline in 20 slices: per slice, fraction of time each stream is busy
    nsl = 16
    print("  slice  " + "  ".join(f"{q[1]:>10}" for q in sorted(by_q)))
    for i in range(nsl):
        a, b = t0 + (t1 - t0) * i // nsl, t0 + (t1 - t0) * (i + 1) // nsl
        cells = []
        for q in sorted(by_q):
            busy = sum(max(0, min(r[1], b) - max(r[0], a)) for r in by_q[q])
            cells.append(f"{100.0*busy/(b-a):9.0f}%")
        print(f"  {i:5d}  " + "  ".join(cells))

# ---- last solve (from its copy of A into the workspace on): 5 ms slices, per stream the busy share by kernel
starts = [r[0] for r in rows if "copyBufferRect" in r[2]]
if starts:
    s0 = starts[-1]
    last = [r for r in rows if r[0] >= s0]
    qs = collections.Counter((r[3], r[4]) for r in last)
    end = max(r[1] for r in last)
    print(f"== last solve: {len(last)} kernels, span {(end - s0)/1e6:.2f} ms")
    T = 5_000_000
    for q, cnt in qs.most_common():
        if cnt < 20: continue
        print(f"  stream {q} ({cnt} kernels)")
        for i in range(int((end - s0) / T) + 1):
            a, b = s0 + i * T, s0 + (i + 1) * T
            acc = collections.Counter()
            for r in last:
                if (r[3], r[4]) != q: continue
                o = min(r[1], b) - max(r[0], a)
                if o > 0: acc[short(r[2])] += o
            print(f"    {i*5:3d} ms busy {100.0*sum(acc.values())/T:5.1f}%  " + " ".join(f"{k}:{100.0*v/T:.0f}" for k, v in acc.most_common()))
