"""Match RMHIP_LU_GEMM_LOG shapes (launch order per stream) with the dgemm dispatches of a rocprofv3 kernel trace of the same run
(scripts/lu_gemm_rates.sh): per stream and per (k, kernel) the flops, busy time and rate, and the slowest / fastest launches."""
import collections, csv, glob, re, sys
out, n = sys.argv[1], int(sys.argv[2])
shapes = collections.defaultdict(list)  # stream ptr -> [(m, n, k)]
for ln in open(f"{out}/shapes.txt"):
    m = re.match(r"\[lu_dgemm\] stream (\S+) m (\d+) n (\d+) k (\d+) pad (\d+)", ln)
    if m:
        shapes[m.group(1)].append((int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))))
rows = []
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dgemm" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rmhip::", "")[:40], r.get("Stream_Id", "?")))
rows.sort()
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r[3]].append(r)
# streams are matched by launch COUNT (the log has the host-side order per stream; the trace the device-side order per stream)
cnt_log = {k: len(v) for k, v in shapes.items()}
print("launches per stream, log:", sorted(cnt_log.values()), " trace:", sorted(len(v) for v in by_stream.values()))
for sid, rs in by_stream.items():
    match = [k for k, v in shapes.items() if len(v) == len(rs)]
    if len(match) != 1:
        print(f"stream {sid}: {len(rs)} dgemm dispatches, no unique match in the log"); continue
    sh = shapes[match[0]]
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    per = []
    for (t0, t1, name, _), (m, nn, k, pad) in zip(rs, sh):
        fl, us = 2.0 * m * nn * k, (t1 - t0) / 1e3
        agg[(k, name, pad)][0] += fl; agg[(k, name, pad)][1] += us; agg[(k, name, pad)][2] += 1
        per.append((fl / us / 1e6, m, nn, k, us, name))
    tot_f, tot_us = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
    print(f"== stream {sid}: {len(rs)} updates, {tot_f/1e12:.3f} TFLOP in {tot_us/1e3:.2f} ms busy = {tot_f/tot_us/1e6:.1f} TFLOP/s  (span {(rs[-1][1]-rs[0][0])/1e6:.2f} ms)")
    for (k, name, pad), (fl, us, c) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   k {k:5d} pad {pad:6d} {name:40s} n={c:4d} {fl/1e12:7.3f} TFLOP {us/1e3:8.2f} ms {fl/us/1e6:6.1f} TFLOP/s")
    big = [p for p in per if p[4] > 200]
    if big:
        big.sort()
        print("   slowest large launches:", ", ".join(f"{p[1]}x{p[2]}x{p[3]} {p[0]:.1f}" for p in big[:5]))
        print("   fastest large launches:", ", ".join(f"{p[1]}x{p[2]}x{p[3]} {p[0]:.1f}" for p in big[-5:]))
