"""Developer tool: per-kernel statistics of the MAIN stream of the last solve in a lu_super_trace.sh output directory, in 10 ms windows
(count, mean duration incl. the wait for its slot) - what the panel chain spends where.  Usage: lu_main_stats.py <dir> [window_ms]"""
import csv, sys, collections, glob
out = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rows = []
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")[:28], r["Stream_Id"]))
rows.sort()
s0 = [r[0] for r in rows if "copyBufferRect" in r[2]][-1]
last = [r for r in rows if r[0] >= s0]
cnt = collections.Counter(r[3] for r in last)
main = cnt.most_common(1)[0][0]
end = max(r[1] for r in last)
print(f"span {(end - s0) / 1e6:.2f} ms; main stream {main} ({cnt[main]} kernels)")
names = ["k_rp_top<false>", "k_rp_below_mfma", "k_laswp_lists", "k_trsm_lower_mfma<4>", "k_trsm_lower_mfma<8>", "k_dgemm_small", "k_dgemm<"]
print("window      " + "".join(f"{n[:20]:>22s}" for n in names) + "   gaps")
nw = int((end - s0) / 1e6 / win) + 1
prev_end = {}
for w in range(nw):
    a, b = s0 + w * win * 1e6, s0 + (w + 1) * win * 1e6
    cells = []
    mr = [r for r in last if r[3] == main and a <= r[0] < b]
    for n in names:
        sel = [r for r in mr if r[2].startswith(n)]
        cells.append(f"{len(sel):5d} x {sum(r[1] - r[0] for r in sel) / max(1, len(sel)) / 1e3:6.1f} us" if sel else " " * 17)
    allm = [r for r in last if r[3] == main]
    gaps = sum(max(0, y[0] - x[1]) for x, y in zip(allm, allm[1:]) if a <= y[0] < b)
    print(f"{w * win:5.0f} ms  " + "".join(f"{c:>22s}" for c in cells) + f"  {gaps / 1e6:6.2f} ms")
