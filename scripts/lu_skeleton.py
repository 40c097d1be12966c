"""Developer tool: the schedule skeleton of the last solve in a lu_super_trace.sh directory: every kernel longer than a threshold (any
stream) and every idle gap of the main stream longer than a threshold, in time order.  Usage: lu_skeleton.py <dir> [kernel_us] [gap_us]"""
import csv, sys, glob, collections
out = sys.argv[1]; kmin = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0; gmin = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
rows = []
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")[:30], r["Stream_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
s0 = [r[0] for r in rows if "copyBufferRect" in r[2]][-1]
last = [r for r in rows if r[0] >= s0]
main = collections.Counter(r[3] for r in last).most_common(1)[0][0]
ev = []
for r in last:
    if (r[1] - r[0]) / 1e3 >= kmin: ev.append((r[0], f"stream {r[3]} {r[2]:30s} blocks {r[4]:6d}  {(r[0]-s0)/1e6:7.2f} -> {(r[1]-s0)/1e6:7.2f} ms ({(r[1]-r[0])/1e6:.2f})"))
mr = [r for r in last if r[3] == main]
for a, b in zip(mr, mr[1:]):
    if (b[0] - a[1]) / 1e3 >= gmin: ev.append((a[1], f"   main idle {(a[1]-s0)/1e6:7.2f} -> {(b[0]-s0)/1e6:7.2f} ms ({(b[0]-a[1])/1e6:.2f}) before {b[2]}"))
for _, line in sorted(ev): print(line)
