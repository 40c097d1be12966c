"""Developer tool: print kernel name / calls / average us from rocprofv3 *kernel_stats.csv files under a directory (optionally filtered)."""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.reader(open(f)))[1:]:
        if pat in r[0]:
            print(f"  {r[0][:56]:56s} calls {r[1]:>5s}  avg {float(r[3]) / 1e3:9.1f} us")
