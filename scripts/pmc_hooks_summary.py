"""Summarise scripts/profile_r04_hooks.sh: per mode (lazy / eager repmat) and kernel the launches, average duration and the
HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters in KiB; gfx950 FETCH_SIZE counts half of a wide
streaming read - MI355X_MICROARCH.md HBM section, as scripts/pmc_summary.py)."""
import collections, csv, glob, sys

out = sys.argv[1]
for mode in ("lazy", "eager"):
    def counters(tag, name):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{tag}_{mode}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == name:
                    agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        return agg
    fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
    dur = {}
    for f in glob.glob(f"{out}/trace_{mode}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    print(f"== RMHIP_EAGER_REPMAT={'1' if mode == 'eager' else '0'} ({mode})")
    print(f"{'kernel':70s} {'calls':>5s} {'avg us':>9s} {'fetch MiB':>10s} {'write MiB':>10s} {'HBM MiB':>9s}")
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        f_mib, w_mib = 2 * sum(fv) / len(fv) / 1024, sum(wv) / len(wv) / 1024
        calls, avg = dur.get(k, (len(fv), float("nan")))
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")[:70]
        print(f"{short:70s} {calls:5d} {avg:9.1f} {f_mib:10.1f} {w_mib:10.1f} {f_mib + w_mib:9.1f}")
