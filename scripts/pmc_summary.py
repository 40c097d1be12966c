"""Summarise scripts/profile_r03.sh's passes: per workload and kernel the launches, the average duration (kernel-trace pass) and the HBM-side
bytes per launch from the FETCH_SIZE / WRITE_SIZE passes - bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on
gfx950 FETCH_SIZE reports half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section; WRITE_SIZE calibrates exactly on a
512 MiB output) - and per workload the bytes per bench step.  Writes <dir>/pmc_summary.json and <dir>/pmc_traffic.json."""
import collections, csv, glob, json, sys

out = sys.argv[1]
workloads = sys.argv[2:]
LEGS = {"fused": 2, "dgemm": 2, "sgemm": 2, "fused_f32": 2, "chain": 2, "bcast": 2, "fft": 2}  # workloads whose record runs its steps twice (wall-clock leg + HIP-event leg)
SETUP = ("k_fill", "k_probe_xcc", "__amd_rocclr", "k_narrow", "k_widen")  # not part of a step
SETUP_BY_WORKLOAD = {"mldivide": ("k_fill", "k_probe_xcc", "__amd_rocclr_fillBuffer", "__amd_rocclr_copyBuffer(", "__amd_rocclr_copyBufferAligned")}  # the rect copy of A into the padded workspace IS part of a solve


def counters(tag, name):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == name:
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return agg


def durations(tag):
    d = {}
    for f in glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            d[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    return d


summary, traffic = [], {}
for w in workloads:
    fetch, write, dur = counters(f"pmc_fetch_{w}", "FETCH_SIZE"), counters(f"pmc_write_{w}", "WRITE_SIZE"), durations(f"trace_{w}")
    steps = None
    try:
        line = [ln for ln in open(f"{out}/fetch_{w}_bench.json") if ln.startswith("{")][-1]
        j = json.loads(line)
        steps = (j["steps"] * LEGS.get(w, 1) + j["warmup"])
    except Exception:
        pass
    total = 0.0
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * sum(fetch.get(k, [0])) + sum(write.get(k, [0])))):
        fv, wv = fetch.get(k, []), write.get(k, [])
        n = max(len(fv), len(wv))
        if n == 0:
            continue
        bytes_per = (2 * (sum(fv) / len(fv) if fv else 0.0) + (sum(wv) / len(wv) if wv else 0.0)) * 1024
        calls, avg_us = dur.get(k, (n, float("nan")))
        rec = {"workload": w, "kernel": k[:120], "launches": n, "fetch_kib_mean": sum(fv) / len(fv) if fv else None,
               "write_kib_mean": sum(wv) / len(wv) if wv else None, "bytes_per_launch": round(bytes_per), "avg_us": round(avg_us, 2),
               "tb_per_s": round(bytes_per / avg_us / 1e6, 3) if avg_us == avg_us and avg_us > 0 else None}
        summary.append(rec)
        if not any(k.startswith(s) or s in k[:40] for s in SETUP_BY_WORKLOAD.get(w, SETUP)):
            total += bytes_per * n
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")
        traffic.setdefault(w, {})[short[:80]] = round(bytes_per)
    if steps:
        traffic.setdefault(w, {})["_bytes_per_step"] = round(total / steps)
        traffic[w]["_steps_in_run"] = steps
    # matrix pipe
    for f in glob.glob(f"{out}/pmc_mfma_{w}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            if "gemm" in row["Kernel_Name"]:
                agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in agg.items():
            rec = {"workload": w, "kernel": k[:120], "launches": len(next(iter(d.values())))}
            for cn, v in d.items():
                rec[cn + "_mean"] = sum(v) / len(v)
            summary.append(rec)
# VALU pass (scripts/profile_r04.sh): wave-level VALU instructions per launch and the share of the SIMDs' cycles they keep busy.
# SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves (MI355X_MICROARCH.md, counter units) - one per issued VALU instruction, whatever
# its execution rate - and GRBM_GUI_ACTIVE comes back summed over the eight XCDs; a SIMD issues one VALU instruction at a time, so the
# issue-slot share is 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).  (Under the profiler the clock sits near 1.9 GHz and
# GUI_ACTIVE includes the dispatch ramp, so bench.py's VALU roofline uses instructions / the launch time IT measures instead.)
valu = {"_durations_us": {}}
for w in workloads:
    valu["_durations_us"][w] = {k.replace("(anonymous namespace)::", "").split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")[:80]: round(v[1], 2)
                                for k, v in durations(f"trace_{w}").items()}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_valu_{w}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if not d.get("SQ_INSTS_VALU"):
            continue
        mean = {cn: sum(v) / len(v) for cn, v in d.items()}
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void rmhip::", "").replace("rmhip::", "")[:80]
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
        rec = {"launches": len(d["SQ_INSTS_VALU"]), "insts_valu_per_launch": round(mean["SQ_INSTS_VALU"]),
               "active_inst_valu_quadcycles": round(mean.get("SQ_ACTIVE_INST_VALU", 0.0)), "wave_quadcycles": round(mean.get("SQ_WAVE_CYCLES", 0.0)),
               "gui_active_cycles": round(gui),
               "valu_busy": round(4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * gui / 8.0), 4) if gui else None,
               "cycles_per_valu_inst": round(4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0) / mean["SQ_INSTS_VALU"], 2) if mean["SQ_INSTS_VALU"] else None}
        valu.setdefault(w, {})[short] = rec
        summary.append({"workload": w, "kernel": k[:120], **{cn + "_mean": v for cn, v in mean.items()}})
json.dump({"_method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over the bench command of each "
                      "workload (scripts/profile_r04.sh), means per launch.  valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs): the share of "
                      "SIMD issue slots taken by VALU instructions while the GPU was active (profiler clock, dispatch ramp included); cycles_per_valu_inst = their average issue cost (fp64 arithmetic 4, conversions 8, "
                      "rcp / rsq / sqrt 16, 32-bit integer 2: profiles/r04_valu_instruction_rates.txt).  _durations_us: average launch durations of the "
                      "--kernel-trace --stats pass of the same command.", **valu},
          open(f"{out}/pmc_valu.json", "w"), indent=1)
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)
traffic = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (plus a --kernel-trace --stats pass for the durations) over "
                      "`bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --workload <w>` (mldivide: --steps 2 --warmup 1; reductions: scripts/red_driver.py) - "
                      "scripts/profile_r03.sh.  bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: both counters are in KiB and on gfx950 FETCH_SIZE reports half of a wide "
                      "(16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrates exactly (524288 KiB = a 512 MiB output).  For the GEMMs the "
                      "fetch figure is memory-side (fabric) traffic including Infinity-Cache hits, not DRAM bytes.  `_bytes_per_step` = all kernels of the run except set-up "
                      "(fills, uploads, narrowing) divided by the steps executed.", **traffic}
json.dump(traffic, open(f"{out}/pmc_traffic.json", "w"), indent=1)
for w, t in traffic.items():
    if not w.startswith("_"):
        print(w, json.dumps(t)[:600])
