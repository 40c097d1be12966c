"""Developer tool (this container): copy what scripts/profile_r03.sh / profile_r04.sh left under gpurun_out/prof_<tag>/ into profiles/ under the round's
names - per workload the `--kernel-trace --stats` kernel summary and the bench line printed under the profiler, the counter summary
and the traffic table bench.py reads.  Usage: collect_profiles.py <tag>   (e.g. r03)"""
import glob, shutil, sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent
src, dst = root / "gpurun_out" / f"prof_{tag}", root / "profiles"
n = 0
for d in sorted(src.glob("trace_*")):
    if not d.is_dir():
        continue
    w = d.name[len("trace_"):]
    for f in glob.glob(str(d / "**" / "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, dst / f"{tag}_{w}_kernel_stats.csv"); n += 1
    b = src / f"trace_{w}_bench.json"
    if b.exists() and b.stat().st_size:
        shutil.copy(b, dst / f"{tag}_{w}_bench_under_rocprof.json"); n += 1
for name in ("mldivide_timeline.txt", "lu_attribution.txt", "mldivide_main_events.txt", "tier2_rates.txt", "red2_rates.txt", "red_shapes.txt", "gemm_variants.txt", "rng_accuracy.txt", "solve_sizes.txt",
             "bench_default.json", "offload_calibration.json"):
    if (src / name).exists() and (src / name).stat().st_size:
        shutil.copy(src / name, dst / f"{tag}_{name}"); n += 1
# merge: a round may re-profile only some workloads; the entries of the others stay
import json


def merge(src_file, dst_file):
    new = json.loads(src_file.read_text())
    if isinstance(new, list):           # rows tagged with their workload: replace the rows of the workloads profiled again
        cur = json.loads(dst_file.read_text()) if dst_file.exists() else []
        again = {r["workload"] for r in new}
        dst_file.write_text(json.dumps([r for r in cur if r["workload"] not in again] + new, indent=1))
        return
    cur = json.loads(dst_file.read_text()) if dst_file.exists() else {}
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(cur.get(k), dict) and k.startswith("_"):
            cur[k].update(v)            # per-workload tables kept under an underscore key (e.g. _durations_us)
        elif not k.startswith("_") or k not in cur:
            cur[k] = v
    dst_file.write_text(json.dumps(cur, indent=1))


merge(src / "pmc_summary.json", dst / f"{tag}_pmc_summary.json")
merge(src / "pmc_traffic.json", dst / "pmc_traffic.json")
if (src / "pmc_valu.json").exists():
    merge(src / "pmc_valu.json", dst / "pmc_valu.json")
print(f"copied {n + 3} files into {dst}")
