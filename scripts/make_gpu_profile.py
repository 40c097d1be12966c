"""Build-container tool: turn the device-side sweeps of the native calibrator (profiles/r03_offload_calibration.json, `rmhip_break_even`,
written by tests/tools/offload_calibrate on an MI355X box) into a GPU profile in the format RunMat's auto-offload reads through
RUNMAT_ACCEL_PROFILE (native_auto.rs:1921-1935, 2081-2125): a JSON array of {category, input_shapes, total_ms{avg_ms}} reports from which
it fits one linear cost model per category.  Usage: python scripts/make_gpu_profile.py [calibration.json] [out.json]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r03_offload_calibration.json"
dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r03_gpu_profile.json"
be = json.loads(src.read_text())["rmhip_break_even"]
reports = []
for row in be["elementwise_sweep"]:
    n = int(row["n"])
    reports.append({"category": "elementwise", "name": f"binary_add {n}", "input_shapes": [[n, 1], [n, 1]], "total_ms": {"avg_ms": row["binary_add"]["gpu_us"] / 1e3}})
    reports.append({"category": "reduction", "name": f"reduce_sum {n}", "input_shapes": [[n, 1]], "total_ms": {"avg_ms": row["reduce_sum"]["gpu_us"] / 1e3}})
for row in be["matmul_sweep"]:
    n = int(row["n"])
    reports.append({"category": "matmul", "name": f"matmul {n}^3", "input_shapes": [[n, n], [n, n]], "total_ms": {"avg_ms": row["gpu_us"] / 1e3}})
dst.write_text(json.dumps(reports, indent=1) + "\n")
print(f"{len(reports)} reports -> {dst}")
