"""SURVEY.md section 8 row a13 - the auto-offload decision (crates/runmat-accelerate/src/native_auto.rs) as the host-side mirror restates
it (include/rmhip_auto_offload.hpp): thresholds, environment overrides, residency / fusion / small-batch rules, the profile cost model and
the calibration file format.  Pure host code: compiled with plain g++ and run on the CPU; the KATs live in examples/auto_offload_kats.cpp.
The calibration file it loads is the one the native calibrator (tests/tools/offload_calibrate.cpp) wrote on an MI355X box."""
import json
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "examples" / "auto_offload_kats"
CAL = ROOT / "profiles" / "r03_offload_calibration.json"


def _build():
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "auto_offload_kats.cpp"), "-o", str(EXE)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_auto_offload_decision_kats():
    _build()
    r = subprocess.run([str(EXE)], capture_output=True, text=True)
    assert r.returncode == 0 and "auto-offload KATs ok" in r.stdout, r.stdout + r.stderr


def test_calibration_file_of_the_native_calibrator_round_trips():
    """what offload_calibrate wrote is what the mirror of `load_calibration_sample` / `apply_calibration_sample` reads: seconds per
    element / per flop = (cpu_time_ms / 1000) / units, the provider block as `rmhip_device_info` reports it"""
    _build()
    r = subprocess.run([str(EXE), str(CAL)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"runs (\d+)\s+cpu_elem_per_elem (\S+)\s+cpu_reduction_per_elem (\S+)\s+cpu_matmul_per_flop (\S+)\s+provider \"(.*)\"", r.stdout)
    assert m, r.stdout
    sample = json.loads(CAL.read_text())["auto_offload_calibration"]
    assert int(m.group(1)) == sample["runs"] > 0
    want = [sample["cpu_time_ms"]["elementwise"] / 1000.0 / sample["units"]["elementwise"],
            sample["cpu_time_ms"]["reduction"] / 1000.0 / sample["units"]["reduction"],
            sample["cpu_time_ms"]["matmul"] / 1000.0 / sample["units"]["matmul_flops"]]
    got = [float(m.group(k)) for k in (2, 3, 4)]
    assert all(abs(g - w) <= 1e-6 * w for g, w in zip(got, want)), (got, want)
    assert m.group(5) == sample["provider"]["name"] and sample["provider"]["vendor"] == "AMD" and sample["provider"]["backend"] == "hip"


def test_header_cites_the_rules_it_restates():
    text = (ROOT / "include" / "rmhip_auto_offload.hpp").read_text()
    for needle in ("native_auto.rs:55-82", ":1416-1449", ":419-476", ":923-1118", "RUNMAT_ACCEL_THRESHOLD_ALL", "RUNMAT_ACCEL_SMALL_BATCH_MIN_ELEMS"):
        assert needle in text, needle


def test_gpu_profile_from_the_device_sweeps_reproduces_the_measured_break_evens():
    """scripts/make_gpu_profile.py turns the calibrator's device-side sweeps into a RUNMAT_ACCEL_PROFILE file; the mirror fits the same
    least-squares lines numpy does, and with the CPU coefficients of the same box its decisions flip where the measured break-evens
    (`recommended_env` of the calibration file) say the device starts to win."""
    import numpy as np

    _build()
    prof = ROOT / "profiles" / "r03_gpu_profile.json"
    r = subprocess.run([str(EXE), "--profile", str(prof), str(CAL)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"profile: reports (\d+)\s+elem (\S+) (\S+)\s+reduction (\S+) (\S+)\s+matmul (\S+) (\S+)", r.stdout)
    assert m, r.stdout
    reports = json.loads(prof.read_text())
    assert int(m.group(1)) == len(reports)
    for cat, (gs, gi) in (("elementwise", (2, 3)), ("reduction", (4, 5)), ("matmul", (6, 7))):
        xs, ys = [], []
        for rep in reports:
            if rep["category"] != cat:
                continue
            a = rep["input_shapes"][0]
            xs.append(float(a[0] * a[1] * rep["input_shapes"][1][1]) if cat == "matmul" else float(np.prod(a)))
            ys.append(rep["total_ms"]["avg_ms"] / 1e3)
        slope, intercept = np.polyfit(np.array(xs), np.array(ys), 1)
        assert abs(float(m.group(gs)) - slope) <= 1e-5 * slope and abs(float(m.group(gi)) - max(intercept, 0.0)) <= 1e-5 * abs(intercept), cat
    env = json.loads(CAL.read_text())["rmhip_break_even"]["recommended_env"]
    dec = {(k, int(n)): v for k, n, v in re.findall(r"decide: (elementwise|matmul) (\d+)\S* -> (gpu|cpu)", r.stdout)}
    be = env["RUNMAT_ACCEL_THRESHOLD_ELEMWISE"]
    assert dec[("elementwise", be)] == "gpu" and dec[("elementwise", be // 2)] == "cpu"
    n_be = round(env["RUNMAT_ACCEL_THRESHOLD_MATMUL"] ** (1.0 / 3.0))
    assert dec[("matmul", n_be)] == "gpu" and dec[("matmul", n_be // 2)] == "cpu"
