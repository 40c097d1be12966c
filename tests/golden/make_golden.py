#!/usr/bin/env python3
"""Regenerates tests/golden/*.json by RUNNING the reference's own Python comparators from
/root/reference (only possible in the build container; the GPU box has no /root/reference and only
reads the committed JSON).  Nothing of the reference is copied: the fixtures are outputs.

  monte_carlo_lcg.json : `python_numpy_lcg.py` (benchmarks/monte-carlo-analysis) printed PRICE for
                         a few (MC_M, MC_T).  The script is the reference's cross-language
                         restatement of runmat_lcg.m (f32 state, f64 LCG arithmetic).
  image_normalize_lcg.json : `python_numpy_lcg.py` (benchmarks/4k-image-processing) printed MSE for a few
                         (IMG_B, IMG_H, IMG_W): LCG image field, per-image mean / variance normalisation,
                         gain, bias, clamp, gamma (the ImageNormalize fusion pattern), float32 pipeline.
  elementwise_math.json : `python_numpy.py` (benchmarks/elementwise-math) prints nothing but RESULT_ok, so its
                         `main()` is run under a tracer and the final `y2` it computed (float32) is sampled
                         at 17 indices for a few ELM_POINTS.
"""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

REF = Path("/root/reference/benchmarks")
OUT = Path(__file__).resolve().parent


def run_price(script: Path, M: int, T: int) -> float:
    env = dict(os.environ, MC_M=str(M), MC_T=str(T))
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, check=True).stdout
    m = re.search(r"RESULT_ok PRICE=([-0-9.eE+]+)", out)
    if not m:
        raise RuntimeError(f"unexpected output from {script}: {out!r}")
    return float(m.group(1))


def run_mse(script: Path, B: int, H: int, W: int) -> float:
    env = dict(os.environ, IMG_B=str(B), IMG_H=str(H), IMG_W=str(W))
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, check=True).stdout
    m = re.search(r"RESULT_ok MSE=([-0-9.eE+]+)", out)
    if not m:
        raise RuntimeError(f"unexpected output from {script}: {out!r}")
    return float(m.group(1))


def traced_locals(script: Path, env: dict, func: str = "main") -> dict:
    """Run the script's `func` in-process and return its local variables at return."""
    import runpy

    captured = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name != func:
            return None

        def local(frame, event, arg):
            if event == "return":
                captured.update(frame.f_locals)
            return local

        return local

    old_env = dict(os.environ)
    os.environ.update(env)
    sys.settrace(tracer)
    try:
        runpy.run_path(str(script), run_name="__main__")
    finally:
        sys.settrace(None)
        os.environ.clear()
        os.environ.update(old_env)
    return captured


def main() -> None:
    img = REF / "4k-image-processing" / "python_numpy_lcg.py"
    icases = [{"B": B, "H": H, "W": W, "mse": run_mse(img, B, H, W)} for B, H, W in [(3, 16, 24), (4, 64, 48), (16, 135, 240)]]
    (OUT / "image_normalize_lcg.json").write_text(json.dumps({
        "source": "benchmarks/4k-image-processing/python_numpy_lcg.py (run here, outputs only)",
        "params": {"gain": 1.0123, "bias": -0.02, "gamma": 1.8, "eps0": 1e-6, "seed": 0},
        "note": "float32 pipeline; MSE printed with 6 decimals of mantissa; field = (1664525*idx + 1013904223 mod 2^32) / 2^32, "
                "idx = b*H*W + y*W + x",
        "cases": icases}, indent=1) + "\n")
    print("wrote", OUT / "image_normalize_lcg.json")
    elm = REF / "elementwise-math" / "python_numpy.py"
    ecases = []
    for points in (1001, 65537):
        loc = traced_locals(elm, {"ELM_POINTS": str(points)})
        y2 = loc["y2"]
        idx = sorted(set(int(round(i)) for i in [k * (points - 1) / 16.0 for k in range(17)]))
        ecases.append({"points": points, "indices": idx, "y2": [float(y2[i]) for i in idx]})
    (OUT / "elementwise_math.json").write_text(json.dumps({
        "source": "benchmarks/elementwise-math/python_numpy.py main() run under a tracer (outputs only)",
        "note": "float32 pipeline: x = linspace(0, 4*pi, points, float32); y0 = sin(x)*exp(-x/10); "
                "y1 = y0*cos(x/4) + 0.25*y0^2; y2 = tanh(y1) + 0.1*y1",
        "cases": ecases}, indent=1) + "\n")
    print("wrote", OUT / "elementwise_math.json")
    script = REF / "monte-carlo-analysis" / "python_numpy_lcg.py"
    cases = []
    for M, T in [(4096, 4), (65536, 8), (200000, 16)]:
        cases.append({"M": M, "T": T, "price": run_price(script, M, T)})
    (OUT / "monte_carlo_lcg.json").write_text(json.dumps({
        "source": "benchmarks/monte-carlo-analysis/python_numpy_lcg.py (run here, outputs only)",
        "params": {"S0": 100.0, "mu": 0.05, "sigma": 0.2, "dt": 1.0 / 252.0, "K": 100.0, "seed": 0},
        "note": "reference pipeline keeps S and Z in float32; prices are printed with 6 decimals",
        "cases": cases}, indent=1) + "\n")
    print("wrote", OUT / "monte_carlo_lcg.json")


if __name__ == "__main__":
    main()
