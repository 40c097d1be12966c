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
  reference_f64.json   : the elementwise chain and the LCG Monte-Carlo script once more with `numpy.float32` resolving to
                         `numpy.float64` inside the script (main_f64 below): the same arithmetic in double precision, which pins the
                         f64 oracle to ~1e-13 instead of the f32 level of the shipped pipelines.
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


class _Float64Numpy:
    """`numpy` as the comparators see it in the f64 runs: every attribute is numpy's own except `float32`, which is `float64` - so
    `dtype=np.float32`, `np.float32(x)` and `.astype(np.float32)` keep double precision and the script's arithmetic, untouched, is the
    f64 pipeline RunMat's CPU path computes.  Patch-at-import in the build container only: nothing of it ships."""

    def __init__(self):
        import numpy

        self._np = numpy

    def __getattr__(self, name):
        if name == "float32":
            return self._np.float64
        return getattr(self._np, name)


def traced_locals_f64(script: Path, env: dict, func: str = "main") -> dict:
    """`traced_locals` with `import numpy` inside the script resolving to the float64 view above."""
    import numpy as real  # noqa: F401 - make sure the real module is loaded first
    sys.modules["numpy"] = _Float64Numpy()
    try:
        return traced_locals(script, env, func)
    finally:
        sys.modules["numpy"] = real


def main_f64() -> None:
    """Fixtures of the same comparators forced to f64 (pins the f64 oracle at rounding level rather than at the f32 level of the
    shipped scripts): the chain's y2 at 17 sample points, and the LCG Monte-Carlo price at full precision (the local `price`)."""
    elm = REF / "elementwise-math" / "python_numpy.py"
    ecases = []
    for points in (1001, 65537):
        loc = traced_locals_f64(elm, {"ELM_POINTS": str(points)})
        y2 = loc["y2"]
        assert str(y2.dtype) == "float64", y2.dtype
        idx = sorted(set(int(round(i)) for i in [k * (points - 1) / 16.0 for k in range(17)]))
        ecases.append({"points": points, "indices": idx, "y2": [float(y2[i]) for i in idx], "y2_sum": float(y2.sum())})
    script = REF / "monte-carlo-analysis" / "python_numpy_lcg.py"
    mcases = []
    for M, T in [(4096, 4), (65536, 8), (200000, 16)]:
        loc = traced_locals_f64(script, {"MC_M": str(M), "MC_T": str(T)})
        assert str(loc["S"].dtype) == "float64" and str(loc["Z"].dtype) == "float64"
        mcases.append({"M": M, "T": T, "price": float(loc["price"]), "payoff_sum": float(loc["payoff"].sum())})
    (OUT / "reference_f64.json").write_text(json.dumps({
        "source": "benchmarks/elementwise-math/python_numpy.py and benchmarks/monte-carlo-analysis/python_numpy_lcg.py, main() run under a "
                  "tracer with `numpy.float32` resolving to `numpy.float64` inside the script (outputs only)",
        "note": "f64 pipelines: x = linspace(0, 4*pi, points) in f64; drift / scale / Z / S in f64",
        "elementwise_math": ecases, "monte_carlo_lcg": {"params": {"S0": 100.0, "mu": 0.05, "sigma": 0.2, "dt": 1.0 / 252.0, "K": 100.0, "seed": 0},
                                                          "cases": mcases}}, indent=1) + "\n")
    print("wrote", OUT / "reference_f64.json")


def main() -> None:
    main_f64()
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
