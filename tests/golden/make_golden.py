#!/usr/bin/env python3
"""Regenerates tests/golden/*.json by RUNNING the reference's own Python comparators from
/root/reference (only possible in the build container; the GPU box has no /root/reference and only
reads the committed JSON).  Nothing of the reference is copied: the fixtures are outputs.

  monte_carlo_lcg.json : `python_numpy_lcg.py` (benchmarks/monte-carlo-analysis) printed PRICE for
                         a few (MC_M, MC_T).  The script is the reference's cross-language
                         restatement of runmat_lcg.m (f32 state, f64 LCG arithmetic).
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


def main() -> None:
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
