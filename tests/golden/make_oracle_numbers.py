#!/usr/bin/env python3
"""Regenerates tests/golden/monte_carlo_rng_oracle.json: prices the CPU oracle (oracle/oracle.c, the
restatement of random.rs + the runmat_rng.m workload) computes at BASELINE.json's full Monte-Carlo
sizes.  The full-size runs take 10-20 s of CPU each, too slow for the GPU-box test budget, so the
numbers are computed once here and committed; the `-m gpu` test compares the device result with them
(rel 1e-10) and the RNG end state bit for bit.  Run from the repo root."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as o  # noqa: E402

cases = []
for M, T in ((100_000_000, 1), (1_000_000, 256), (10_000_001, 2)):
    price, state = o.monte_carlo_price(o.rng_default_seed(), M, T)
    cases.append({"M": M, "T": T, "seed_state": o.rng_default_seed(), "price": price, "price_hex": float(price).hex(),
                  "final_state": int(state)})
    print(cases[-1])
(Path(__file__).resolve().parent / "monte_carlo_rng_oracle.json").write_text(json.dumps(
    {"generator": "tests/golden/make_oracle_numbers.py (oracle.monte_carlo_price: S0=100, mu=0.05, sigma=0.2, dt=1/252, K=100)",
     "cases": cases}, indent=1) + "\n")
