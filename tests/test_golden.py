"""Oracle vs golden vectors produced by the reference's own Python comparators
(tests/golden/make_golden.py ran /root/reference/benchmarks/... in the build container)."""
import pytest

import numpy as np

from workloads import (OracleOps, golden_elementwise_math, golden_image_cases, golden_monte_carlo_cases, golden_reference_f64,
                       lcg_image_field, lcg_monte_carlo_price)


@pytest.mark.parametrize("case", golden_monte_carlo_cases(), ids=lambda c: f"M{c['M']}_T{c['T']}")
def test_oracle_lcg_monte_carlo_matches_reference_script(oracle, case):
    # The reference script keeps S and Z in float32 and prints 6 decimals; the f64 oracle pipeline
    # agrees to the f32 rounding level of the payoff mean.
    price = lcg_monte_carlo_price(OracleOps(oracle), case["M"], case["T"])
    assert abs(price - case["price"]) <= 2e-4 * max(1.0, abs(case["price"])), (price, case["price"])


@pytest.mark.parametrize("case", golden_image_cases()["cases"], ids=lambda c: f"B{c['B']}_H{c['H']}_W{c['W']}")
def test_oracle_image_normalize_matches_reference_script(oracle, case):
    # benchmarks/4k-image-processing/python_numpy_lcg.py: MSE between the normalised/gamma-corrected frames and the input
    # field.  The script runs in float32; the f64 oracle agrees to the f32 rounding level of the mean.
    p = golden_image_cases()["params"]
    imgs = lcg_image_field(case["B"], case["H"], case["W"], p["seed"])
    f32 = lambda v: float(np.float32(v))
    out = oracle.image_normalize(imgs, f32(p["eps0"]), gain=f32(p["gain"]), bias=f32(p["bias"]), gamma=f32(p["gamma"]),
                                 clamp_zero=True)
    mse = float(np.mean((out - imgs) ** 2))
    assert abs(mse - case["mse"]) <= 2e-5 * case["mse"], (mse, case["mse"])


@pytest.mark.parametrize("case", golden_elementwise_math()["cases"], ids=lambda c: f"points{c['points']}")
def test_oracle_elementwise_chain_matches_reference_script(oracle, case):
    # benchmarks/elementwise-math/python_numpy.py: y2 at 17 sample points (float32 pipeline there, f64 here)
    n = case["points"]
    x = np.linspace(0.0, 4.0 * np.pi, n, dtype=np.float32).astype(np.float64).reshape(n, 1)
    y2 = oracle.elementwise_math_chain(x)[:, 0]
    got = y2[case["indices"]]
    assert np.max(np.abs(got - np.array(case["y2"]))) <= 2e-6, np.max(np.abs(got - np.array(case["y2"])))


# ---- the same comparators forced to f64 (tests/golden/reference_f64.json): the f64 oracle pinned at rounding level ---------------------
@pytest.mark.parametrize("case", golden_reference_f64()["elementwise_math"], ids=lambda c: f"f64_points{c['points']}")
def test_oracle_elementwise_chain_matches_the_f64_run_of_the_reference_script(oracle, case):
    n = case["points"]
    x = np.linspace(0.0, 4.0 * np.pi, n).reshape(n, 1)
    y2 = oracle.elementwise_math_chain(x)[:, 0]
    # numpy's and the C library's sin / exp / cos / tanh differ by an ulp or two per call; |y2| <= 1.1
    assert np.max(np.abs(y2[case["indices"]] - np.array(case["y2"]))) <= 1e-14
    assert abs(float(y2.sum()) - case["y2_sum"]) <= 1e-12 * n ** 0.5 + 1e-13 * abs(case["y2_sum"])


@pytest.mark.parametrize("case", golden_reference_f64()["monte_carlo_lcg"]["cases"], ids=lambda c: f"f64_M{c['M']}_T{c['T']}")
def test_oracle_lcg_monte_carlo_matches_the_f64_run_of_the_reference_script(oracle, case):
    price = lcg_monte_carlo_price(OracleOps(oracle), case["M"], case["T"], f32_constants=False)
    assert abs(price - case["price"]) <= 1e-12 * max(1.0, abs(case["price"])), (price, case["price"])
