"""Oracle vs golden vectors produced by the reference's own Python comparators
(tests/golden/make_golden.py ran /root/reference/benchmarks/... in the build container)."""
import pytest

from workloads import OracleOps, golden_monte_carlo_cases, lcg_monte_carlo_price


@pytest.mark.parametrize("case", golden_monte_carlo_cases(), ids=lambda c: f"M{c['M']}_T{c['T']}")
def test_oracle_lcg_monte_carlo_matches_reference_script(oracle, case):
    # The reference script keeps S and Z in float32 and prints 6 decimals; the f64 oracle pipeline
    # agrees to the f32 rounding level of the payoff mean.
    price = lcg_monte_carlo_price(OracleOps(oracle), case["M"], case["T"])
    assert abs(price - case["price"]) <= 2e-4 * max(1.0, abs(case["price"])), (price, case["price"])
