"""Round-5 kernels of the solve path through the C ABI (x = A\\b): the two-level driver (super-panels), the matrix-core forms of the
rows-below kernel and of the unit-lower triangular solve (inverted 16 x 16 diagonal blocks), the cooperative yield.  Each is a
rounding-level variation of the same factorisation, so every combination of the switches must solve the same systems to the same
accuracy, bit-identically from run to run; the reference pins `A\\b` by residual (mldivide.rs:662-696)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(n, env, seed=41, reps=1, nrhs=1):
    """x = A\\b in a fresh provider under `env` (the switches are read once per process or per call: set before the first solve)."""
    import subprocess
    import sys
    import json
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = f"""
import json, sys, numpy as np
sys.path.insert(0, {str(root)!r})
from runmat_amd import HipProvider
prov = HipProvider(0)
n = {n}
a = prov.fill_uniform({seed}, -1, 1, (n, n))
b = prov.fill_uniform({seed + 1}, -1, 1, (n, {nrhs}))
xs = []
for rep in range({reps}):
    x = prov.mldivide(a, b)
    xs.append(prov.download(x))
    prov.free(x)
A = prov.download(a).reshape(n, n, order="F"); B = prov.download(b).reshape(n, {nrhs}, order="F")
X = np.asarray(xs[0]).reshape(n, {nrhs}, order="F")
res = float(np.linalg.norm(A @ X - B) / (np.linalg.norm(A) * np.linalg.norm(X)))
same = all(np.array_equal(np.asarray(xs[0]).view(np.uint64), np.asarray(x).view(np.uint64)) for x in xs[1:])
st = prov.lu_stats()
print(json.dumps({{"res": res, "same": same, "x0": float(X[0, 0]), "xsum": float(X.sum()), "fast": st.get("solve_path_factorizations", None), "fallbacks": st.get("pivot_growth_fallbacks", None)}}))
prov.close()
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


SWITCHES = [
    {},                                                          # two-level driver, both matrix-core kernels, yield
    {"RMHIP_LU_SUPER": "0"},                                     # one-level driver
    {"RMHIP_LU_RB_MFMA": "0", "RMHIP_LU_TRSM_MFMA": "0"},         # fp64-VALU rows-below kernel and triangular solves
    {"RMHIP_LU_TRSM_MFMA": "0"},
    {"RMHIP_LU_YIELD": "0", "RMHIP_LU_GEMM_PRIO": "0"},
    {"RMHIP_LU_SUPER_SEQ": "512:256/1024:256", "RMHIP_LU_SUPER_ROWS": "2048", "RMHIP_LU_SUPER_LATE": "512:128"},  # another plan
    {"RMHIP_LU_IPREP": "0", "RMHIP_LU_SMALL_UPD": "0"},           # W-wide solve at the boundary, 128 x 128 tiles only on the update streams
    {"RMHIP_LU_IPREP": "1", "RMHIP_LU_IPREP_SPLIT": "0", "RMHIP_LU_SUPER_SEQ": "256:256/1024:256/2048:256", "RMHIP_LU_SUPER_ROWS": "4096"},  # the large-order plan with incremental block rows
]


@pytest.mark.parametrize("n", [8192, 9000])
def test_every_switch_combination_solves_to_the_same_accuracy(n):
    base = None
    for env in SWITCHES:
        r = _solve(n, env)
        assert r["res"] <= 1e-12 * n, (env, r)
        assert r["fallbacks"] in (0, None), (env, r)  # U(-1,1): the diagonal-domain rule holds, no refactorisation
        if base is None:
            base = r
        else:  # same system, rounding-level variations of one factorisation
            assert abs(r["x0"] - base["x0"]) <= 1e-6 * max(1.0, abs(base["x0"])), (env, r, base)


def test_two_level_driver_is_bit_identical_from_run_to_run():
    r = _solve(12288, {}, reps=3)  # 512-column plan, incremental block rows of U for the next super-panel
    assert r["same"] and r["res"] <= 1e-12 * 12288, r


def test_large_order_plan_is_bit_identical_from_run_to_run():
    r = _solve(14336, {}, reps=2)  # first order of the 256-column plan (W-wide solves at the boundaries)
    assert r["same"] and r["res"] <= 1e-12 * 14336, r
    assert r["fallbacks"] in (0, None), r


def test_small_orders_take_the_matrix_core_kernels_too():
    # below the two-level driver's threshold (8192) the one-level look-ahead driver and, below 1152, the recursive one run the same
    # panel kernels: ragged orders (base panels that are not 64 wide leave no inverses: the old kernels take those blocks)
    for n in (1200, 2048, 3000, 5250):
        r = _solve(n, {})
        assert r["res"] <= 1e-12 * n, (n, r)
    r = _solve(4096, {}, nrhs=5)
    assert r["res"] <= 1e-12 * 4096, r
