"""Ordering / race evidence for the multi-stream LU of the solve path (x = A\\b, SURVEY.md section 5 "race detection"; round-5 review item 6).

The two-level driver runs three HIP streams tied by events, a device-side yield table (common.h CuAnnounce) and flag-ordered substitution
kernels.  A missing dependency would show as a result that depends on the schedule.  Here every switch combination solves the same
systems (1) concurrently, three times, and (2) with the runtime made to SERIALIZE every launch and copy (AMD_SERIALIZE_KERNEL=3,
AMD_SERIALIZE_COPY=3: each kernel completes before the next is enqueued, so no two streams ever overlap): all solutions must be
bit-identical.  Sizes cover the recursive driver (2048), the one-level look-ahead (4096), the two-level driver's 512-column plan (8192,
12288) and its large-order plan (16384).  Each (size, switches) pair is 3 + 1 solves: 100+ solves in all.
A second test feeds graded / ill-conditioned matrices to the matrix-core panel kernels, which multiply by explicitly inverted 16 x 16
diagonal blocks where the fp64-VALU kernels substitute (round-5 advisor finding): their backward error must stay at the level of the
substitution kernels'."""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import hashlib, json, sys, numpy as np
sys.path.insert(0, %(root)r)
from runmat_amd import HipProvider
prov = HipProvider(0)
n, reps, kind = %(n)d, %(reps)d, %(kind)r
if kind == "uniform":
    a = prov.fill_uniform(41, -1, 1, (n, n))
    b = prov.fill_uniform(42, -1, 1, (n, 1))
else:
    rng = np.random.default_rng(7)
    R = rng.uniform(-1, 1, (n, n))
    e = float(kind.split(":")[1])
    d1 = 10.0 ** (-e * np.arange(n) / n)
    d2 = 10.0 ** (-e * rng.permutation(n) / n)
    if kind.startswith("rows"):
        A = d1[:, None] * R
    elif kind.startswith("cols"):
        A = R * d2[None, :]
    else:
        A = d1[:, None] * R * d2[None, :]
    a = prov.upload(A)
    b = prov.upload(A @ np.ones((n, 1)))
digests, res = [], None
for rep in range(reps):
    x = prov.mldivide(a, b)
    xh = np.asarray(prov.download(x))
    digests.append(hashlib.sha256(xh.tobytes()).hexdigest())
    if rep == 0:
        A_ = prov.download(a).reshape(n, n, order="F"); B_ = prov.download(b).reshape(n, 1, order="F"); X = xh.reshape(n, 1, order="F")
        res = float(np.linalg.norm(A_ @ X - B_) / (np.linalg.norm(A_) * np.linalg.norm(X)))
    prov.free(x)
st = prov.lu_stats()
print(json.dumps({"digests": digests, "res": res, "fallbacks": st.get("pivot_growth_fallbacks", None)}))
prov.close()
"""


def _run(n, env, reps=1, kind="uniform"):
    code = CHILD % {"root": str(ROOT), "n": n, "reps": reps, "kind": kind}
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


SERIAL = {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"}
CASES = [
    (2048, {}),
    (4096, {}),
    (4096, {"RMHIP_LU_RB_MFMA": "0", "RMHIP_LU_TRSM_MFMA": "0"}),
    (8192, {}),
    (8192, {"RMHIP_LU_SUPER": "0"}),
    (8192, {"RMHIP_LU_YIELD": "0", "RMHIP_LU_GEMM_PRIO": "0"}),
    (8192, {"RMHIP_LU_YIELD_ALL": "7"}),                                   # round 6: every chain kernel counts itself into the yield table
    (8192, {"RMHIP_LU_IPREP": "0", "RMHIP_LU_SMALL_UPD": "0"}),
    (12288, {}),
    (12288, {"RMHIP_LU_TRSM_MFMA": "0"}),
    (16384, {}),
    (16384, {"RMHIP_LU_YIELD_ALL": "7"}),
    (16384, {"RMHIP_LU_MINV": "1"}),                                       # round 6: W-wide solves as products with inverted L11 blocks
    (16384, {"RMHIP_LU_SUPER_SEQ": "512:256/1024:256", "RMHIP_LU_SUPER_ROWS": "2048", "RMHIP_LU_SUPER_LATE": "512:128"}),
]


@pytest.mark.parametrize("n,env", CASES, ids=lambda v: str(v).replace("RMHIP_LU_", "").replace("'", "") if not isinstance(v, int) else str(v))
def test_concurrent_schedule_is_bitwise_the_serialized_one(n, env):
    conc = _run(n, env, reps=3)
    assert len(set(conc["digests"])) == 1, ("run-to-run difference under the concurrent schedule", n, env, conc)
    assert conc["res"] <= 1e-12 * n and conc["fallbacks"] in (0, None), conc
    ser = _run(n, dict(env, **SERIAL), reps=1)
    assert ser["digests"][0] == conc["digests"][0], ("the serialized schedule gives a different solution", n, env, ser["res"], conc["res"])


@pytest.mark.parametrize("kind", ["rows:8", "cols:8", "both:6", "rows:12", "cols:12"])
def test_graded_matrices_matrix_core_panels_against_substitution_panels(kind):
    """cond(A) ~ 1e8 .. 1e12 by graded row / column scalings of a U(-1,1) matrix, b = A*1.  The solve path's matrix-core kernels
    (inverted diagonal blocks of U11 and L11) against the substitution kernels (RMHIP_LU_RB_MFMA=0, RMHIP_LU_TRSM_MFMA=0): the
    normwise backward error of both stays within 1e-12 n - partial pivoting inside the panels bounds the blocks being inverted."""
    n = 2048
    mfma = _run(n, {}, kind=kind)
    valu = _run(n, {"RMHIP_LU_RB_MFMA": "0", "RMHIP_LU_TRSM_MFMA": "0"}, kind=kind)
    assert mfma["res"] <= 1e-12 * n, (kind, mfma)
    assert valu["res"] <= 1e-12 * n, (kind, valu)
    assert mfma["res"] <= 50.0 * max(valu["res"], 1e-17), (kind, mfma, valu)
