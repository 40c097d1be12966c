"""runmat_amd/csrc/pow_tables.h (the lookup tables of the image_normalize gamma step, pow_tab.h) is what scripts/gen_pow_tables.py
derives, and the tables have the properties the kernel's error analysis relies on.  CPU only."""
import re
import subprocess
import sys
from decimal import Decimal, getcontext
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _arr(text, name):
    body = text[text.index(name):]
    body = body[body.index("{") + 1:body.index("};")]
    return [float.fromhex(h) for h in re.findall(r"-?0x[0-9a-f.]+p[-+]\d+", body)]


def test_header_is_what_the_generator_writes():
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_pow_tables.py"), "--check"])
    assert r.returncode == 0, "runmat_amd/csrc/pow_tables.h is stale: run scripts/gen_pow_tables.py"


def test_table_properties():
    getcontext().prec = 50
    text = (ROOT / "runmat_amd" / "csrc" / "pow_tables.h").read_text()
    lg, lo, ex = _arr(text, "kPowLog["), _arr(text, "kPowLogLo["), _arr(text, "kPowExp[")
    assert len(lg) == 258 and len(lo) == 129 and len(ex) == 256
    ln2 = Decimal(2).ln()
    for j in range(129):
        inv, hi = lg[2 * j], lg[2 * j + 1]
        c = 1 + j / 128
        assert abs(inv * c - 1.0) < 2.0 ** -50
        assert (hi * 2.0 ** 42).is_integer() and abs(hi) < 0.35            # e ln2_hi + lnc_hi stays exact
        assert abs(lo[j]) <= 2.0 ** -43
        want = -Decimal(inv).ln() - (ln2 if j >= 54 else 0)                # from the STORED reciprocal
        assert abs(Decimal(hi) + Decimal(lo[j]) - want) < Decimal(2) ** -90
        # Fast2Sum(T, r) needs |T| >= |r| or T == 0: the smallest non-zero |lnc| against the largest |r| of its cell
        if hi != 0.0:
            assert abs(hi) > (1 / 256) / c + 2.0 ** -20
    for k in range(128):
        t, tail = ex[2 * k], ex[2 * k + 1]
        want = (ln2 * k / 128).exp()
        assert abs(Decimal(t) * (1 + Decimal(tail)) - want) / want < Decimal(2) ** -75 and abs(tail) <= 2.0 ** -53
