"""The lookup tables of the device Box-Muller step (runmat_amd/csrc/rng_tables.h) are generated data: re-derive them with the
generator's 60-digit arithmetic, check the committed header is what the generator writes, and check the identities the kernel
relies on (exact 0 / +-1 / 0.5 entries, the zero logarithm at c = 2, the bound on the polynomial's argument)."""
import math
import sys
from decimal import Decimal
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))

import gen_rng_tables as gen  # noqa: E402


def test_header_is_up_to_date():
    assert (ROOT / "runmat_amd" / "csrc" / "rng_tables.h").read_text() == gen.render()


def test_sincos_table_identities():
    sc, _ = gen.tables()
    assert len(sc) == 513
    assert sc[0] == (0.0, 1.0) and sc[128] == (1.0, 0.0) and sc[256] == (0.0, -1.0) and sc[384] == (-1.0, 0.0) and sc[512] == (0.0, 1.0)
    for j, (s, c) in enumerate(sc):
        assert abs(s * s + c * c - 1.0) <= 3e-16
        assert abs(s - math.sin(math.pi * j / 256)) <= 1e-15 and abs(c - math.cos(math.pi * j / 256)) <= 1e-15
        # symmetries hold bit for bit: the kernel's results are then symmetric too
        assert sc[512 - j] == (-s + 0.0 if s else 0.0, c) or (sc[512 - j][0] == -s and sc[512 - j][1] == c)
        if j <= 256:
            assert sc[j + 256][0] == -s + 0.0 or sc[j + 256][0] == -s
            assert sc[j + 256][1] == -c + 0.0 or sc[j + 256][1] == -c


def test_log_table_identities():
    _, lg = gen.tables()
    assert len(lg) == 129
    assert lg[0] == (1.0, 0.0) and lg[128] == (0.5, 0.0)  # u1 -> 1: ln m = log1p(m / 2 - 1) alone, no cancellation
    for j, (inv, lnc) in enumerate(lg):
        c = 1.0 + j / 128.0
        assert abs(inv * c - 1.0) <= 2.3e-16
        k = 2 if j >= gen.LOG_SPLIT else 1
        assert abs(Decimal(lnc) + (Decimal(inv) * k).ln()) <= Decimal(2) ** -53 * max(abs(Decimal(lnc)), Decimal(2) ** -60)  # rounded once
        # |m inv - 1| <= 2^-8 (+ the rounding of inv) over the cell [c - 1/256, c + 1/256] the kernel maps to entry j
        for m in (max(1.0, c - 1 / 256), min(2.0, c + 1 / 256)):
            assert abs(m * inv - 1.0) <= 2.0 ** -8 + 1e-15
    assert 1.0 + gen.LOG_SPLIT / 128.0 > math.sqrt(2.0) > 1.0 + (gen.LOG_SPLIT - 1) / 128.0
