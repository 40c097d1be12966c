"""pytest configuration.

Markers: `gpu` = needs a real MI355X (run by the driver with `-m gpu`); everything else runs on CPU
(`-m "not gpu"`): oracle vs the reference's KATs/golden vectors, the WGSL front-end + hipRTC
cross-compile, host logic, the exported C ABI, and the world_size-2 gloo tests of the sharding path.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def built():
    """Make sure librmhip.so and liboracle.so exist (build in-tree if the checkout is fresh)."""
    lib = ROOT / "runmat_amd" / "csrc" / "librmhip.so"
    orc = ROOT / "oracle" / "liboracle.so"
    if not lib.exists() or not orc.exists():
        import __graft_entry__ as g

        g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as o

    o.lib()
    return o


@pytest.fixture(scope="session")
def prov(built):
    """A HipProvider on device 0. No fallback: without a gfx950 device construction raises."""
    from runmat_amd import HipProvider

    p = HipProvider(int(os.environ.get("RMHIP_TEST_DEVICE", "0")))
    yield p
    p.close()
