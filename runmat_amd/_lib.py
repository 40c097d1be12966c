"""ctypes binding of librmhip.so (include/rmhip.h).

There is deliberately no fallback: if the shared library is missing or fails to load this module
raises, and if no gfx950 device is present `rmhip_init` fails (RMHIP_ERR_NO_DEVICE).  The product
path never imports anything under oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "librmhip.so"

OK = 0
COMM_ID_BYTES = 128
COMM_RCCL, COMM_HOST_SHM = 0, 1
ERR_INVALID, ERR_UNSUPPORTED, ERR_SHAPE, ERR_HIP, ERR_NOT_FOUND = 1, 2, 3, 4, 5
ERR_COMPILE, ERR_SINGULAR, ERR_OOM, ERR_NO_DEVICE, ERR_GROWTH = 6, 7, 8, 9, 10


# Structures, signatures and constants are GENERATED from include/rmhip.h (scripts/gen_bindings.py -> _abi.py); the table is
# used both to bind and by the CPU-side test that checks the library exports the full ABI.
from ._abi import (CONSTANTS, ENUMS, SERVES, SIGNATURES, DeviceInfo, ImageNormalize, KernelAttr, KernelLaunch,  # noqa: E402,F401
                   LinsolveOptions, LuStats, MatmulEpilogue, Telemetry, View)

_lib = None


def load() -> C.CDLL:
    """Load librmhip.so (once). Raises OSError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("RMHIP_LIBRARY", str(LIB_PATH)))
    if not path.exists():
        raise OSError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C runmat_amd/csrc`. runmat_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().rmhip_last_error().decode("utf-8", "replace")
