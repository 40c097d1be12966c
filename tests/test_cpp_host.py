"""The C++ host side above the C ABI (include/rmhip_provider.hpp): compiles with plain g++ against
the header + librmhip.so; without a GPU it must fail loudly (exit 2: no device, no CPU fallback);
on the GPU box it runs the reference's provider KATs (examples/provider_kats.cpp)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "examples" / "provider_kats"


def _build():
    lib_dir = ROOT / "runmat_amd" / "csrc"
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "provider_kats.cpp"),
           f"-L{lib_dir}", "-lrmhip", f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", str(EXE)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_cpp_provider_header_compiles_and_fails_loudly_without_gpu(built):
    import torch

    _build()
    r = subprocess.run([str(EXE)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_provider_kats_on_gpu(built):
    _build()
    r = subprocess.run([str(EXE)], capture_output=True, text=True)
    assert r.returncode == 0 and "provider KATs ok" in r.stdout, r.stdout + r.stderr
