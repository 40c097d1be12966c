"""The four host-side mirrors of the C ABI stay in step mechanically (CPU only):
  * runmat_amd/_abi.py and shim/rmhip_sys.rs are GENERATED from include/rmhip.h (scripts/gen_bindings.py) and must not be stale;
  * every `AccelProvider` method an `@serves` tag of the header names (a) is a method of the reference's trait
    (tests/golden/accel_provider_methods.json, extracted from lib.rs by tests/golden/make_trait_methods.py) and (b) exists in the
    Python mirror (HipProvider), the C++ mirror (include/rmhip_provider.hpp) and the Rust binding (shim/hip_provider.rs)."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def served():
    from runmat_amd import _lib

    names = set()
    for methods in _lib.SERVES.values():
        names.update(methods)
    return sorted(names)


def test_generated_bindings_are_current():
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_prototype_has_a_serves_tag_and_a_binding():
    from runmat_amd import _lib

    header = (ROOT / "include" / "rmhip.h").read_text()
    protos = re.findall(r"^RMHIP_API\s+[^;(]*?\b(rmhip_[a-z0-9_]+)\s*\(", header, flags=re.M)
    assert len(protos) == len(set(protos)) and set(protos) == set(_lib.SIGNATURES) == set(_lib.SERVES)
    tags = re.findall(r"@serves", header)
    assert len(tags) == len(protos) + 1  # one per prototype + the convention note at the top
    rs = (ROOT / "shim" / "rmhip_sys.rs").read_text()
    for name in protos:
        assert f"pub fn {name}(" in rs, name


def test_served_methods_are_trait_methods():
    trait = json.loads((ROOT / "tests" / "golden" / "accel_provider_methods.json").read_text())["methods"]
    unknown = [n for n in served() if n not in trait]
    assert not unknown, f"@serves names that are not AccelProvider methods: {unknown}"
    assert len(served()) >= 130


def test_python_mirror_implements_every_served_method():
    from runmat_amd import HipProvider

    missing = [n for n in served() if not callable(getattr(HipProvider, n, None))]
    assert not missing, missing


def test_cpp_mirror_implements_every_served_method():
    text = (ROOT / "include" / "rmhip_provider.hpp").read_text()
    # a method, or one line of a hook macro; a trait method whose name is a C++ keyword carries a prefix (`union` -> `set_union`)
    cpp_name = {"union": "set_union"}
    missing = [n for n in served() if not re.search(rf"\b{cpp_name.get(n, n)}\s*\(|_HOOK\({n},", text)]
    assert not missing, missing


def test_rust_binding_implements_every_served_method():
    text = (ROOT / "shim" / "hip_provider.rs").read_text()
    assert '#[path = "rmhip_sys.rs"]' in text and 'extern "C" {' not in text  # the FFI block is the generated file
    body = text[text.index("impl AccelProvider for HipProvider"):]
    missing = [n for n in served() if not re.search(rf"\bfn {n}\b|\b{n}\s*=>", body)]
    assert not missing, missing
    assert "same pattern" not in text  # no elided methods
