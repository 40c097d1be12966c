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


# ---- signature level: the shim's impl block against the trait, method by method (no Rust toolchain in the image) --------------------
def _shim_signatures():
    """Every method of `impl AccelProvider for HipProvider`, hook macros expanded (their templates are `fn $name ...` bodies)."""
    sys.path.insert(0, str(ROOT / "tests"))
    from rust_sig import parse_fn_signatures, parse_signature

    text = (ROOT / "shim" / "hip_provider.rs").read_text()
    macros = {}
    for m in re.finditer(r"macro_rules!\s+([a-z_]+)\s*\{\s*\(\$\(\$name:ident => \$op:expr\),\*\s*\$\(,\)\?\)\s*=>\s*\{\s*\$\(\s*(.*?)\)\*\s*\}\s*\}", text, flags=re.S):
        macros[m.group(1)] = m.group(2)
    start = text.index("impl AccelProvider for HipProvider")
    depth, i = 0, text.index("{", start)
    j = i
    while True:
        depth += text[j] == "{"
        depth -= text[j] == "}"
        j += 1
        if depth == 0:
            break
    body = text[i:j]
    first_line = text[:i].count("\n") + 1
    sigs = parse_fn_signatures(body, first_line=first_line, indent="    ")
    for mname, tmpl in macros.items():
        for inv in re.finditer(rf"\b{mname}!\s*\{{(.*?)\}}", body, flags=re.S):
            for name in re.findall(r"([a-z_0-9]+)\s*=>", inv.group(1)):
                head = re.split(r"\{", tmpl.replace("$name", name), maxsplit=1)[0]
                sig = parse_signature(head)
                assert sig, (mname, name)
                sig["line"] = first_line + body[:inv.start()].count("\n")
                sig["macro"] = mname
                sigs[name] = sig
    return sigs


def _canon(t: str) -> str:
    return t.replace("crate::", "").replace("runmat_accelerate_api::", "")


def test_shim_signatures_match_the_trait():
    golden = json.loads((ROOT / "tests" / "golden" / "accel_provider_methods.json").read_text())
    trait = golden["signatures"]
    assert len(trait) == 243 and set(trait) == set(golden["methods"])
    shim = _shim_signatures()
    unknown = sorted(set(shim) - set(trait))
    assert not unknown, f"methods of the impl block that the trait does not have: {unknown}"
    assert len(shim) >= 200, len(shim)
    bad = []
    for name, s in shim.items():
        t = trait[name]
        if s["async"] != t["async"]:
            bad.append((name, "async", s["async"], t["async"]))
        if len(s["params"]) != len(t["params"]):
            bad.append((name, "arity", len(s["params"]), len(t["params"])))
            continue
        for k, (ps, pt) in enumerate(zip(s["params"], t["params"])):
            if ps["ref"] != pt["ref"] or _canon(ps["type"]) != _canon(pt["type"]):
                bad.append((name, f"param {k}", ps["ref"] + " " + ps["type"], pt["ref"] + " " + pt["type"]))
        if _canon(s["ret"]) != _canon(t["ret"]):
            bad.append((name, "return", s["ret"], t["ret"]))
        if ("'a" in s["generics"]) != ("'a" in t["generics"]):
            bad.append((name, "lifetime parameter", s["generics"], t["generics"]))
    assert not bad, bad[:20]
    # every served method is in the impl block with the trait's signature
    missing = [n for n in served() if n not in shim]
    assert not missing, missing


def test_shim_enum_variants_exist_in_the_api_crate():
    golden = json.loads((ROOT / "tests" / "golden" / "accel_provider_methods.json").read_text())
    enums = golden["enums"]
    assert "ReductionFlavor" in enums and "ProviderPrecision" in enums and len(enums) >= 40
    text = (ROOT / "shim" / "hip_provider.rs").read_text() + (ROOT / "shim" / "wiring.rs").read_text()
    bad = []
    for m in re.finditer(r"\b([A-Z][A-Za-z0-9]+)::([A-Z][A-Za-z0-9]+)\b", text):
        e, v = m.group(1), m.group(2)
        if e in enums and v not in enums[e]:
            bad.append(f"{e}::{v}")
    assert not bad, sorted(set(bad))
