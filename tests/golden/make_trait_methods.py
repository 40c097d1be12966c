"""Extract `trait AccelProvider` (crates/runmat-accelerate-api/src/lib.rs:1386-3151) into tests/golden/accel_provider_methods.json:
data for tests/test_bindings.py.  Per method: the line, the parameters (reference kind + type), the return type - normalised so that
the shim's `impl AccelProvider for HipProvider` can be compared signature by signature - and, for the whole file, the variants of
every `pub enum` (the shim's `match` arms must name variants that exist).  Run in the build container (the GPU box has no
/root/reference):
    python tests/golden/make_trait_methods.py
"""
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from rust_sig import parse_fn_signatures, parse_enums  # noqa: E402  (tests/rust_sig.py: the same parser reads the shim)

SRC = Path("/root/reference/crates/runmat-accelerate-api/src/lib.rs")
OUT = Path(__file__).resolve().parent / "accel_provider_methods.json"

text = SRC.read_text()
lines = text.splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"pub trait AccelProvider\b", l))
depth, end = 0, None
for i in range(start, len(lines)):
    depth += lines[i].count("{") - lines[i].count("}")
    if depth == 0 and i > start:
        end = i
        break
body = "\n".join(lines[start:end + 1])
sigs = parse_fn_signatures(body, first_line=start + 1, indent="    ")
methods = {name: s["line"] for name, s in sigs.items()}
OUT.write_text(json.dumps({"source": "crates/runmat-accelerate-api/src/lib.rs", "trait_lines": [start + 1, end + 1], "methods": methods,
                           "signatures": sigs, "enums": parse_enums(text)}, indent=1) + "\n")
print(f"{len(methods)} methods, trait at lines {start + 1}-{end + 1}, {len(parse_enums(text))} enums")
