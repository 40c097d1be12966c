"""Extract the method names of `trait AccelProvider` (crates/runmat-accelerate-api/src/lib.rs:1386-3151) into
tests/golden/accel_provider_methods.json: data for tests/test_bindings.py, which checks that every method an `@serves`
tag of include/rmhip.h names really is a trait method.  Run in the build container (the GPU box has no /root/reference):
    python tests/golden/make_trait_methods.py
"""
import json
import re
from pathlib import Path

SRC = Path("/root/reference/crates/runmat-accelerate-api/src/lib.rs")
OUT = Path(__file__).resolve().parent / "accel_provider_methods.json"

lines = SRC.read_text().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"pub trait AccelProvider\b", l))
depth, end = 0, None
for i in range(start, len(lines)):
    depth += lines[i].count("{") - lines[i].count("}")
    if depth == 0 and i > start:
        end = i
        break
methods = {}
for i in range(start, end):
    m = re.match(r"    (?:async )?fn ([a-z_0-9]+)", lines[i])
    if m:
        methods[m.group(1)] = i + 1
OUT.write_text(json.dumps({"source": "crates/runmat-accelerate-api/src/lib.rs", "trait_lines": [start + 1, end + 1], "methods": methods},
                          indent=1) + "\n")
print(f"{len(methods)} methods, trait at lines {start + 1}-{end + 1}")
