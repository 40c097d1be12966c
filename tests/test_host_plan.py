"""Host logic that needs no GPU, checked by small C++ programs over sweeps of shapes: the reduction launch geometry
(reduce_plan.h) and the broadcast stride preparation / dimension collapsing of the elementwise entry points
(host_shape.h), including the in-place addressing of lazy repmat views."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("name,needle", [("reduce_plan_check", "reduce plan ok"), ("broadcast_prep_check", "broadcast prep ok"),
                                         ("repmat_view_check", "repmat view ok")])
def test_host_logic(tmp_path, name, needle):
    exe = tmp_path / name
    c = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'runmat_amd' / 'csrc'}",
                        str(ROOT / "tests" / "cpp" / f"{name}.cpp"), "-o", str(exe)], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and needle in r.stdout, r.stdout + r.stderr
