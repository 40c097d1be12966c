"""The default bench line must keep BASELINE.json's own configs where the driver can see them: it records the TAIL of the line
(8 081 characters in rounds 1-4), and in round 4 a longer `also` list pushed the dgemm and Monte-Carlo entries out of it.
Checked here without a GPU: the order bench.py emits the secondary workloads in, and - on the committed default line of the
round (profiles/r05_bench_default.json, written on the GPU box by `python bench.py > ...`) - that the last 6 000 characters still
hold the dgemm, Monte-Carlo (1e8 samples) and mldivide entries with their ms_per_step."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_also_list_ends_with_the_baseline_configs():
    src = (ROOT / "bench.py").read_text()
    m = re.search(r'others = \[w for w in \(([^)]*)\) if w != args\.workload\]', src)
    assert m, "bench.py: the list of secondary workloads moved"
    order = [w.strip().strip('"') for w in m.group(1).split(",") if w.strip()]
    assert order[-3:] == ["chain", "mc", "dgemm"], order           # ... and mldivide is appended last of all
    assert "bcast" not in order and "fft" not in order            # workloads of their own, not part of the default line
    assert 'others.append("mldivide")' in src


def _entry_span(line: str, needle: str):
    """(start, end) of the `also` entry whose metric contains `needle`, located in the serialised line."""
    i = line.find(needle)
    assert i >= 0, needle
    start = line.rfind('{"metric"', 0, i)
    j = line.find('{"metric"', i)
    return start, (j if j >= 0 else len(line))


def test_committed_default_line_keeps_the_baseline_configs_in_its_tail():
    f = ROOT / "profiles" / "r05_bench_default.json"
    line = f.read_text().strip().splitlines()[-1]
    out = json.loads(line)
    metrics = [a.get("metric", "") for a in out["also"]]
    assert any("8192^3 matmul" in m and m.startswith("fp64") for m in metrics)
    tail_start = len(line) - 6000
    for needle in ("fp64 GFLOP/s (8192^3 matmul", "Monte-Carlo samples/s (1e8-sample", "fp64 GFLOP/s (x = A\\\\b"):
        start, end = _entry_span(line, needle)
        assert start >= tail_start, (needle, start, tail_start, len(line))
        assert '"ms_per_step"' in line[start:end]
    # the headline itself (fused D = sin(A).*B + C) is the first thing on the line; its number is repeated by the driver's parser
    assert out["metric"].startswith("fused elementwise") and out["roofline"]["frac"] > 0
