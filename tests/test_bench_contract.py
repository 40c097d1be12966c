"""The default bench line must keep BASELINE.json's own configs where the driver can see them: it records the TAIL of the line
(8 081 characters in rounds 1-4), and in round 4 a longer `also` list pushed the dgemm and Monte-Carlo entries out of it.
Checked here without a GPU: the order bench.py emits the secondary workloads in, and - on the committed default line of the
round (profiles/r06_bench_default.json, written on the GPU box by `python bench.py > ...`) - that the last 8 000 characters still
hold the dgemm, Monte-Carlo (1e8 samples) and mldivide entries with their ms_per_step and the last 2 000 the compact `baseline_configs`."""
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
    """profiles/r06_bench_default.json is the line `python bench.py` printed on the GPU box this round: its `also` list still ends with
    BASELINE's configs inside the last 8 000 characters (the driver's long tail), and - round 6 - its LAST 2 000 characters (the driver's
    short `tail`) hold the compact `baseline_configs` object with all five configs."""
    f = ROOT / "profiles" / "r06_bench_default.json"
    line = f.read_text().strip().splitlines()[-1]
    out = json.loads(line)
    metrics = [a.get("metric", "") for a in out["also"]]
    assert any("8192^3 matmul" in m and m.startswith("fp64") for m in metrics)
    tail_start = len(line) - 8000
    for needle in ("fp64 GFLOP/s (8192^3 matmul", "Monte-Carlo samples/s (1e8-sample", "fp64 GFLOP/s (x = A\\\\b"):
        start, end = _entry_span(line, needle)
        assert start >= tail_start, (needle, start, tail_start, len(line))
        assert '"ms_per_step"' in line[start:end]
    # the headline itself (fused D = sin(A).*B + C) is the first thing on the line; its number is repeated by the driver's parser
    assert out["metric"].startswith("fused elementwise") and out["roofline"]["frac"] > 0
    assert list(out)[-1] == "baseline_configs"
    short = line[-2000:]
    i = short.find('"baseline_configs"')
    assert i >= 0, "the compact object does not fit the driver's 2 000-character tail"
    bc = json.loads(short[i + len('"baseline_configs": '):-1])
    for k in ("c0_chain_1024", "c1_fused_8192", "c2_dgemm_8192", "c3_mc_1e8", "c4_mldivide_16384"):
        assert bc[k]["ms"] > 0 and 0 < bc[k]["frac"] < 1 and bc[k]["cpu"] > 0, (k, bc[k])
    assert bc["c2_dgemm_8192"]["unit"] == "GFLOP/s" and bc["c1_fused_8192"]["frac_slow_sin"] > 0.5 and bc["c3_mc_1e8"]["frac_40B"] > bc["c3_mc_1e8"]["frac"]
    assert "roofline_slow_path" in out and "untimed_busy_loop" not in out["config"]
    lazy = [a for a in out["also"] if "lazy randn" in a.get("metric", "")]
    assert lazy and lazy[0]["ms_per_step"] < 0.45  # the lazy-Z Monte-Carlo step as its own entry


def _fake_line():
    """A bench line shaped like bench.py's, with long strings where the real one has them (no GPU needed)."""
    import bench

    long = "x" * 400

    def rec(metric, unit, bound, extra=None, cpu=True):
        r = {"metric": metric, "value": 123456.789, "unit": unit, "ms_per_step": 12.34567, "scaling": "weak",
             "config": {"workload": long}, "roofline": {"bound": bound, "achieved": 1234.56, "peak": 8192.0, "unit": "GB/s", "frac": 0.7236,
                                                        "kernel": long, "kernel_ms": 0.36229, **(extra or {})}}
        if cpu:
            r["cpu_baseline"] = {"value": 2.2784, "unit": "GB/s", "cores": 1, "kind": "port", "sample": long}
        return r

    out = rec("fused elementwise GB/s (D = sin(A).*B + C, 8192x8192 f64, per-GPU matrices)", "GB/s", "hbm")
    out["roofline_slow_path"] = {"bound": "hbm", "frac": 0.5, "kernel_ms": 0.5}
    out["n_gpus"] = 1
    out["also"] = [
        rec("image_normalize GB/s (4k-image-processing: ...)", "GB/s", "hbm", cpu=False),
        rec("fused elementwise GB/s (elementwise-math 14-op chain, 1024x1024 f64)", "GB/s", "valu"),
        rec("Monte-Carlo samples/s (1e8-sample randn + fused elementwise + sum reduction)", "samples/s", "hbm",
            {"frac_on_32B_moved": 0.68, "frac_on_40B_survey_8d_plan": 0.85}),
        rec("fp64 GFLOP/s (8192^3 matmul, row-block sharded across GPUs)", "GFLOP/s", "mfma"),
        rec("fp64 GFLOP/s (x = A\\b, 16384x16384, blocked LU)", "GFLOP/s", "mfma"),
    ]
    out["device"] = {"arch": "gfx950"}
    out["baseline_configs"] = bench.baseline_configs(out)
    return out


def test_baseline_configs_is_the_last_key_and_fits_the_drivers_2000_character_tail():
    """Round-5 review: the driver's short `tail` is 2 000 characters and began mid-dgemm.  bench.py now ends the line with a compact
    `baseline_configs` object; whatever keeps only the last 2 000 characters still holds all five BASELINE configs."""
    out = _fake_line()
    assert list(out)[-1] == "baseline_configs"
    bc = out["baseline_configs"]
    assert [k for k in bc if k.startswith("c")] == ["c0_chain_1024", "c1_fused_8192", "c2_dgemm_8192", "c3_mc_1e8", "c4_mldivide_16384"]
    for k in ("c0_chain_1024", "c1_fused_8192", "c2_dgemm_8192", "c3_mc_1e8", "c4_mldivide_16384"):
        assert bc[k] is not None and bc[k]["ms"] == 12.34567 and bc[k]["frac"] == 0.7236 and bc[k]["cpu"] == 2.2784, (k, bc[k])
    assert bc["c1_fused_8192"]["frac_slow_sin"] == 0.5 and bc["c3_mc_1e8"]["frac_40B"] == 0.85
    line = json.dumps(out)
    i = line.rfind('"baseline_configs"')
    assert len(line) - i <= 1800, len(line) - i
    tail = line[-2000:]
    assert '"baseline_configs"' in tail and '"c2_dgemm_8192"' in tail and '"c4_mldivide_16384"' in tail
    # and bench.py really emits it last
    src = (ROOT / "bench.py").read_text()
    assert src.index('out["baseline_configs"] = baseline_configs(out)') > src.index('out["device"] =')


def test_cpu_baselines_are_bounded_and_the_busy_loop_is_opt_in():
    src = (ROOT / "bench.py").read_text()
    assert 'os.environ.get("RMHIP_BENCH_BUSY_S", "0")' in src          # advisor r5 (medium): no default busy loop
    assert "n, cols = 2048, 256" in src and "np.log(t1_ / t0_)" not in src  # dgemm: measured rate on a bounded sample, no fitted exponent
