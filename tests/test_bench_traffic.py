"""bench.py's `roofline.traffic` comes from the committed counter table (profiles/pmc_traffic.json, written by
scripts/profile_r03.sh + scripts/pmc_summary.py): every workload bench.py prints must have its entry, and the streaming kernels'
measured bytes must sit within a few percent of the algorithmic bytes DESIGN.md states (no wasted re-reads)."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_bench_workload_has_measured_traffic():
    b = _bench()
    for w, k in (("fused", "rm_ew_fast"), ("dgemm", "k_dgemm_w8"), ("mc", None), ("mc_evolved", None), ("image", None), ("mldivide", None),
                 ("chain", "rm_ew_fast"), ("fused_f32", "rm_ew_fast"), ("sgemm", "k_sgemm_w8"), ("bcast", "k_bcast2"), ("fft", "k_fft_tile")):
        v = b.pmc_traffic(w, k)
        assert isinstance(v, int) and v > 0, (w, k)
    assert '"traffic": None' not in (ROOT / "bench.py").read_text()


def test_streaming_kernels_move_their_algorithmic_bytes():
    t = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    n = 8192 * 8192
    assert abs(t["fused"]["rm_ew_fast"] / (32 * n) - 1) < 0.02          # three reads + one write of 8192^2 f64
    assert abs(t["fused_f32"]["rm_ew_fast"] / (16 * n) - 1) < 0.02      # the same in f32 storage
    assert abs([v for k, v in t["bcast"].items() if k.startswith("k_bcast2")][0] / (8 * n) - 1) < 0.02  # repmat views: only the product is written
    assert abs([v for k, v in t["fft"].items() if k.startswith("k_fft_tile")][0] / (24 * n) - 1) < 0.02  # one pass: 8 B read + 16 B written per element
    assert abs([v for k, v in t["mc"].items() if k.startswith("k_rng_normal<double")][0] / 8e8 - 1) < 0.02  # 1e8 normals written once
    assert abs(t["mc"]["_bytes_per_step"] / (40 * 1e8) - 1) < 0.25      # (32 T + 8) M, T = 1: generator + update + payoff sum
    frames = 16 * 2160 * 3840 * 8
    assert abs(t["image"]["_bytes_per_step"] / (3 * frames) - 1) < 0.02  # moments pass reads, apply pass reads + writes
    for k, v in t["reductions"].items():
        if k.startswith(("k_reduce_contig_v2", "k_reduce_strided_v2", "k_r2_contig<ArgAcc", "k_r2_strided_v2<ArgAcc")):
            assert abs(v / (8 * n) - 1) < 0.03, (k, v)                   # one read of the 8192^2 operand
