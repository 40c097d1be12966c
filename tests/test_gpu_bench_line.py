"""The driver's N > 1 bench line, exercised on ONE GPU: two ranks share cuda:0 (RMHIP_BENCH_BACKEND=gloo: gloo control plane, the
library's host shared-memory transport as the data path).  The numbers of such a run mean nothing; what is checked is that the
first run on a multi-GPU node cannot die of a schema slip or of one failing workload: ONE JSON line, every BASELINE config present
with its roofline, the communicator the library itself reports, and a failing workload turned into an "error" record."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra_args, extra_env, nproc=2, timeout=900, backend="gloo"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra_env)
    env.pop("RMHIP_BENCH_BACKEND", None)
    if backend:
        env["RMHIP_BENCH_BACKEND"] = backend
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(nproc)] + extra_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {r.stdout[-1500:]}"
    return json.loads(lines[0])


def _check_roofline(rf):
    assert rf["bound"] in ("hbm", "mfma", "valu") and rf["unit"] in ("GB/s", "TFLOP/s", "Ginstr/s")
    assert rf["achieved"] > 0 and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3
    assert "kernel" in rf and rf["peak_spec"] > 0 and abs(rf["peak"] / rf["peak_spec"] - 1.0) < 0.1  # peaks derived from the box, the spec beside
    if rf["bound"] == "valu":  # instruction counts from the committed counter pass, time measured in this run
        assert rf["valu_insts_per_launch"] > 0 and "pmc_valu.json" in rf["counters_source"]
    # every roofline that names a kernel carries the counter-measured bytes (profiles/pmc_traffic.json): never null
    assert isinstance(rf["traffic"], int) and rf["traffic"] > 0 and "pmc_traffic.json" in rf["traffic_source"]


def test_two_rank_line_has_every_config_and_a_communicator():
    out = _run_bench(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "comm", "also"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert out["value"] > 0 and out["scaling"] == "weak" and "workload" in out["config"]
    _check_roofline(out["roofline"])
    assert out["comm"] == {"transport": "host-shm", "world_seen": 2, "rank_seen": 0}
    seen = {}
    for a in out["also"]:
        assert "error" not in a, a
        for k in ("metric", "value", "unit", "ms_per_step", "scaling", "config", "roofline"):
            assert k in a, (k, a.get("metric"))
        assert a["value"] > 0 and a["ms_per_step"] > 0
        _check_roofline(a["roofline"])
        seen[a["metric"]] = a
    joined = " | ".join(seen)
    # BASELINE.json configs[2..4] under N > 1: row-sharded dgemm, sample-sharded Monte-Carlo, the multi-GPU solve
    assert "8192^3 matmul" in joined and "Monte-Carlo" in joined and "A\\b" in joined
    solve = next(a for m, a in seen.items() if "A\\b" in m)
    assert "row-partitioned x2" in solve["config"]["parallelism"] and solve["config"]["max_abs_err_vs_ones"] < 1e-6


def test_a_failing_workload_becomes_an_error_record():
    out = _run_bench(["--steps", "2", "--warmup", "1", "--workload", "chain", "--no-also", "--no-cpu-baseline"], {"RMHIP_BENCH_TEST_FAIL": "chain"})
    assert out["value"] is None and "forced by RMHIP_BENCH_TEST_FAIL" in out["error"] and out["n_gpus"] == 2
    assert out["comm"]["world_seen"] == 2


def test_eight_rank_line_on_one_gpu():
    """The line the driver launches on an 8-GPU node (`--gpus 8`), with all eight ranks sharing cuda:0 over the host shared-memory
    transport and every linear size divided by four (RMHIP_BENCH_SHRINK): the row-sharded dgemm, the sample-sharded Monte-Carlo and the
    row-partitioned solve all run with world = 8 - shard arithmetic, skip-ahead, eight-way exchanges - and the line keeps its schema."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {"RMHIP_BENCH_SHRINK": "4", "RMHIP_BENCH_BUSY_S": "0"}, nproc=8,
                     timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["shrink"] == 4 and out["value"] > 0
    assert out["comm"] == {"transport": "host-shm", "world_seen": 8, "rank_seen": 0}
    seen = {}
    for a in out["also"]:
        assert "error" not in a, a
        assert a["value"] > 0 and a["ms_per_step"] > 0
        seen[a["metric"]] = a
    solve = next(a for m, a in seen.items() if "A\\b" in m)
    assert "row-partitioned x8" in solve["config"]["parallelism"] and solve["config"]["max_abs_err_vs_ones"] < 1e-6
    mc = next(a for m, a in seen.items() if m.startswith("Monte-Carlo samples/s"))
    assert "x8" in mc["config"]["parallelism"]
    gemm = next(a for m, a in seen.items() if "8192^3 matmul" in m and m.startswith("fp64"))
    assert "x8" in gemm["config"]["parallelism"] or "8" in gemm["config"]["parallelism"]
    # the BASELINE configs close the line (the driver records its tail)
    order = [a["metric"] for a in out["also"]]
    assert "A\\b" in order[-1] and "8192^3 matmul" in order[-2] and order[-3].startswith("Monte-Carlo samples/s")


def test_a_rank_that_fails_its_comm_init_makes_every_rank_fall_back():
    """RCCL's ncclCommInitRank only returns when every rank joined.  Rank 1 is made to fail before it gets there
    (RMHIP_COMM_TEST_FAIL_RANK); rank 0 is inside the real call on the real device and must come back - rmhip_comm_init's bounded
    wait, 4 s here - after which the ranks agree (sharding.Group.try_native_comm) and the line is produced over the control plane."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--workload", "mc", "--no-also", "--no-cpu-baseline"],
                     {"RMHIP_BENCH_TRANSPORT": "rccl", "RMHIP_COMM_TEST_FAIL_RANK": "1", "RMHIP_COMM_INIT_TIMEOUT_S": "4",
                      "RMHIP_BENCH_SHRINK": "4", "RMHIP_BENCH_BUSY_S": "0"})
    assert out["n_gpus"] == 2 and out["value"] > 0 and "error" not in out
    assert out["comm"]["transport"].startswith("torch.distributed") and out["comm"]["world_seen"] == 2
    assert "native communicator unavailable" in out["config"]["collectives"]


@pytest.mark.timeout(240)
def test_a_failing_torch_process_group_does_not_cost_the_line():
    """The driver's launch (default backend: torch's RCCL process group as control plane) with that process group made to fail on every
    rank: the ranks meet again on gloo through a file store and the line comes out (data path here: the host transport, as two ranks
    share the device)."""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also"],
                     {"RMHIP_BENCH_TEST_PG_FAIL": "1", "RMHIP_BENCH_SHRINK": "4", "RMHIP_BENCH_TRANSPORT": "shm", "RMHIP_BENCH_BUSY_S": "0"},
                     backend=None, timeout=200)
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "gloo" in out["config"]["control_plane"] and "forced by RMHIP_BENCH_TEST_PG_FAIL" in out["config"]["control_plane"]
    assert out["comm"]["transport"] == "host-shm"
