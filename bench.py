#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X accelerate backend (contract: see task brief).

    python bench.py --gpus N --steps K --warmup W [--workload fused|dgemm]

A "step" is one pass of the hot path over one batch of synthetic input, through the C ABI:
  fused (default, BASELINE.json configs[1]): D = sin(A).*B + C on 8192x8192 f64 per GPU, i.e. one
      `rmhip_fused_elementwise` call with the WGSL text the reference planner emits, inputs resident
      in HBM, a NEW output buffer per call (freed to the pool afterwards), exactly what
      `AccelProvider::fused_elementwise` does per fused span.  Algorithmic bytes per step per GPU
      = 4 arrays x 8 B x 67 108 864 = 2 147 483 648.  Weak scaling: every rank owns
      independent matrices, no data-path collective.
  dgemm (BASELINE.json configs[2]): C = A*B 8192^3 f64 on the fp64 MFMA kernel; with N > 1 the
      matrix is row-block sharded (rank g computes C[rows_g,:] = A[rows_g,:]*B, B replicated, no
      collective in the timed region) -- strong scaling.
The JSON line carries `roofline` (dominant kernel, HIP-event timed on the library's stream) and,
on rank 0 at N=1, `cpu_baseline` (the oracle = C port of the reference CPU path, 1 thread, on a
bounded sample). The secondary workload is reported under "also" at N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / "tests"))  # the request emitter (tests/planner_requests.py: what RunMat's planner would send) is test / bench infrastructure

import numpy as np  # noqa: E402

N_DIM = 8192
# Stated spec (MI355X_MICROARCH.md): what the `peak_spec` fields carry.  The `peak` every roofline fraction is taken against is derived
# from the box the run is on (device_peaks below: rmhip_device_info's CU count, clocks and memory bus) - BASELINE.md section 4.
HBM_PEAK_GBS = 8000.0      # HBM3E 8.0 TB/s
FP64_MFMA_PEAK_TF = 78.6   # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64, 64 cyc)
F32_MFMA_PEAK_TF = 157.3   # 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz (v_mfma_f32_32x32x2_f32)
VALU_F64_PEAK_GIPS = 614.4  # wave64 fp64 VALU instructions per second: 1024 SIMDs x 2.4 GHz / 4 cycles (16 lanes per clock)


def device_peaks(info: dict) -> dict:
    """Peaks of THIS box from rmhip_device_info (hipDeviceProp_t): compute_units x clock for the matrix / vector pipes, memory bus x
    pin rate for HBM (HBM3E moves 4 bits per pin per reported memory clock: 2000 MHz -> 8 Gb/s per pin).  amd-smi's view of the memory
    clock is recorded beside it when the tool is there."""
    cus, mhz = int(info["compute_units"]), int(info["clock_mhz"])
    simds = cus * 4
    mem_khz, bus = int(info.get("memory_clock_khz", 0)), int(info.get("memory_bus_width_bits", 0))
    peaks = {
        "source": "rmhip_device_info (hipDeviceProp_t) of the device this run used",
        "compute_units": cus, "clock_mhz": mhz, "xcd_count": int(info.get("xcd_count", 0)),
        "memory_clock_khz": mem_khz, "memory_bus_width_bits": bus,
        "mfma_f64_tflops": round(simds * 32 * mhz * 1e6 / 1e12, 2),
        "mfma_f32_tflops": round(simds * 64 * mhz * 1e6 / 1e12, 2),
        "valu_f64_ginstr_per_s": round(simds * mhz * 1e6 / 4 / 1e9, 1),
        "hbm_gbs": round(bus / 8 * 4 * mem_khz * 1e3 / 1e9, 1) if mem_khz and bus else HBM_PEAK_GBS,
        "spec": {"hbm_gbs": HBM_PEAK_GBS, "mfma_f64_tflops": FP64_MFMA_PEAK_TF, "mfma_f32_tflops": F32_MFMA_PEAK_TF,
                 "valu_f64_ginstr_per_s": VALU_F64_PEAK_GIPS},
    }
    try:
        import subprocess

        r = subprocess.run(["amd-smi", "metric", "--clock", "--json"], capture_output=True, text=True, timeout=20)
        if r.returncode == 0:
            j = json.loads(r.stdout)
            clk = (j[0] if isinstance(j, list) else j).get("clock", {})
            peaks["amd_smi_mem_clock"] = {k: v for k, v in clk.items() if k.lower().startswith("mem")}
    except Exception:  # noqa: BLE001 - optional context only
        pass
    return peaks


def host_info() -> dict:
    """The box the CPU baseline ran on: CPU model and core count; the baselines use ONE thread, like
    the reference's single-threaded CPU path."""
    model = "unknown"
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"host_cpu": model, "host_logical_cores": os.cpu_count() or 0}


def cpu_baseline_fused(budget_s: float = 6.0):
    """Oracle (C port of the reference CPU path: 3 passes, 3 temporaries, libm sin) on one core."""
    from oracle import oracle

    rows, cols = N_DIM, 1024  # 1/8 of the workload
    n = rows * cols
    A = oracle.fill_uniform(1, -np.pi, np.pi, n)
    B = oracle.fill_uniform(2, -1.0, 1.0, n)
    Cc = oracle.fill_uniform(3, -1.0, 1.0, n)
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 64:
        t0 = time.perf_counter()
        oracle.sin_mul_add(A, B, Cc)
        t_total += time.perf_counter() - t0
        reps += 1
    gbs = 32.0 * n * reps / t_total / 1e9
    return {"value": round(gbs, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x D=sin(A).*B+C on {rows}x{cols} f64 (1/8 of the workload), oracle/oracle.c "
                      f"orc_sin_mul_add, {t_total:.1f} s"}


def cpu_baseline_fft():
    """The oracle transforms by DIRECT evaluation of the DFT in long double (a definition, not an algorithm: O(n^2) per line) - timed on a
    few 2048-point lines and reported in the workload's unit on its 24 B per input element; not a statement about CPU FFT libraries."""
    from oracle import oracle

    lines, m = 4, 2048
    x = oracle.fill_uniform(41, -1.0, 1.0, lines * m).reshape(m, lines, order="F")
    t0 = time.perf_counter()
    oracle.fft_dim(x, None, 0)
    dt = time.perf_counter() - t0
    return {"value": round(24.0 * lines * m / dt / 1e9, 6), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{lines} lines of {m} points, oracle/oracle.c orc_dft_dim (direct O(n^2) sums in long double), {dt:.1f} s"}


def cpu_baseline_dgemm():
    """Naive column-major triple loop (linalg.rs:6-32), one core.  SURVEY.md 8(d): measure at n = 2048 and extrapolate to 8192^3 with
    time ~ n^3.  The loop's outermost index runs over the columns of B and every column costs the same (same strides, same cache
    footprint: all of A, one column of B), so a bounded sample is the first 256 of the 2048 columns of the 2048^3 product (1/8 of it,
    ~12 s); `value` is the rate measured on that sample, nothing fitted."""
    from oracle import oracle

    n, cols = 2048, 256
    A = oracle.fill_uniform(11, -1.0, 1.0, n * n).reshape(n, n, order="F")
    B = oracle.fill_uniform(12, -1.0, 1.0, n * n).reshape(n, n, order="F")[:, :cols]
    t0 = time.perf_counter()
    oracle.matmul(A, np.asfortranarray(B))
    dt = time.perf_counter() - t0
    rate = 2.0 * n * n * cols / dt / 1e9
    est_8192_s = 2.0 * 8192.0 ** 3 / (rate * 1e9)
    return {"value": round(rate, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"naive triple loop, {n}x{n} * {n}x{cols} (the first {cols} columns of the {n}^3 product: same loop, same strides), "
                      f"{dt:.1f} s measured; 8192^3 at this rate (time ~ n^3, SURVEY 8(d)): {est_8192_s / 60.0:.0f} min (not run; "
                      f"the rate falls further once A leaves the caches, so this is a lower bound of the time)"}


def cpu_baseline_chain():
    """BASELINE configs[0]: the reference CPU path on elementwise-math @1024x1024 f64 -- 14 builtin
    passes with one temporary each (oracle/oracle.c orc_elementwise_math_chain), one core."""
    from oracle import oracle

    m = 1024
    x = np.linspace(0.0, 4.0 * np.pi, m * m)
    reps, t_total = 0, 0.0
    while t_total < 3.0 and reps < 200:
        t0 = time.perf_counter()
        oracle.elementwise_math_chain(x)
        t_total += time.perf_counter() - t0
        reps += 1
    ms = t_total / reps * 1e3
    return {"value": round(16.0 * m * m / (ms * 1e-3) / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x full 1024x1024 chain, {ms:.1f} ms each (16 B/elem algorithmic, fused form)"}


def cpu_baseline_mc():
    """Oracle Monte-Carlo (serial LCG + Box-Muller + one temporary per op) on a bounded sample."""
    from oracle import oracle

    M = 10_000_000
    t0 = time.perf_counter()
    price, _ = oracle.monte_carlo_price(oracle.rng_default_seed(), M, 1)
    dt = time.perf_counter() - t0
    return {"value": round(M / dt, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"M=1e7 paths, T=1 (1/10 of the workload), oracle/oracle.c orc_monte_carlo_price, {dt:.1f} s, "
                      f"price {price:.6f}"}


def cpu_baseline_mldivide():
    """Oracle A\\b (SVD pseudo-inverse solve, the reference CPU algorithm restated with a Jacobi SVD)."""
    from oracle import oracle

    n = 768
    A = oracle.fill_uniform(31, -1.0, 1.0, n * n).reshape(n, n, order="F") + n * np.eye(n)
    b = A @ np.ones((n, 1))
    t0 = time.perf_counter()
    x = oracle.mldivide_svd(A, b)
    dt = time.perf_counter() - t0
    flops = (2.0 / 3.0) * n ** 3 + 2.0 * n * n
    return {"value": round(flops / dt / 1e9, 5), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"SVD-based solve (mldivide.rs:380-404 restated) at n={n}, {dt:.1f} s, LU-equivalent flop count; "
                      f"max|x-1|={float(np.max(np.abs(x - 1.0))):.1e}; cost grows ~n^3"}


BASELINE_CONFIG_KEYS = (  # BASELINE.json configs[i] -> (short key, a substring of the record's metric)
    ("c0_chain_1024", "14-op chain"), ("c1_fused_8192", "D = sin(A).*B + C, 8192x8192 f64"), ("c2_dgemm_8192", "fp64 GFLOP/s (8192^3 matmul"),
    ("c3_mc_1e8", "Monte-Carlo samples/s (1e8-sample"), ("c4_mldivide_16384", "fp64 GFLOP/s (x = A\\b"))


def baseline_configs(out: dict) -> dict:
    """The five BASELINE.json configs of one bench line in compact form: value, unit, ms_per_step, roofline bound / fraction, kernel
    milliseconds (HIP events) and the CPU baseline's value, taken from the headline record and the `also` entries of `out`."""
    recs = [out] + [a for a in out.get("also", []) if "error" not in a]
    res = {}
    for key, needle in BASELINE_CONFIG_KEYS:
        r = next((x for x in recs if needle in (x.get("metric") or "")), None)
        if r is None:
            res[key] = None
            continue
        rf = r.get("roofline") or {}
        e = {"value": r.get("value"), "unit": r.get("unit"), "ms": r.get("ms_per_step"), "bound": rf.get("bound"), "frac": rf.get("frac")}
        if rf.get("kernel_ms") is not None:
            e["kernel_ms"] = rf["kernel_ms"]
        if r.get("roofline_slow_path"):
            e["frac_slow_sin"] = r["roofline_slow_path"].get("frac")
        if "frac_on_40B_survey_8d_plan" in rf:
            e["frac_40B"] = rf["frac_on_40B_survey_8d_plan"]
        cb = r.get("cpu_baseline")
        if cb:
            e["cpu"] = cb.get("value")
            e["cpu_unit"] = cb.get("unit")
        res[key] = e
    res["n_gpus"] = out.get("n_gpus")
    return res


# (what the key means: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, committed under profiles/; not re-measured in this run)
PMC_TRAFFIC_SOURCE = "profiles/pmc_traffic.json"


def pmc_traffic(workload: str, kernel: str = None):
    """HBM-side bytes measured with rocprofv3 --pmc (committed under profiles/, scripts/profile_r03.sh): per launch of `kernel`
    (prefix match) in the run of `workload`, or - kernel None - per bench step of that workload (all its kernels).  None when the
    file or the entry is missing."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    try:
        t = json.loads(f.read_text()).get(workload) or {}
    except Exception:
        return None
    if kernel is None:
        return t.get("_bytes_per_step")
    for k, v in t.items():
        if k.startswith(kernel):
            return v
    return None


# (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... pass of the same command, committed; not re-measured in this run)
PMC_VALU_SOURCE = "profiles/pmc_valu.json"


def pmc_valu(workload: str, kernel: str):
    """VALU counters of `kernel` (prefix match) in the profiled run of `workload` (profiles/pmc_valu.json, scripts/profile_r04.sh): wave-level
    VALU instructions per launch, the share of SIMD cycles they kept busy under the profiler, their mean issue cost.  None if absent."""
    try:
        t = json.loads((ROOT / "profiles" / "pmc_valu.json").read_text()).get(workload) or {}
    except Exception:
        return None
    for k, v in t.items():
        if k.startswith(kernel):
            return v
    return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["fused", "dgemm", "mc", "mc_lazy", "mldivide", "chain", "mc_evolved", "image", "fused_f32", "sgemm", "bcast", "fft"], default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    args = ap.parse_args()

    import torch

    if os.environ.get("RMHIP_BENCH_GC", "0") != "1":
        # the timed loops are a few hundred Python calls each; a generation-2 collection over torch's module graph in the middle of
        # one is tens of milliseconds of host time that has nothing to do with the device
        import gc

        gc.collect()
        gc.freeze()
        gc.disable()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dist = None
    # Developer knob: RMHIP_BENCH_BACKEND=gloo lets several ranks share ONE GPU (device = local_rank mod
    # device count) so the multi-rank control flow can be exercised on a single-GPU box; numbers from such a
    # run are meaningless.  The driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("RMHIP_BENCH_BACKEND", "nccl")
    if backend != "nccl" or os.environ.get("RMHIP_BENCH_TEST_PG_FAIL"):
        local_rank = local_rank % max(1, torch.cuda.device_count())
    coll_device = "cuda" if backend == "nccl" else "cpu"
    control_plane = "none (single rank)"
    if world > 1:
        import datetime

        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            # The control plane must not cost the line either: if torch's own RCCL process group cannot be made (the eager init is a
            # collective, so every rank sees the failure), the ranks of this node meet again on gloo through a file store (under
            # torchrun the workers never host a TCP store themselves); the data path stays rmhip_comm_*.
            try:
                if os.environ.get("RMHIP_BENCH_TEST_PG_FAIL"):  # test hook (every rank)
                    raise RuntimeError("forced by RMHIP_BENCH_TEST_PG_FAIL")
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                control_plane = "torch.distributed nccl"
            except Exception as e:  # noqa: BLE001
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                # One file per RUN (the launcher's run id is the nonce: a stale file of an earlier run with the same launcher pid and port
                # must not be met again).  Single-node only - the path is node-local, like this bench (--nnodes=1).  Known limit: the
                # fall-back assumes the eager RCCL init failed on EVERY rank (it is a collective); a rank that alone fails waits here
                # for its peers until the 120 s timeout and the run ends with an error record rather than a hang.
                nonce = os.environ.get("TORCHELASTIC_RUN_ID", "norunid") + "_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
                store = f"/tmp/rmhip_bench_pg_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{nonce}"
                dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=120))
                dist.barrier()
                if rank == 0:
                    try:
                        os.unlink(store)  # every rank has joined: the rendezvous file is no longer needed
                    except OSError:
                        pass
                coll_device = "cpu"
                control_plane = f"torch.distributed gloo (nccl process group failed: {type(e).__name__}: {str(e)[:120]})"
        else:
            dist.init_process_group(backend)
            control_plane = f"torch.distributed {backend}"
    else:
        torch.cuda.set_device(local_rank)

    from runmat_amd import HipProvider
    from planner_requests import sin_mul_add_plan

    prov = HipProvider(local_rank)
    # Developer / test knob: RMHIP_BENCH_SHRINK=k divides every workload's linear size by k (the 8-rank run of this line on ONE GPU,
    # tests/test_gpu_bench_line.py).  The metric strings keep naming the full sizes; `config.shrink` says what ran.  Default 1.
    shrink = max(1, int(os.environ.get("RMHIP_BENCH_SHRINK", "1")))
    n = N_DIM // shrink
    peaks = device_peaks(prov.device_info_struct())
    # every roofline below divides by the peaks of the box it ran on; the spec values ride along as `peak_spec`
    hbm_peak, mfma_peak, mfma32_peak, valu_peak = peaks["hbm_gbs"], peaks["mfma_f64_tflops"], peaks["mfma_f32_tflops"], peaks["valu_f64_ginstr_per_s"]

    def roofline(bound: str, achieved: float, digits: int = 2, **extra) -> dict:
        peak, spec, unit = {"hbm": (hbm_peak, HBM_PEAK_GBS, "GB/s"), "mfma": (mfma_peak, FP64_MFMA_PEAK_TF, "TFLOP/s"),
                            "mfma_f32": (mfma32_peak, F32_MFMA_PEAK_TF, "TFLOP/s"), "valu": (valu_peak, VALU_F64_PEAK_GIPS, "Ginstr/s")}[bound]
        return {"bound": "mfma" if bound == "mfma_f32" else bound, "achieved": round(achieved, digits), "peak": peak, "peak_spec": spec, "unit": unit,
                "frac": round(achieved / peak, 4), **extra}

    def valu_roofline(workload: str, kname: str, kernel_s: float, **extra) -> dict:
        """fp64-VALU roofline of one kernel: achieved = wave-level VALU instructions per launch (committed counter pass) / the launch time
        measured in THIS run; peak = SIMDs x clock / 4 (an fp64 instruction occupies its SIMD for four cycles).  The counter pass's busy
        share and mean issue cost ride along.  Falls back to an HBM block when the counter file has no entry (fresh checkout)."""
        v = pmc_valu(workload, kname)
        if not v or not v.get("insts_valu_per_launch"):
            return {**roofline("hbm", 0.0), "note": f"no VALU counters for {workload}/{kname} in profiles/pmc_valu.json", **extra}
        return roofline("valu", v["insts_valu_per_launch"] / kernel_s / 1e9, 1, valu_insts_per_launch=v["insts_valu_per_launch"],
                        valu_busy_profiled=v.get("valu_busy"), cycles_per_valu_inst=v.get("cycles_per_valu_inst"), counters_source=PMC_VALU_SOURCE, **extra)

    def mc_kernel_rooflines(samples: int):
        """Per-kernel view of one Monte-Carlo step from the committed profile (durations, bytes and VALU counters of the same command):
        which of the three kernels is bound by what."""
        out = {}
        try:
            tr = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get("mc", {})
            dur = json.loads((ROOT / "profiles" / "pmc_valu.json").read_text()).get("_durations_us", {}).get("mc", {})
        except Exception:  # noqa: BLE001
            return None
        for name, moved in (("k_rng_normal", 8 * samples), ("rm_ew_fast", 16 * samples), ("rm_red_contig", 8 * samples)):
            us = next((v for k, v in dur.items() if k.startswith(name)), None)
            val = pmc_valu("mc", name)
            if us is None:
                continue
            rec = {"avg_us_profiled": us, "hbm_gbs": round(moved / (us * 1e-6) / 1e9, 1), "hbm_frac": round(moved / (us * 1e-6) / 1e9 / hbm_peak, 4),
                   "measured_bytes": next((v for k, v in tr.items() if k.startswith(name)), None)}
            if val and val.get("insts_valu_per_launch"):
                rec["valu_ginstr_per_s"] = round(val["insts_valu_per_launch"] / (us * 1e-6) / 1e9, 1)
                rec["valu_frac"] = round(rec["valu_ginstr_per_s"] / valu_peak, 4)
                rec["valu_busy_profiled"] = val.get("valu_busy")
                rec["bound"] = "valu" if rec["valu_frac"] > rec["hbm_frac"] else "hbm"
            out[name] = rec
        return out or None
    # Data-path collectives go through the C ABI (rmhip_comm_*: RCCL over xGMI, one rank per GPU); torch.distributed
    # is the control plane only (rendezvous of the 128-byte communicator id, the timing reduction).
    from runmat_amd import sharding as sh

    group = sh.Group.from_env()
    comm_note = "none (single rank)"
    if world > 1:
        # every rank tries; the ranks then agree (sharding.Group.try_native_comm) so that either all of them use the native
        # communicator or all fall back to exchanging through torch.distributed - the run must not die (or hang) on a comm init
        transport = os.environ.get("RMHIP_BENCH_TRANSPORT", "rccl" if backend == "nccl" else "shm")
        _, comm_note = group.try_native_comm(prov, transport=transport)

    def barrier():
        prov.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run_fused(steps, warmup, a_range=np.pi):
        plan, out_id = sin_mul_add_plan()
        shader = plan.generate_wgsl_for_output(out_id, "f64")
        base = 100 * rank  # independent matrices per rank
        ha = prov.fill_uniform(1 + base, -a_range, a_range, (n, n))
        hb = prov.fill_uniform(2 + base, -1.0, 1.0, (n, n))
        hc = prov.fill_uniform(3 + base, -1.0, 1.0, (n, n))

        def step():
            prov.free(prov.fused_elementwise(shader, [ha, hb, hc], (n, n), n * n))

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        wall = time.perf_counter() - t0
        # roofline leg: HIP events on the library's stream around the same launches
        prov.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = prov.timer_end() / steps
        for h in (ha, hb, hc):
            prov.free(h)
        return wall, kern_ms

    def run_dgemm(steps, warmup):
        rows = n // world  # row-block shard of A and C; B replicated
        ha = prov.fill_uniform(11 + 1000 * rank, -1.0, 1.0, (rows, n))
        hb = prov.fill_uniform(12, -1.0, 1.0, (n, n))

        def step():
            prov.free(prov.matmul(ha, hb))

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        wall = time.perf_counter() - t0
        prov.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = prov.timer_end() / steps
        prov.free(ha)
        prov.free(hb)
        return wall, kern_ms

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    fused_bytes = 4 * 8 * n * n          # per GPU per step
    dgemm_flops = 2.0 * n ** 3           # whole job per step

    def fused_record(steps, warmup):
        wall, kern_ms = run_fused(steps, warmup)
        wall = max_over_ranks(wall)
        ms = wall / steps * 1e3
        achieved = fused_bytes / (kern_ms * 1e-3) / 1e9
        # SURVEY.md 8(d) config 2's second run: A in [-1e6, 1e6], where sin leaves its small-argument path (Payne-Hanek-style
        # reduction for most lanes) - same request, same bytes, a short leg of its own; the headline stays the |A| <= pi run.
        slow_steps = max(5, min(steps, 50))
        _, slow_ms = run_fused(slow_steps, 3, a_range=1.0e6)
        slow_ms = max_over_ranks(slow_ms)
        slow = roofline("hbm", fused_bytes / (slow_ms * 1e-3) / 1e9, kernel="rm_ew_fast, same request with A = U(-1e6, 1e6) (large-argument sin)",
                        kernel_ms=round(slow_ms, 5), steps=slow_steps)
        return {
            "roofline_slow_path": slow,
            "metric": "fused elementwise GB/s (D = sin(A).*B + C, 8192x8192 f64, per-GPU matrices)",
            "value": round(world * fused_bytes / (ms * 1e-3) / 1e9, 2), "unit": "GB/s",
            "ms_per_step": round(ms, 5), "scaling": "weak", "dtype": "f64",
            "config": {"workload": "fused D=sin(A).*B+C 8192x8192 f64 via rmhip_fused_elementwise (WGSL request)",
                       "bytes_per_step_per_gpu": fused_bytes, "parallelism": f"independent x{world}"},
            "roofline": roofline("hbm", achieved, traffic=pmc_traffic("fused", "rm_ew_fast"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="rm_ew_fast (hipRTC, generated)", kernel_ms=round(kern_ms, 5)),
        }

    def dgemm_record(steps, warmup):
        wall, kern_ms = run_dgemm(steps, warmup)
        wall = max_over_ranks(wall)
        ms = wall / steps * 1e3
        achieved = (dgemm_flops / world) / (kern_ms * 1e-3) / 1e12
        return {
            "metric": "fp64 GFLOP/s (8192^3 matmul, row-block sharded across GPUs)",
            "value": round(dgemm_flops / (ms * 1e-3) / 1e9, 1), "unit": "GFLOP/s",
            "ms_per_step": round(ms, 5), "scaling": "strong", "dtype": "f64",
            "config": {"workload": "C=A*B dgemm 8192x8192x8192 f64 via rmhip_matmul", "flops_per_step": dgemm_flops,
                       "parallelism": f"row-block x{world}, B replicated, no collective"},
            "roofline": roofline("mfma", achieved, 3, traffic=pmc_traffic("dgemm", "k_dgemm_w8"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="k_dgemm_w8<false> (eight waves, pipelined k loop; v_mfma_f64_16x16x4_f64)", kernel_ms=round(kern_ms, 5)),
        }

    def mc_record(steps, warmup, lazy=False):
        """BASELINE configs[3]: Monte-Carlo GBM, M = 1e8 paths (sharded over ranks with LCG skip-ahead),
        T = 1 step, planner-shaped fused kernels; one step = one full pricing.  `lazy` False: Z = randn(M, 1) is MATERIALISED (the
        plan SURVEY.md 8(d) prices: rmhip_set_lazy_random off for this record); True: the library's default, Z is a lazy handle the
        update kernel generates in registers - reported as its own entry (mc_lazy_record)."""
        from planner_requests import monte_carlo_shaders

        M, T = 100_000_000 // (shrink * shrink), 1
        shaders = monte_carlo_shaders(100.0)  # compiled once per script by the planner, not per call (fusion.rs:679-682)
        price = 0.0
        prov.set_lazy_random(lazy)
        try:
            for _ in range(warmup):
                price, _ = sh.monte_carlo_price_fused(prov, group, M, T, shaders, rng_state=0x9E3779B97F4A7C15)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                price, _ = sh.monte_carlo_price_fused(prov, group, M, T, shaders, rng_state=0x9E3779B97F4A7C15)
            barrier()
            wall = max_over_ranks(time.perf_counter() - t0)
        finally:
            prov.set_lazy_random(True)
        ms = wall / steps * 1e3
        if lazy:
            # bytes the two kernels move per sample: the update writes S (8), the payoff reduction reads it (8).  The update kernel now
            # carries the Box-Muller step and exp: it is bound by the fp64 VALU, the fraction of HBM is what the step as a whole reaches
            moved = 16 * T * M
            return {
                "metric": "Monte-Carlo samples/s (1e8 samples, lazy randn generated inside the fused update + sum reduction)",
                "value": round(M * T / (ms * 1e-3), 1), "unit": "samples/s", "ms_per_step": round(ms, 4), "scaling": "strong", "dtype": "f64",
                "config": {"workload": "monte-carlo-analysis f64, M=1e8, T=1, CPU-parity LCG randn stream, Z never materialised "
                                       "(rmhip_set_lazy_random on: the library's default)", "price": price, "algorithmic_bytes": moved,
                           "parallelism": f"sample ranges x{world}, ordered 1-value exchange"},
                "roofline": roofline("hbm", moved / world / (ms * 1e-3) / 1e9, 1,
                                     kernel="rm_ew_fast with an in-register Box-Muller operand (fp64 VALU bound) + rm_red_contig (whole step, wall clock; "
                                            "16 B per sample moved)",
                                     traffic=pmc_traffic("mc_lazy"), traffic_source=PMC_TRAFFIC_SOURCE + " (per step)"),
            }
        # Bytes the three kernels MOVE per sample: randn writes Z (8), the fused update reads Z and writes S (16; S0 is a scalar
        # operand at T = 1), the payoff reduction reads S (8) = 32.  SURVEY.md 8(d) prices the reference's MATERIALISED plan at
        # (32 T + 8) = 40 B/sample (it also reads a resident S vector); that figure is reported beside, it is not what the roofline
        # fraction is taken on (round-3 review: a fraction on bytes the kernels do not move is not a roofline fraction).
        bytes_moved = 32 * T * M
        bytes_materialised = (32 * T + 8) * M
        return {
            "metric": "Monte-Carlo samples/s (1e8-sample randn + fused elementwise + sum reduction)",
            "value": round(M * T / (ms * 1e-3), 1), "unit": "samples/s", "ms_per_step": round(ms, 4), "scaling": "strong",
            "dtype": "f64",
            "config": {"workload": "monte-carlo-analysis f64, M=1e8, T=1, CPU-parity LCG randn stream", "price": price,
                       "algorithmic_bytes": bytes_moved, "materialised_plan_bytes_survey_8d": bytes_materialised,
                       "parallelism": f"sample ranges x{world}, ordered 1-value exchange"},
            "roofline": roofline("hbm", bytes_moved / world / (ms * 1e-3) / 1e9, 1, traffic=pmc_traffic("mc"),
                                 traffic_source=PMC_TRAFFIC_SOURCE + " (per step)",
                                 frac_on_32B_moved=round(bytes_moved / world / (ms * 1e-3) / 1e9 / hbm_peak, 4),
                                 frac_on_40B_survey_8d_plan=round(bytes_materialised / world / (ms * 1e-3) / 1e9 / hbm_peak, 4),
                                 kernel="k_rng_normal + rm_ew_fast + rm_red_contig (whole step, wall clock; 32 B per sample moved)",
                                 **({"per_kernel": mc_kernel_rooflines(M // world)} if args.workload == "mc" else {})),
        }

    def mc_lazy_record(steps, warmup):
        return mc_record(steps, warmup, lazy=True)

    def mc_evolved_record(steps, warmup):
        """Benchmark-shaped secondary of SURVEY.md 8(d) config 4: M = 1e6 paths, T = 256 steps, the whole time
        loop as ONE `stochastic_evolution` call (state in registers) + one fused payoff reduction."""
        from planner_requests import monte_carlo_shaders

        M, T = 1_000_000 // shrink, 256
        payoff = monte_carlo_shaders(100.0)[1]
        price = 0.0
        for _ in range(warmup):
            price, _ = sh.monte_carlo_price_evolved(prov, group, M, T, rng_state=0x9E3779B97F4A7C15, payoff_shader=payoff)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            price, _ = sh.monte_carlo_price_evolved(prov, group, M, T, rng_state=0x9E3779B97F4A7C15, payoff_shader=payoff)
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = wall / steps * 1e3
        # per normal: LCG step + half a Box-Muller (log, sqrt, sincos) + exp + 3 mul/add: ~150 fp64 VALU
        # instructions per sample-step (counted from the ISA); the kernel is VALU bound, not HBM bound
        return {
            "metric": "Monte-Carlo sample-steps/s (M=1e6 paths x T=256 steps, one stochastic_evolution call)",
            "value": round(M * T / (ms * 1e-3), 1), "unit": "samples/s", "ms_per_step": round(ms, 4), "scaling": "strong",
            "dtype": "f64",
            "config": {"workload": "monte-carlo-analysis f64, M=1e6, T=256, CPU-parity LCG randn stream, fused time loop",
                       "price": price, "algorithmic_bytes": 32 * M,
                       "parallelism": f"sample ranges x{world}, ordered 1-value exchange"},
            "roofline": valu_roofline("mc_evolved", "k_stochastic_evolution", ms * 1e-3,
                                      kernel="k_stochastic_evolution (state in registers: 24 B per path for the whole time loop, fp64 VALU bound)",
                                      hbm_frac_for_completeness=round(24 * M / world / (ms * 1e-3) / 1e9 / hbm_peak, 4),
                                      traffic=pmc_traffic("mc_evolved"), traffic_source=PMC_TRAFFIC_SOURCE + " (per step)"),
        }

    def image_record(steps, warmup):
        """benchmarks/4k-image-processing in f64: 16 frames of 2160 x 3840, per-frame mean / variance normalisation,
        gain, bias, clamp, gamma (the ImageNormalize fusion pattern = ONE provider call) -- frames sharded over ranks."""
        B, H, W = max(1, 16 // world), 2160 // shrink, 3840 // shrink
        hx = prov.fill_uniform(41 + 100 * rank, 0.0, 1.0, (B, H, W))

        def step():
            prov.free(prov.image_normalize(hx, B, H, W, 1e-6, gain=1.0123, bias=-0.02, gamma=1.8, clamp_zero=True))

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        prov.free(hx)
        ms = wall / steps * 1e3
        nbytes = 24 * B * H * W  # one-pass plane statistics (one read), normalise (one read + one write) per element
        return {
            "metric": "image_normalize GB/s (4k-image-processing: 16 x 2160 x 3840 f64 frames, one provider call)",
            "value": round(nbytes * world / (ms * 1e-3) / 1e9, 1), "unit": "GB/s", "ms_per_step": round(ms, 4), "scaling": "strong",
            "dtype": "f64",
            "config": {"workload": "benchmarks/4k-image-processing f64, image_normalize(gain, bias, clamp, gamma = 1.8)",
                       "bytes_per_step_per_gpu": nbytes, "parallelism": f"frames x{world}, no collective"},
            "roofline": {**roofline("hbm", nbytes / (ms * 1e-3) / 1e9, 1),
                         "traffic": pmc_traffic("image"), "traffic_source": PMC_TRAFFIC_SOURCE + " (per step)",
                         "valu": pmc_valu("image", "k_imgnorm_apply"),
                         "kernel": "k_plane_moments (one-pass mean / M2, fixed-order Chan merge), k_plane_moments_final, k_imgnorm_apply "
                                   "(24 B per element; round 1 moved 32 with a two-pass variance)"},
        }

    def mldivide_record(steps, warmup):
        """BASELINE configs[4] at the single-GPU size: x = A\\b, 16384x16384 f64, blocked recursive LU."""
        nn = max(16384 // shrink, 512 * world)
        ha = prov.fill_uniform(31, -1.0, 1.0, (nn, nn))
        ones = prov.ones((nn, 1))
        hb = prov.matmul(ha, ones)  # b = A*1  => x = 1
        cyclic = world > 1 or os.environ.get("RMHIP_BENCH_FORCE_CYCLIC") == "1"
        form = {"name": "single GPU, blocked LU with look-ahead (solve path: diagonal-domain pivoting + multiplier check)"}
        if cyclic:
            # multi-GPU: BASELINE configs[4]'s row-partitioned form (runmat_amd/sharding.py mldivide_row_partitioned: every rank keeps
            # the row blocks of [A | b] it owns, pivoting inside the owner's rows, one tile-row broadcast per panel, the last `world`
            # blocks all-gathered); if its multiplier guard refuses the matrix, the block-column cyclic form with the grid-wide rule.
            # Every rank builds the same A and keeps only what it owns.
            nbk = int(os.environ.get("RMHIP_BENCH_RB", "512"))  # row-block height = panel width of the row-partitioned form (developer knob)
            blocks = sh.owned_blocks(nn, nbk, group)
            row_blocks = sh.owned_row_blocks(nn, nbk, group)
            form["name"] = (f"row-partitioned x{world}, rb={nbk}: diagonal-domain pivoting, one tile-row broadcast per panel (rmhip_comm_bcast), "
                            f"last {world} row blocks all-gathered"
                            + ("; driver inside the library (look-ahead 1)" if world == 1 or group.native is not None else "; Python driver over the control plane"))

            def solve_rows():
                nloc = sum(min(nbk, nn - q * nbk) for q in row_blocks)
                ab = prov.zeros((max(nloc, 1), nn + 1))
                for q in row_blocks:
                    h = min(nbk, nn - q * nbk)
                    lo = sh.local_row_offset(q, nbk, group)
                    for src, c0, wd in ((ha, 0, nn), (hb, nn, 1)):
                        blk = prov.blk_copy((src, q * nbk, 0, h, wd))
                        prov.blk_assign((ab, lo, c0, h, wd), blk)
                        prov.free(blk)
                try:
                    if world == 1 or group.native is not None:
                        # the driver inside the library (csrc/sharded.cpp: depth-1 look-ahead, asynchronous panel broadcasts, panels on
                        # the solve path's kernels); its multiplier guard reports through the error code
                        try:
                            return prov.mldivide_row_partitioned(ab, nn, 1, rb=nbk)
                        except Exception as e:  # noqa: BLE001 - ProviderError carrying RMHIP_ERR_GROWTH on every rank alike
                            if "multiplier" in str(e) or "growth" in str(e).lower() or "failed" in str(e):
                                raise sh.PivotGrowth(str(e)) from e
                            raise
                    return sh.mldivide_row_partitioned(prov, group, ab, nn, 1, rb=nbk)
                finally:
                    prov.free(ab)

            def solve_cols():
                ncl = sum(min(nbk, nn - p * nbk) for p in blocks)
                a_loc = prov.zeros((nn, max(ncl, 1)))
                for p in blocks:
                    wp = min(nbk, nn - p * nbk)
                    blk = prov.blk_copy((ha, 0, p * nbk, nn, wp))
                    prov.blk_assign((a_loc, 0, sh.local_col_offset(p, nbk, group), nn, wp), blk)
                    prov.free(blk)
                x = sh.mldivide_block_cyclic(prov, group, a_loc, nn, hb, nb=nbk)
                prov.free(a_loc)
                return x

            def solve():
                if not form.get("cols"):
                    try:
                        return solve_rows()
                    except sh.PivotGrowth:  # raised on every rank alike (the guard is one exchange)
                        form["cols"] = True
                        form["name"] = f"block-column cyclic x{world}, nb=512, one panel broadcast per block (rmhip_comm_bcast, depth-1 look-ahead)"
                return solve_cols()
        else:
            def solve():
                return prov.mldivide(ha, hb)
        err = None
        quality = None
        for _ in range(max(1, warmup)):
            hx = solve()
            err = float(np.max(np.abs(prov.download(hx) - 1.0)))
            if quality is None and not cyclic:
                # What the forward error may be for THIS matrix (round-5 review: a flat 1e-7 says little): the normwise backward error
                # of the computed x, and a lower bound of cond_inf(A) from one more solve - A z = e with e = +-1 gives
                # ||A^-1||_inf >= ||z||_inf - so that bound ~ cond x backward error.  Device ops outside the timed region.
                try:
                    tmp = []
                    keep = lambda h: (tmp.append(h), h)[1]  # noqa: E731
                    r = keep(prov.elem_sub(keep(prov.matmul(ha, hx)), hb))
                    rmax = prov.read_scalar(keep(prov.reduce_max(keep(prov.unary_abs(r)))), 0)
                    anorm = prov.read_scalar(keep(prov.reduce_max(keep(prov.reduce_sum_dim(keep(prov.unary_abs(ha)), 1)))), 0)
                    xnorm = prov.read_scalar(keep(prov.reduce_max(keep(prov.unary_abs(hx)))), 0)
                    bnorm = prov.read_scalar(keep(prov.reduce_max(keep(prov.unary_abs(hb)))), 0)
                    e = keep(prov.upload(np.where(np.arange(nn) % 3 == 0, -1.0, 1.0).reshape(nn, 1)))
                    z = keep(prov.mldivide(ha, e))
                    znorm = prov.read_scalar(keep(prov.reduce_max(keep(prov.unary_abs(z)))), 0)
                    for h in tmp:
                        prov.free(h)
                    bwd = rmax / (anorm * xnorm + bnorm)
                    quality = {"normwise_backward_error_inf": float(f"{bwd:.3e}"), "cond_inf_lower_bound": float(f"{anorm * znorm:.3e}"),
                               "forward_error_estimate": float(f"{anorm * znorm * bwd:.3e}"),
                               "note": "bound ~ cond x backward error; cond from one extra solve A z = (+-1): a lower bound"}
                except Exception as ex:  # noqa: BLE001 - the quality block never costs the record
                    quality = {"error": f"{type(ex).__name__}: {ex}"[:160]}
            prov.free(hx)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            prov.free(solve())
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = wall / steps * 1e3
        flops = (2.0 / 3.0) * nn ** 3 + 2.0 * nn * nn
        for h in (ha, ones, hb):
            prov.free(h)
        phases = None
        if cyclic and not form.get("cols"):
            try:  # device time by phase of the LAST solve on this rank (rmhip_rp_phase_ms): what a measured SCALE run is laid against
                phases = {k: round(v, 3) for k, v in prov.rp_phase_ms().items()}
            except Exception:  # noqa: BLE001
                phases = None
        return {
            "phases_ms_rank0": phases,
            "metric": "fp64 GFLOP/s (x = A\\b, 16384x16384, blocked LU)",
            "value": round(flops / (ms * 1e-3) / 1e9, 1), "unit": "GFLOP/s", "ms_per_step": round(ms, 3), "scaling": "strong",
            "dtype": "f64",
            "config": {"workload": "x=A\\b 16384x16384 f64 via rmhip_mldivide, A=U(-1,1), b=A*1", "flops_per_step": flops,
                       "max_abs_err_vs_ones": err,
                       # forward-error bounds by generator (tests/test_gpu_lookahead.py): U(-1,1) as here - cond ~ 1e5 at this order - 1e-7;
                       # SURVEY.md 8(d)'s 1e-9 belongs to the diagonally dominant U(-1,1) + n*I generator
                       "max_abs_err_bound": {"generator": "U(-1,1) (this run)", "bound": 1e-7, "diagonally_dominant_U_plus_nI_bound": 1e-9},
                       **({"solution_quality": quality} if quality else {}),
                       **({"row_partitioned_phases_ms_rank0": phases} if phases else {}),
                       "parallelism": form["name"]},
            "roofline": {**roofline("mfma", flops / (ms * 1e-3) / 1e12, 3),
                         "traffic": pmc_traffic("mldivide"), "traffic_source": PMC_TRAFFIC_SOURCE + " (per solve, fabric side)",
                         "kernel": "k_rp_top / k_rp_below panels + k_dgemm_w8 trailing updates (whole solve, wall clock)"},
        }

    def chain_record(steps, warmup):
        """BASELINE configs[0] (the reference's CPU-runnable case) on the GPU: the 14-op
        elementwise-math chain (benchmarks/elementwise-math/runmat.m:10-13, f64) as ONE fused kernel
        over a 1024x1024 tensor; constants arrive as 1-element inputs like the planner sends them."""
        from planner_requests import elementwise_math_plan

        m = 1024
        plan, out_id = elementwise_math_plan()
        shader = plan.generate_wgsl_for_output(out_id, "f64")
        hx = prov.upload(np.linspace(0.0, 4.0 * np.pi, m * m), (m, m))
        consts = [prov.upload(np.array([v]), (1, 1)) for v in (10.0, 4.0, 0.25, 2.0, 0.1)]

        def step():
            prov.free(prov.fused_elementwise(shader, [hx] + consts, (m, m), m * m))

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = wall / steps * 1e3
        # kernel time: HIP events on the library's stream around a back-to-back run (the host issues a call in less time than the kernel
        # runs, so the stream stays full and elapsed / steps is the launch duration plus the dispatch gap - an upper bound of the former)
        prov.synchronize()
        prov.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = max_over_ranks(prov.timer_end() / steps)
        for h in [hx] + consts:
            prov.free(h)
        nbytes = 16 * m * m  # fused form: one read + one write per element
        return {
            "metric": "fused elementwise GB/s (elementwise-math 14-op chain, 1024x1024 f64)",
            "value": round(world * nbytes / (ms * 1e-3) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(ms, 5),
            "scaling": "weak", "dtype": "f64",
            "config": {"workload": "benchmarks/elementwise-math chain, 1024x1024 f64, one fused kernel",
                       "bytes_per_step_per_gpu": nbytes, "parallelism": f"independent x{world}"},
            # 14 fp64 ops per element with five transcendentals (sin, exp, cos, tanh, pow 2 -> x*x): ~2 us of HBM time against a kernel of
            # ~16 us - the bound is the fp64 VALU (16 lanes per clock and SIMD), not HBM and not the launch
            "roofline": valu_roofline("chain", "rm_ew_fast", kern_ms * 1e-3,
                                      kernel="rm_ew_fast (14-op chain, 1024 x 1024: fp64 VALU bound)", kernel_ms=round(kern_ms, 5),
                                      hbm_frac_for_completeness=round(nbytes / (kern_ms * 1e-3) / 1e9 / hbm_peak, 4),
                                      host_ms_per_call=round(ms, 5), traffic=pmc_traffic("chain", "rm_ew_fast"), traffic_source=PMC_TRAFFIC_SOURCE),
        }

    def bcast_record(steps, warmup):
        """The reference's UNFUSED implicit-expansion path (north_star: "broadcast"): A (8192 x 1) .* B (1 x 8192) through the call sequence
        of its times builtin (math/elementwise/times.rs:501-543): broadcast_reps -> repmat each operand -> elem_mul -> free the
        expansions.  `repmat` is a zero-copy view here and elem_mul reads it in place, so a step must move one 512 MiB write."""
        ha, hb = prov.fill_uniform(31 + rank, -1.0, 1.0, (n, 1)), prov.fill_uniform(32 + rank, -1.0, 1.0, (1, n))

        def step():
            le, re = prov.repmat(ha, [1, n]), prov.repmat(hb, [n, 1])
            h = prov.elem_mul(le, re)
            prov.free(le)
            prov.free(re)
            prov.free(h)

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = wall / steps * 1e3
        prov.synchronize()
        prov.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = max_over_ranks(prov.timer_end() / steps)
        prov.free(ha)
        prov.free(hb)
        nbytes = 8 * n * n + 16 * n  # the product written once, the two vectors read once
        return {
            "metric": "unfused implicit expansion GB/s (A(8192x1) .* B(1x8192) via repmat -> elem_mul -> free, f64)",
            "value": round(world * nbytes / (ms * 1e-3) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(ms, 5), "scaling": "weak", "dtype": "f64",
            "config": {"workload": "times builtin's unfused broadcast sequence on resident operands (repmat views + rmhip_binary)",
                       "bytes_per_step_per_gpu": nbytes, "parallelism": f"independent x{world}"},
            "roofline": roofline("hbm", nbytes / (kern_ms * 1e-3) / 1e9, traffic=pmc_traffic("bcast", "k_bcast2"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="k_bcast2<double, mul> (both operands stride-0 views: write-only traffic)", kernel_ms=round(kern_ms, 5)),
        }

    def fft_record(steps, warmup):
        """fft(X) of the 8192 x 8192 real operand along its columns (widened surface: lib.rs:2622, fft.hip): every line one pass over HBM -
        the 8 B/element input read once, the 16 B/element complex result written once."""
        h = prov.fill_uniform(41 + rank, -1.0, 1.0, (n, n))

        def step():
            prov.free(prov.fft_dim(h, None, 0))

        for _ in range(warmup):
            step()
        prov.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        prov.synchronize()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        ms = wall / steps * 1e3
        prov.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = max_over_ranks(prov.timer_end() / steps)
        prov.free(h)
        nbytes = 24 * n * n
        return {
            "metric": "fft GB/s (fft(X, [], 1) of an 8192x8192 real f64 matrix, complex-interleaved result)",
            "value": round(world * nbytes / (ms * 1e-3) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(ms, 5), "scaling": "weak", "dtype": "f64",
            "config": {"workload": "fft_dim along dimension 0 of 8192x8192 f64 (8192-point lines, one workgroup per line)", "bytes_per_step_per_gpu": nbytes,
                       "flops_per_step": 5.0 * n * n * 13, "parallelism": f"independent x{world}"},
            "roofline": roofline("hbm", nbytes / (kern_ms * 1e-3) / 1e9, traffic=pmc_traffic("fft", "k_fft_tile"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="k_fft_tile<1024> (Stockham radix-8 passes in LDS; first pass loads, last pass stores)",
                                 kernel_ms=round(kern_ms, 5)),
        }

    def fused_f32_record(steps, warmup):
        # the same request on a precision-32 provider (f32 in HBM, f64 arithmetic in registers)
        p32 = HipProvider(local_rank, precision="F32")
        plan, out_id = sin_mul_add_plan()
        shader = plan.generate_wgsl_for_output(out_id, "f32")
        base = 100 * rank
        hs = [p32.fill_uniform(s + base, lo, hi, (n, n)) for s, lo, hi in ((1, -np.pi, np.pi), (2, -1.0, 1.0), (3, -1.0, 1.0))]

        def step():
            p32.free(p32.fused_elementwise(shader, hs, (n, n), n * n))

        for _ in range(warmup):
            step()
        p32.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        p32.synchronize()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        p32.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = p32.timer_end() / steps
        p32.close()
        ms = wall / steps * 1e3
        nbytes = 4 * 4 * n * n
        achieved = nbytes / (kern_ms * 1e-3) / 1e9
        return {
            "metric": "fused elementwise GB/s (D = sin(A).*B + C, 8192x8192, f32 storage on a precision-32 provider)",
            "value": round(world * nbytes / (ms * 1e-3) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(ms, 5),
            "scaling": "weak", "dtype": "f32 storage, f64 arithmetic",
            "config": {"workload": "fused D=sin(A).*B+C 8192x8192 via rmhip_fused_elementwise (f32 WGSL request)",
                       "bytes_per_step_per_gpu": nbytes, "elements_per_s": round(world * n * n / (ms * 1e-3), 1),
                       "parallelism": f"independent x{world}"},
            "roofline": roofline("hbm", achieved, traffic=pmc_traffic("fused_f32", "rm_ew_fast"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="rm_ew_fast (f32 variant: 16-byte vectors of four, body in f64)", kernel_ms=round(kern_ms, 5)),
        }

    def sgemm_record(steps, warmup):
        # precision-32 provider: C = A*B on the f32 matrix cores (sgemm.hip), row-block sharded like the f64 workload
        p32 = HipProvider(local_rank, precision="F32")
        rows = n // world
        ha = p32.fill_uniform(11 + 1000 * rank, -1.0, 1.0, (rows, n))
        hb = p32.fill_uniform(12, -1.0, 1.0, (n, n))

        def step():
            p32.free(p32.matmul(ha, hb))

        for _ in range(warmup):
            step()
        p32.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        p32.synchronize()
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        p32.timer_begin()
        for _ in range(steps):
            step()
        kern_ms = p32.timer_end() / steps
        p32.close()
        ms = wall / steps * 1e3
        achieved = (dgemm_flops / world) / (kern_ms * 1e-3) / 1e12
        return {
            "metric": "fp32 GFLOP/s (8192^3 matmul on a precision-32 provider, row-block sharded across GPUs)",
            "value": round(dgemm_flops / (ms * 1e-3) / 1e9, 1), "unit": "GFLOP/s", "ms_per_step": round(ms, 5),
            "scaling": "strong", "dtype": "f32 (f32 MFMA accumulation)",
            "config": {"workload": "C=A*B 8192x8192x8192 f32 storage via rmhip_matmul", "flops_per_step": dgemm_flops,
                       "parallelism": f"row-block x{world}, B replicated, no collective"},
            "roofline": roofline("mfma_f32", achieved, 3, traffic=pmc_traffic("sgemm", "k_sgemm_w8"), traffic_source=PMC_TRAFFIC_SOURCE,
                                 kernel="k_sgemm_w8 (eight waves, pipelined k loop; v_mfma_f32_16x16x4_f32)", kernel_ms=round(kern_ms, 5)),
        }

    def agreed(ok: bool) -> bool:
        """Every rank reports whether its last workload went through; a failure anywhere fails it everywhere, so that the ranks stay in
        step for the next workload's collectives (control plane, one small all-reduce)."""
        if dist is None:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def safe_record(name, fn, steps, warmup):
        """One workload; an exception becomes an "error" record instead of a dead run (a missing workload costs the line one entry)."""
        rec, why = None, ""
        try:
            if os.environ.get("RMHIP_BENCH_TEST_FAIL") == name:  # test hook (every rank): the error path of this function
                raise RuntimeError("forced by RMHIP_BENCH_TEST_FAIL")
            rec = fn(steps, warmup)
        except Exception as e:  # noqa: BLE001
            why = f"{type(e).__name__}: {e}"[:300]
            try:
                prov.synchronize()
            except Exception:  # noqa: BLE001
                pass
        if not agreed(rec is not None):
            return {"workload": name, "error": why or "failed on another rank"}
        return rec

    records = {"mc_lazy": mc_lazy_record, "sgemm": sgemm_record, "fused_f32": fused_f32_record, "fused": fused_record, "dgemm": dgemm_record, "mc": mc_record, "mldivide": mldivide_record,
               "chain": chain_record, "mc_evolved": mc_evolved_record, "image": image_record, "bcast": bcast_record, "fft": fft_record}
    primary = records[args.workload]
    rec = safe_record(args.workload, primary, args.steps, args.warmup)
    # Developer knob, OFF by default (round-5 advisor finding: an untimed loop that exists to move an external utilisation signal
    # measures nothing): RMHIP_BENCH_BUSY_S=<seconds> keeps the device busy with untimed headline launches, e.g. while watching clocks.
    busy_s = float(os.environ.get("RMHIP_BENCH_BUSY_S", "0"))
    if busy_s > 0 and "error" not in rec:
        try:
            plan_b, out_b = sin_mul_add_plan()
            shader_b = plan_b.generate_wgsl_for_output(out_b, "f64")
            hb_in = [prov.fill_uniform(71 + i, -1.0, 1.0, (n, n)) for i in range(3)]
            t_end = time.perf_counter() + busy_s
            launches = 0
            while time.perf_counter() < t_end:
                for _ in range(200):
                    prov.free(prov.fused_elementwise(shader_b, hb_in, (n, n), n * n))
                prov.synchronize()
                launches += 200
            for h in hb_in:
                prov.free(h)
            rec["config"] = dict(rec["config"], untimed_busy_loop=f"{launches} fused launches over {busy_s:.0f} s after the timed region")
        except Exception:  # noqa: BLE001 - never costs the line
            pass
        barrier()
    if "error" in rec:  # the line still comes out, with the reason where the number would be
        rec = {"metric": f"{args.workload} (failed)", "value": None, "unit": "", "ms_per_step": None, "scaling": "weak", "dtype": "f64",
               "config": {"workload": args.workload}, "roofline": None, "error": rec["error"]}
    out = {
        "metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": rec["scaling"],
        "vs_baseline": None, "dtype": rec["dtype"], "data": "synthetic", "config": rec["config"],
        "roofline": rec["roofline"],
    }
    if "error" in rec:
        out["error"] = rec["error"]
    if rec.get("roofline_slow_path"):
        out["roofline_slow_path"] = rec["roofline_slow_path"]
    out["config"] = dict(out["config"], collectives=comm_note, control_plane=control_plane)
    if shrink > 1:
        out["config"]["shrink"] = shrink  # NOT the benchmark: every linear size divided by this (test runs only)
    # what the data path actually ran on: the library's own view of the communicator (rmhip_comm_rank), not what was asked for
    comm = {"transport": "none", "world_seen": 1}
    if world > 1:
        try:
            if group.native is not None:
                r_seen, w_seen = prov.comm_rank()
                used = os.environ.get("RMHIP_BENCH_TRANSPORT", "rccl" if backend == "nccl" else "shm")
                comm = {"transport": "rccl" if used == "rccl" else "host-shm", "world_seen": int(w_seen), "rank_seen": int(r_seen)}
            else:
                comm = {"transport": f"torch.distributed/{backend}", "world_seen": int(dist.get_world_size())}
        except Exception as e:  # noqa: BLE001
            comm = {"transport": "unknown", "error": str(e)[:200]}
    out["comm"] = comm
    if not args.no_also:
        # the other configs of BASELINE.json, short runs; every rank takes part (collectives inside)
        # Order matters: the driver keeps the TAIL of this line, so the extras come first and BASELINE.json's own configs - the 1024^2
        # chain, the 1e8-sample Monte-Carlo, the 8192^3 dgemm, the 16384^2 solve - last (tests/test_bench_contract.py pins that they
        # sit inside the last 6000 characters).  bcast and fft are workloads of their own (--workload), not part of the default line.
        others = [w for w in ("image", "fused_f32", "sgemm", "mc_evolved", "mc_lazy", "fused", "chain", "mc", "dgemm") if w != args.workload]
        if args.workload != "mldivide":
            others.append("mldivide")  # one GPU: rmhip_mldivide; N > 1: the block-column cyclic driver (BASELINE configs[4])
        also = []
        for w in others:
            # enough steps that the two synchronisations around the timed region (~1 ms together) stay below 1 % of it: a 20-step run
            # of the 0.65 ms image workload read 0.73 ms per step
            steps = {"fused": 200, "dgemm": 10, "mc": 200, "mc_lazy": 200, "mc_evolved": 100, "image": 200, "mldivide": 3, "chain": 2000, "fused_f32": 200, "sgemm": 10, "bcast": 500, "fft": 100}[w]
            sec = safe_record(w, records[w], steps, 5 if w != "mldivide" else 1)
            if "error" in sec:
                also.append(sec)
            else:
                also.append({**{k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "scaling", "config", "roofline", "roofline_slow_path") if k in sec},
                             "steps": steps, "warmup": 5 if w != "mldivide" else 1})
        out["also"] = also
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = {"fused": cpu_baseline_fused, "dgemm": cpu_baseline_dgemm, "mc": cpu_baseline_mc,
                               "mldivide": cpu_baseline_mldivide, "chain": cpu_baseline_chain,
                               "mc_evolved": cpu_baseline_mc, "mc_lazy": cpu_baseline_mc, "image": cpu_baseline_fused,
                               "fused_f32": cpu_baseline_fused, "sgemm": cpu_baseline_dgemm, "bcast": cpu_baseline_fused, "fft": cpu_baseline_fft}[args.workload]()
        for a in out.get("also", []):
            if "error" in a:
                continue
            if a["unit"] == "GFLOP/s" and "matmul" in a["metric"] and a["metric"].startswith("fp64"):
                a["cpu_baseline"] = cpu_baseline_dgemm()
            elif a["unit"] == "samples/s" and "stochastic_evolution" not in a["metric"]:
                a["cpu_baseline"] = cpu_baseline_mc()
            elif "A\\b" in a["metric"]:
                a["cpu_baseline"] = cpu_baseline_mldivide()
            elif "14-op chain" in a["metric"]:
                a["cpu_baseline"] = cpu_baseline_chain()
    if "cpu_baseline" in out:
        out["cpu_baseline"].update(host_info())
    for a in out.get("also", []):
        if "cpu_baseline" in a:
            a["cpu_baseline"].update(host_info())
    info = prov.device_info_struct()
    out["device"] = {"arch": info["arch"], "compute_units": info["compute_units"], "clock_mhz": info["clock_mhz"],
                     "hbm_bytes": info["total_memory_bytes"]}
    # LAST key of the line: BASELINE.json's five configs in compact form, so that whatever keeps only the end of the line (the driver's
    # `tail` is 2 000 characters) still holds the dgemm half of BASELINE's metric.  tests/test_bench_contract.py pins <= 1 800 characters.
    out["baseline_configs"] = baseline_configs(out)
    prov.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sh._flush_c_stdio()  # anything native libraries buffered (RCCL's banner) goes out BEFORE the JSON line
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
