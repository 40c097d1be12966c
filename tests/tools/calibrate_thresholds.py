"""Break-even sizes between this backend (operands resident on the device) and the reference's CPU path (the oracle on
one host core) for RunMat's auto-offload thresholds (crates/runmat-accelerate/src/native_auto.rs:55-81, env overrides
:1399-1444).  Prints one JSON object; profiles/r01_offload_calibration.json is a run of this on an MI355X box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from runmat_amd import HipProvider

prov = HipProvider(0)

def gpu_time(f, reps):
    for _ in range(3): prov.free(f())
    prov.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): prov.free(f())
    prov.synchronize()
    return (time.perf_counter() - t0) / reps

def cpu_time(f):
    f(); reps, t = 0, 0.0
    t0 = time.perf_counter()
    while t < 0.02:
        f(); reps += 1; t = time.perf_counter() - t0
    return t / reps

def sweep(name, sizes, make_gpu, make_cpu):
    rows, even = [], None
    for n in sizes:
        g, c = gpu_time(make_gpu(n), 200 if n < 1 << 18 else 30), cpu_time(make_cpu(n))
        rows.append({"n": n, "gpu_us": round(g * 1e6, 2), "cpu_us": round(c * 1e6, 2)})
        if even is None and g < c: even = n
    return {"op": name, "break_even": even, "points": rows}

out = {"device": prov.device_info_struct()["name"], "host": "one core, oracle/oracle.c (-O2, no FMA contraction)", "sweeps": []}
sizes = [1 << e for e in range(8, 23, 2)]
def unary_gpu(n):
    h = prov.upload(np.linspace(-3, 3, n).reshape(n, 1)); return lambda: prov.unary_sin(h)
def unary_cpu(n):
    x = np.linspace(-3, 3, n).reshape(n, 1); return lambda: oracle.unary("sin", x)
def bin_gpu(n):
    a = prov.upload(np.linspace(-3, 3, n).reshape(n, 1)); b = prov.upload(np.linspace(1, 2, n).reshape(n, 1)); return lambda: prov.elem_add(a, b)
def bin_cpu(n):
    a = np.linspace(-3, 3, n).reshape(n, 1); b = np.linspace(1, 2, n).reshape(n, 1); return lambda: oracle.binary("add", a, b)
def red_gpu(n):
    a = prov.upload(np.linspace(-3, 3, n).reshape(n, 1)); return lambda: prov.reduce_sum(a)
def red_cpu(n):
    a = np.linspace(-3, 3, n).reshape(n, 1); return lambda: oracle.reduce_sum(a, "all")
out["sweeps"].append(sweep("unary sin(x)", sizes, unary_gpu, unary_cpu))
out["sweeps"].append(sweep("binary a+b", sizes, bin_gpu, bin_cpu))
out["sweeps"].append(sweep("reduction sum(x,'all')", sizes, red_gpu, red_cpu))
mm = {"op": "matmul n x n x n", "break_even_flops": None, "points": []}
for n in (8, 16, 32, 48, 64, 96, 128, 192, 256):
    rng = np.random.default_rng(n); A, B = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))
    ha, hb = prov.upload(A), prov.upload(B)
    g, c = gpu_time(lambda: prov.matmul(ha, hb), 100), cpu_time(lambda: oracle.matmul(A, B))
    mm["points"].append({"n": n, "flops": n * n * n, "gpu_us": round(g * 1e6, 2), "cpu_us": round(c * 1e6, 2)})
    if mm["break_even_flops"] is None and g < c: mm["break_even_flops"] = n * n * n
out["sweeps"].append(mm)
be = {s["op"]: s.get("break_even", s.get("break_even_flops")) for s in out["sweeps"]}
out["recommended_env"] = {
    "RUNMAT_ACCEL_THRESHOLD_UNARY": be["unary sin(x)"], "RUNMAT_ACCEL_THRESHOLD_ELEMWISE": be["binary a+b"],
    "RUNMAT_ACCEL_THRESHOLD_REDUCTION": be["reduction sum(x,'all')"], "RUNMAT_ACCEL_THRESHOLD_MATMUL": be["matmul n x n x n"],
    "_note": "resident operands; the reference defaults are 4096 / 4096 / 256 elements and 1e6 flops (native_auto.rs:67-80)"}
print(json.dumps(out))
