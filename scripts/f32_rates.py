"""Developer tool: precision-32 provider rates beside the f64 ones (same shapes, HIP-event timing on the library stream).
Bytes are the algorithmic bytes of each storage type."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan

N = 8192


def timed(p, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    p.timer_begin()
    for _ in range(reps):
        fn()
    return p.timer_end() / reps


def run(prec):
    p = HipProvider(0, precision=prec)
    eb = 4 if prec == "F32" else 8
    out = {}
    a = p.fill_uniform(1, -3.0, 3.0, (N, N))
    b = p.fill_uniform(2, -1.0, 1.0, (N, N))
    free = lambda h: p.free(h)  # noqa: E731
    n2 = N * N
    cases = {
        "unary_sin": (lambda: free(p.unary_sin(a)), 2 * eb * n2),
        "elem_add": (lambda: free(p.elem_add(a, b)), 3 * eb * n2),
        "scalar_mul": (lambda: free(p.scalar_mul(a, 0.5)), 2 * eb * n2),
        "sum_all": (lambda: free(p.reduce_sum(a)), eb * n2),
        "sum_dim0": (lambda: free(p.reduce_sum_dim(a, 0)), eb * n2),
        "sum_dim1": (lambda: free(p.reduce_sum_dim(a, 1)), eb * n2),
        "dot": (lambda: free(p.dot(a, b)), 2 * eb * n2),
    }
    row = p.fill_uniform(3, -1.0, 1.0, (1, N))
    cases["bcast_add_row"] = (lambda: free(p.elem_add(a, row)), 2 * eb * n2)
    for k, (fn, nbytes) in cases.items():
        ms = timed(p, fn)
        out[k] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}
    img = p.fill_uniform(4, 0.0, 1.0, (16, 2160, 3840))
    ms = timed(p, lambda: free(p.image_normalize(img, 16, 2160, 3840, 1e-6, gain=1.2, bias=0.05, gamma=1.8)), reps=5, warm=2)
    out["image_normalize_gamma"] = {"ms": round(ms, 4), "GBps": round(4 * eb * 16 * 2160 * 3840 / ms / 1e6, 1)}
    ms = timed(p, lambda: free(p.image_normalize(img, 16, 2160, 3840, 1e-6, gain=1.2, bias=0.05)), reps=5, warm=2)
    out["image_normalize"] = {"ms": round(ms, 4), "GBps": round(4 * eb * 16 * 2160 * 3840 / ms / 1e6, 1)}
    p.free(img)
    s0 = p.fill((1000000, 1), 100.0)
    ms = timed(p, lambda: free(p.stochastic_evolution(s0, 0.0002, 0.0126, 256)), reps=5, warm=2)
    out["stochastic_evolution_1e6x256"] = {"ms": round(ms, 4)}
    ms = timed(p, lambda: free(p.random_normal((100000000, 1))), reps=5, warm=2)
    out["randn_1e8"] = {"ms": round(ms, 4), "GBps": round(eb * 1e8 / ms / 1e6, 1)}
    ms = timed(p, lambda: free(p.matmul(a, b)), reps=3, warm=1)
    out["matmul_8192"] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * N ** 3 / ms / 1e9, 2)}
    if prec == "F32":
        os.environ["RMHIP_F32_MATMUL"] = "f64"
        ms = timed(p, lambda: free(p.matmul(a, b)), reps=3, warm=1)
        out["matmul_8192_f64_path"] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * N ** 3 / ms / 1e9, 2)}
        del os.environ["RMHIP_F32_MATMUL"]
        at = p.transpose(a)
        ms = timed(p, lambda: free(p.matmul(at, b)), reps=3, warm=1)
        out["matmul_8192_At_B"] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * N ** 3 / ms / 1e9, 2)}
        for sz in (1024, 2048, 4096):
            x = p.fill_uniform(7, -1.0, 1.0, (sz, sz))
            ms = timed(p, lambda: free(p.matmul(x, x)), reps=5, warm=2)
            out[f"matmul_{sz}"] = {"ms": round(ms, 4), "TFLOPs": round(2.0 * sz ** 3 / ms / 1e9, 2)}
            p.free(x)
    p.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["F64", "F32"]
    print(json.dumps({w: run(w) for w in which}, indent=1))
