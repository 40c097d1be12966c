"""Workloads written once against a tiny op-adapter so the SAME op sequence runs on the oracle
(numpy arrays, CPU restatement) and on the HipProvider (device handles, C ABI)."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


class OracleOps:
    """CPU side: each method is one reference builtin on the oracle."""

    def __init__(self, oracle):
        self.o = oracle

    def tensor(self, a):
        return np.asarray(a, dtype=np.float64)

    def unary(self, op, a):
        return self.o.unary(op, a)

    def binary(self, op, a, b):
        return self.o.binary(op, a, b)

    def scalar(self, op, a, s):
        s = np.array([[float(s)]])
        if op == "rsub":
            return self.o.binary("sub", s, a)
        if op == "rdiv":
            return self.o.binary("div", s, a)
        return self.o.binary(op, a, s)

    def mean_all(self, a):
        return float(self.o.reduce_sum(a, "all", mean=True).reshape(-1)[0])

    def sum_all(self, a):
        return float(self.o.reduce_sum(a, "all").reshape(-1)[0])

    def free(self, a):
        pass


class ProviderOps:
    """GPU side: the same ops through the C ABI (per-op kernels)."""

    def __init__(self, prov):
        self.p = prov

    def tensor(self, a):
        return self.p.upload(np.asarray(a, dtype=np.float64))

    def unary(self, op, a):
        return getattr(self.p, "unary_" + op)(a)

    def binary(self, op, a, b):
        return self.p._binary(op, a, b)

    def scalar(self, op, a, s):
        return self.p._scalar(op, a, s)

    def mean_all(self, a):
        return float(self.p.download(self.p.reduce_mean(a))[0])

    def sum_all(self, a):
        return float(self.p.download(self.p.reduce_sum(a))[0])

    def free(self, a):
        self.p.free(a)


def lcg_monte_carlo_price(ops, M, T, S0=100.0, mu=0.05, sigma=0.2, dt=1.0 / 252.0, K=100.0, seed=0, f32_constants=True):
    """benchmarks/monte-carlo-analysis/runmat_lcg.m:27-52 in f64 (drift/scale pre-rounded through
    f32 exactly as the reference scripts do; f32_constants=False: the all-f64 pipeline of tests/golden/reference_f64.json)."""
    if f32_constants:
        drift = float(np.float32((mu - 0.5 * sigma * sigma) * dt))
        scale = float(np.float32(sigma) * np.sqrt(np.float32(dt)))
    else:
        drift = float((mu - 0.5 * sigma * sigma) * dt)
        scale = float(sigma * np.sqrt(dt))
    rid = ops.tensor(np.arange(M, dtype=np.float64).reshape(M, 1))
    S = ops.tensor(np.full((M, 1), S0))
    two32 = ops.tensor(np.array([[4294967296.0]]))
    for t in range(T):
        salt = float(t) * 2.0 * M
        idx1 = ops.scalar("add", rid, salt + seed)
        idx2 = ops.scalar("add", rid, salt + M + seed)
        st1 = ops.binary("mod", ops.scalar("add", ops.scalar("mul", idx1, 1664525.0), 1013904223.0), two32)
        st2 = ops.binary("mod", ops.scalar("add", ops.scalar("mul", idx2, 1664525.0), 1013904223.0), two32)
        u1 = ops.scalar("max", ops.scalar("div", st1, 4294967296.0), 1.0 / 4294967296.0)
        u2 = ops.scalar("div", st2, 4294967296.0)
        r = ops.unary("sqrt", ops.scalar("mul", ops.unary("log", u1), -2.0))
        theta = ops.scalar("mul", u2, 2.0 * np.pi)
        z = ops.binary("mul", r, ops.unary("cos", theta))
        S = ops.binary("mul", S, ops.unary("exp", ops.scalar("add", ops.scalar("mul", z, scale), drift)))
    payoff = ops.scalar("max", ops.scalar("sub", S, K), 0.0)
    return ops.mean_all(payoff) * float(np.exp(-mu * T * dt))


def golden_monte_carlo_cases():
    return json.loads((GOLDEN / "monte_carlo_lcg.json").read_text())["cases"]


def golden_reference_f64():
    """The comparators forced to f64 (tests/golden/make_golden.py main_f64): chain samples and LCG Monte-Carlo prices at full precision."""
    return json.loads((GOLDEN / "reference_f64.json").read_text())


def golden_image_cases():
    return json.loads((GOLDEN / "image_normalize_lcg.json").read_text())


def golden_elementwise_math():
    return json.loads((GOLDEN / "elementwise_math.json").read_text())


def lcg_image_field(B, H, W, seed=0):
    """benchmarks/4k-image-processing/runmat_lcg.m (and python_numpy_lcg.py): imgs(b,y,x) = single(mod(1664525*idx +
    1013904223, 2^32)) / 2^32 with idx = b*H*W + y*W + x + seed; returned as f64 holding the f32-rounded values,
    shape [B, H, W]."""
    b = np.arange(B, dtype=np.uint64)[:, None, None]
    y = np.arange(H, dtype=np.uint64)[None, :, None]
    x = np.arange(W, dtype=np.uint64)[None, None, :]
    idx = b * np.uint64(H * W) + y * np.uint64(W) + x + np.uint64(seed)
    state = (np.uint64(1664525) * idx + np.uint64(1013904223)) % np.uint64(1 << 32)
    return (state.astype(np.float32) / np.float32(2.0 ** 32)).astype(np.float64)
