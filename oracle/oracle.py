"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY (see oracle.c header): importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from runmat_amd/.
All arrays are column-major f64; matrices are passed as numpy arrays and flattened in Fortran order.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "liboracle.so"

UNARY = {n: i for i, n in enumerate((
    "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh", "exp", "expm1",
    "log", "log2", "log10", "log1p", "sqrt", "abs", "sign", "floor", "ceil", "round", "fix", "neg", "exp2",
    "heaviside", "isnan", "isinf", "isfinite", "uplus", "single", "double", "erf", "sinc", "not",
    "gamma", "factorial", "nextpow2", "gammaln", "erfcinv", "nan_to_zero", "not_nan"))}
BINARY = {"add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "max": 5, "min": 6, "hypot": 7, "atan2": 8, "mod": 9,
          "rem": 10, "eq": 11, "ne": 12, "lt": 13, "le": 14, "gt": 15, "ge": 16, "and": 17, "or": 18, "xor": 19}

_DP = C.POINTER(C.c_double)
_SZP = C.POINTER(C.c_size_t)
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        l = C.CDLL(str(LIB_PATH))
        l.orc_matmul.restype = C.c_int
        l.orc_matmul.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, C.c_size_t, C.c_size_t, _DP]
        l.orc_matmul_epilogue.restype = C.c_int
        l.orc_matmul_epilogue.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                          _DP, C.c_int, _DP, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int,
                                          C.c_double, _DP, _DP]
        l.orc_unary.restype = C.c_int
        l.orc_unary.argtypes = [C.c_int, _DP, C.c_size_t, _DP]
        l.orc_binary.restype = C.c_int
        l.orc_binary.argtypes = [C.c_int, _DP, _SZP, C.c_size_t, _DP, _SZP, C.c_size_t, _DP, _SZP, _SZP]
        l.orc_broadcast_shape.restype = C.c_size_t
        l.orc_broadcast_shape.argtypes = [_SZP, C.c_size_t, _SZP, C.c_size_t, _SZP]
        l.orc_sum.restype = C.c_int
        l.orc_sum.argtypes = [_DP, _SZP, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.c_int, _DP]
        l.orc_rng_default_seed.restype = C.c_uint64
        l.orc_rng_mix_seed.restype = C.c_uint64
        l.orc_rng_mix_seed.argtypes = [C.c_uint64]
        l.orc_rng_advance.restype = C.c_uint64
        l.orc_rng_advance.argtypes = [C.c_uint64, C.c_uint64]
        l.orc_rng_uniform.restype = None
        l.orc_rng_uniform.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, _DP]
        l.orc_rng_normal.restype = None
        l.orc_rng_normal.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, _DP]
        l.orc_lu.restype = C.c_int
        l.orc_lu.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, _DP, _DP, _DP, _DP]
        l.orc_mldivide_svd.restype = C.c_int
        l.orc_mldivide_svd.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, C.c_size_t, C.c_size_t, _DP]
        l.orc_mldivide_lu.restype = C.c_int
        l.orc_mldivide_lu.argtypes = [_DP, C.c_size_t, _DP, C.c_size_t, _DP]
        l.orc_stochastic_evolution.restype = C.c_int
        l.orc_stochastic_evolution.argtypes = [C.POINTER(C.c_uint64), _DP, C.c_size_t, C.c_double, C.c_double, C.c_uint32]
        l.orc_linsolve_tri.restype = C.c_int
        l.orc_linsolve_tri.argtypes = [C.c_int, _DP, C.c_size_t, _DP, C.c_size_t, _DP, _DP]
        l.orc_image_normalize.restype = None
        l.orc_image_normalize.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_double, C.c_int,
                                          C.c_double, C.c_int, C.c_int, C.c_double, _DP]
        l.orc_matmul_power_step.restype = C.c_int
        l.orc_matmul_power_step.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, C.c_size_t, C.c_size_t, C.c_double, _DP]
        l.orc_covariance.restype = None
        l.orc_covariance.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_int, _DP]
        l.orc_syrk.restype = None
        l.orc_syrk.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP]
        l.orc_transpose.restype = None
        l.orc_transpose.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP]
        l.orc_sin_mul_add.restype = C.c_int
        l.orc_sin_mul_add.argtypes = [_DP, _DP, _DP, C.c_size_t, _DP]
        l.orc_elementwise_math_chain.restype = C.c_int
        l.orc_elementwise_math_chain.argtypes = [_DP, C.c_size_t, _DP]
        l.orc_monte_carlo_price.restype = C.c_double
        l.orc_monte_carlo_price.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                            C.c_double, C.c_double, C.c_double]
        l.orc_fill_uniform.restype = None
        l.orc_fill_uniform.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_size_t, _DP]
        _lib = l
    return _lib


def _f(a) -> np.ndarray:
    """Column-major flat f64 copy."""
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(a.reshape(-1, order="F"))


def _p(a: np.ndarray):
    return a.ctypes.data_as(_DP)


def _shape(s):
    return (C.c_size_t * max(len(s), 1))(*[int(x) for x in s])


def matmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fa, fb = _f(a), _f(b)
    out = np.empty(a.shape[0] * b.shape[1], dtype=np.float64)
    rc = lib().orc_matmul(_p(fa), a.shape[0], a.shape[1], _p(fb), b.shape[0], b.shape[1], _p(out))
    if rc:
        raise ValueError("Inner matrix dimensions must agree")
    return out.reshape((a.shape[0], b.shape[1]), order="F")


def matmul_epilogue(a, b, alpha=1.0, beta=0.0, row_scale=None, col_scale=None, row_op="multiply", col_op="multiply",
                    clamp_min=None, clamp_max=None, pow_exponent=None, diag=False):
    """Returns (C, diag or None)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    fa, fb = _f(a), _f(b)
    out = np.empty(a.shape[0] * b.shape[1], dtype=np.float64)
    rs = _f(row_scale) if row_scale is not None else None
    cs = _f(col_scale) if col_scale is not None else None
    dg = np.zeros(min(a.shape[0], b.shape[1])) if diag else None
    rc = lib().orc_matmul_epilogue(_p(fa), a.shape[0], a.shape[1], _p(fb), b.shape[0], b.shape[1], alpha, beta,
                                   _p(rs) if rs is not None else None, int(row_op == "divide"),
                                   _p(cs) if cs is not None else None, int(col_op == "divide"),
                                   int(clamp_min is not None), float(clamp_min or 0.0), int(clamp_max is not None),
                                   float(clamp_max or 0.0), int(pow_exponent is not None), float(pow_exponent or 0.0),
                                   _p(dg) if dg is not None else None, _p(out))
    if rc:
        raise ValueError("Inner matrix dimensions must agree")
    return out.reshape((a.shape[0], b.shape[1]), order="F"), dg


def unary(op: str, x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    fx = _f(x)
    out = np.empty_like(fx)
    if lib().orc_unary(UNARY[op], _p(fx), fx.size, _p(out)):
        raise ValueError(op)
    return out.reshape(x.shape, order="F")


def binary(op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    sa, sb = _shape(a.shape), _shape(b.shape)
    oshape = (C.c_size_t * 16)()
    rank = lib().orc_broadcast_shape(sa, a.ndim, sb, b.ndim, oshape)
    if rank == C.c_size_t(-1).value:
        raise ValueError("size mismatch between inputs")
    shape = tuple(int(oshape[i]) for i in range(rank))
    out = np.empty(int(np.prod(shape, dtype=np.int64)), dtype=np.float64)
    fa, fb = _f(a), _f(b)
    orank = C.c_size_t()
    if lib().orc_binary(BINARY[op], _p(fa), sa, a.ndim, _p(fb), sb, b.ndim, _p(out), oshape, C.byref(orank)):
        raise ValueError("size mismatch between inputs")
    return out.reshape(shape, order="F")


def reduce_sum(x: np.ndarray, dims, omitnan: bool = False, mean: bool = False) -> np.ndarray:
    """dims: iterable of zero-based dims to reduce, or 'all'."""
    x = np.asarray(x, dtype=np.float64)
    mask = np.zeros(max(x.ndim, 1), dtype=np.int32)
    if dims == "all":
        mask[:] = 1
    else:
        for d in dims:
            mask[d] = 1
    oshape = tuple(1 if mask[d] else x.shape[d] for d in range(x.ndim))
    fx = _f(x)
    out = np.empty(int(np.prod(oshape, dtype=np.int64)) if oshape else 1, dtype=np.float64)
    rc = lib().orc_sum(_p(fx), _shape(x.shape), x.ndim, mask.ctypes.data_as(C.POINTER(C.c_int)), 1 if omitnan else 0,
                       1 if mean else 0, _p(out))
    if rc:
        raise ValueError("sum")
    return out.reshape(oshape, order="F")


def rng_default_seed() -> int:
    return int(lib().orc_rng_default_seed())


def rng_mix_seed(seed: int) -> int:
    return int(lib().orc_rng_mix_seed(seed))


def rng_advance(state: int, delta: int) -> int:
    return int(lib().orc_rng_advance(state, delta))


def rng_uniform(state: int, n: int):
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    lib().orc_rng_uniform(C.byref(s), n, _p(out))
    return out, int(s.value)


def rng_normal(state: int, n: int):
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    lib().orc_rng_normal(C.byref(s), n, _p(out))
    return out, int(s.value)


def rng_unifrnd(state: int, a: float, b: float, n: int):
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    l = lib()
    l.orc_rng_unifrnd.restype = None
    l.orc_rng_unifrnd.argtypes = [C.POINTER(C.c_uint64), C.c_double, C.c_double, C.c_size_t, _DP]
    l.orc_rng_unifrnd(C.byref(s), a, b, n, _p(out))
    return out, int(s.value)


def rng_exponential(state: int, mu: float, n: int):
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    l = lib()
    l.orc_rng_exponential.restype = None
    l.orc_rng_exponential.argtypes = [C.POINTER(C.c_uint64), C.c_double, C.c_size_t, _DP]
    l.orc_rng_exponential(C.byref(s), mu, n, _p(out))
    return out, int(s.value)


def rng_normrnd(state: int, mu: float, sigma: float, n: int):
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    l = lib()
    l.orc_rng_normrnd.restype = None
    l.orc_rng_normrnd.argtypes = [C.POINTER(C.c_uint64), C.c_double, C.c_double, C.c_size_t, _DP]
    l.orc_rng_normrnd(C.byref(s), mu, sigma, n, _p(out))
    return out, int(s.value)


def rng_integer_range(state: int, lower: int, upper: int, n: int):
    """(values, state) or None when the reference refuses the range (simple_provider.rs:3689-3698)."""
    s = C.c_uint64(state)
    out = np.empty(n, dtype=np.float64)
    l = lib()
    l.orc_rng_integer_range.restype = C.c_int
    l.orc_rng_integer_range.argtypes = [C.POINTER(C.c_uint64), C.c_longlong, C.c_longlong, C.c_size_t, _DP]
    if l.orc_rng_integer_range(C.byref(s), lower, upper, n, _p(out)):
        return None
    return out, int(s.value)


def lu(a: np.ndarray):
    a = np.asarray(a, dtype=np.float64)
    rows, cols = a.shape
    fa = _f(a)
    comb = np.empty(rows * cols)
    low = np.empty(rows * rows)
    up = np.empty(rows * cols)
    pm = np.empty(rows * rows)
    pv = np.empty(rows)
    if lib().orc_lu(_p(fa), rows, cols, _p(comb), _p(low), _p(up), _p(pm), _p(pv)):
        raise MemoryError
    return (comb.reshape((rows, cols), order="F"), low.reshape((rows, rows), order="F"),
            up.reshape((rows, cols), order="F"), pm.reshape((rows, rows), order="F"), pv.reshape((rows, 1)))


def mldivide_svd(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
        b = b.reshape(-1, 1)
    m, n = a.shape
    fa, fb = _f(a), _f(b)
    scalar = a.size == 1
    xr = b.shape[0] if scalar else n
    out = np.empty(xr * b.shape[1])
    if lib().orc_mldivide_svd(_p(fa), m, n, _p(fb), b.shape[0], b.shape[1], _p(out)):
        raise ValueError("mldivide: row mismatch")
    return out.reshape((xr, b.shape[1]), order="F")


def mldivide_lu(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
        b = b.reshape(-1, 1)
    n = a.shape[0]
    fa, fb = _f(a), _f(b)
    out = np.empty(n * b.shape[1])
    rc = lib().orc_mldivide_lu(_p(fa), n, _p(fb), b.shape[1], _p(out))
    if rc == 3:
        raise np.linalg.LinAlgError("singular")
    return out.reshape((n, b.shape[1]), order="F")


def mrdivide(b: np.ndarray, a: np.ndarray) -> np.ndarray:
    """X = B / A as the CPU builtin computes it (crates/runmat-runtime/src/builtins/math/linalg/ops/mrdivide.rs:317-341,
    379-388): scalar A divides by multiplying with its reciprocal; otherwise the SVD solve of A' X' = B', transposed back."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 1:
        return b * (1.0 / float(a.reshape(-1)[0]))
    if b.shape[1] != a.shape[1]:
        raise ValueError("mrdivide: column mismatch")
    return mldivide_svd(np.ascontiguousarray(a.T), np.ascontiguousarray(b.T)).T


def stochastic_evolution(state: int, data, drift: float, scale: float, steps: int):
    """-> (evolved array, new rng state)"""
    a = np.asarray(data, dtype=np.float64)
    flat = _f(a).copy()
    st = C.c_uint64(state)
    rc = lib().orc_stochastic_evolution(C.byref(st), _p(flat), flat.size, drift, scale, steps)
    if rc:
        raise MemoryError("orc_stochastic_evolution")
    return flat.reshape(a.shape, order="F"), st.value


def image_normalize(x, epsilon, gain=None, bias=None, gamma=None, clamp_zero=True) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    b, h, w = x.shape
    out = np.empty(x.size)
    lib().orc_image_normalize(_p(_f(x)), b, h, w, epsilon, int(gain is not None), float(gain or 0.0), int(bias is not None),
                              float(bias or 0.0), int(bool(clamp_zero)), int(gamma is not None), float(gamma or 0.0), _p(out))
    return out.reshape(x.shape, order="F")


def matmul_power_step(a, b, epsilon=0.0) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    out = np.empty(a.shape[0] * b.shape[1])
    rc = lib().orc_matmul_power_step(_p(_f(a)), a.shape[0], a.shape[1], _p(_f(b)), b.shape[0], b.shape[1], epsilon, _p(out))
    if rc:
        raise ValueError("matmul_power_step: inner dims must agree")
    return out.reshape((a.shape[0], b.shape[1]), order="F")


def covariance(x: np.ndarray, biased: bool = False) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    out = np.empty(x.shape[1] * x.shape[1])
    lib().orc_covariance(_p(_f(x)), x.shape[0], x.shape[1], int(bool(biased)), _p(out))
    return out.reshape((x.shape[1], x.shape[1]), order="F")


def syrk(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.shape[1] * a.shape[1])
    lib().orc_syrk(_p(_f(a)), a.shape[0], a.shape[1], _p(out))
    return out.reshape((a.shape[1], a.shape[1]), order="F")


def transpose(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.size)
    lib().orc_transpose(_p(_f(a)), a.shape[0], a.shape[1], _p(out))
    return out.reshape((a.shape[1], a.shape[0]), order="F")


def linsolve(a: np.ndarray, b: np.ndarray, lower=False, upper=False, transposed=False):
    """solve_real, linsolve.rs:691-726 -> (solution, rcond). General systems go through the SVD restatement
    (rcond = sigma_min / sigma_max is not restated: returned as None)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
        b = b.reshape(-1, 1)
    if transposed:
        a = transpose(a)
        if lower or upper:
            lower, upper = upper, lower
    if a.shape[0] != b.shape[0]:
        raise ValueError("Matrix dimensions must agree.")
    if not (lower or upper):
        return mldivide_svd(a, b), None
    n = a.shape[0]
    out = np.empty(n * b.shape[1])
    rcond = C.c_double(0.0)
    rc = lib().orc_linsolve_tri(1 if lower else 0, _p(_f(a)), n, _p(_f(b)), b.shape[1], _p(out), C.byref(rcond))
    if rc == 3:
        raise np.linalg.LinAlgError("linsolve: matrix is singular to working precision.")
    return out.reshape((n, b.shape[1]), order="F"), rcond.value


def sin_mul_add(a, b, c) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    fa, fb, fc = _f(a), _f(b), _f(c)
    out = np.empty_like(fa)
    lib().orc_sin_mul_add(_p(fa), _p(fb), _p(fc), fa.size, _p(out))
    return out.reshape(a.shape, order="F")


def elementwise_math_chain(x) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    fx = _f(x)
    out = np.empty_like(fx)
    lib().orc_elementwise_math_chain(_p(fx), fx.size, _p(out))
    return out.reshape(x.shape, order="F")


def monte_carlo_price(state: int, M: int, T: int, S0=100.0, mu=0.05, sigma=0.2, dt=1.0 / 252.0, K=100.0):
    s = C.c_uint64(state)
    price = lib().orc_monte_carlo_price(C.byref(s), M, T, S0, mu, sigma, dt, K)
    return float(price), int(s.value)


def fill_uniform(seed: int, lo: float, hi: float, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float64)
    lib().orc_fill_uniform(seed, lo, hi, n, _p(out))
    return out


# ---- reductions next to sum / mean (round 3): [pre, red, post] view around a zero-based `dim`; dim None = all elements ----
def _dim_view(x: np.ndarray, dim):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim < 2:
        x = x.reshape(-1, 1)
    if dim is None:
        return x, 1, x.size, 1, (1, 1)
    shape = x.shape
    pre = int(np.prod(shape[:dim], dtype=np.int64))
    post = int(np.prod(shape[dim + 1:], dtype=np.int64))
    oshape = tuple(1 if d == dim else e for d, e in enumerate(shape))
    return x, pre, shape[dim], post, oshape


def _bind(name, argtypes):
    f = getattr(lib(), name)
    f.restype = C.c_int
    f.argtypes = argtypes
    return f


def minmax_dim(x, dim: int, is_max: bool, omitnan: bool = False):
    """(values, indices) of min / max along `dim` with the CPU builtin's rules (min.rs:1437-1531, max.rs:1715-1727)."""
    x, pre, red, post, oshape = _dim_view(x, dim)
    f = _bind("orc_minmax_dim", [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _DP, _DP])
    xf = _f(x)
    v, i = np.empty(pre * post), np.empty(pre * post)
    f(_p(xf), pre, red, post, int(is_max), int(omitnan), _p(v), _p(i))
    return v.reshape(oshape, order="F"), i.reshape(oshape, order="F")


def std_dim(x, dim, population: bool = False, omitnan: bool = False) -> np.ndarray:
    x, pre, red, post, oshape = _dim_view(x, dim)
    f = _bind("orc_std_dim", [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _DP])
    xf = _f(x)
    out = np.empty(pre * post)
    f(_p(xf), pre, red, post, int(population), int(omitnan), _p(out))
    return out.reshape(oshape, order="F")


def truth_dim(x, dim, op: str, omitnan: bool = False) -> np.ndarray:
    """op: "nnz" | "any" | "all" (nnz.rs:358, any.rs:722-733, all.rs:671-703)."""
    x, pre, red, post, oshape = _dim_view(x, dim)
    f = _bind("orc_truth_dim", [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _DP])
    xf = _f(x)
    out = np.empty(pre * post)
    f(_p(xf), pre, red, post, {"nnz": 0, "any": 1, "all": 2}[op], int(omitnan), _p(out))
    return out.reshape(oshape, order="F")


def cumulative(x, dim: int, prod: bool = False, reverse: bool = False, omitnan: bool = False) -> np.ndarray:
    x, pre, red, post, _ = _dim_view(x, dim)
    f = _bind("orc_cumulative", [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, _DP])
    xf = _f(x)
    out = np.empty(xf.size)
    f(_p(xf), pre, red, post, int(prod), int(reverse), int(omitnan), _p(out))
    return out.reshape(x.shape, order="F")


# ---- shape / indexing hooks (simple_provider.rs) -------------------------------------------------------------------------
_U32P = C.POINTER(C.c_uint32)


def repmat(x: np.ndarray, reps, shape=None) -> np.ndarray:
    """repmat_numeric (simple_provider.rs:2174-2240); `shape` overrides x.shape (rank-1 / padded shapes numpy cannot carry)."""
    x = np.asarray(x, dtype=np.float64)
    shape = tuple(x.shape if shape is None else shape)
    fx = _f(x) if shape == x.shape else np.ascontiguousarray(x.reshape(-1, order="F"))
    reps = [int(r) for r in reps]
    sh = (C.c_size_t * max(len(shape), 1))(*shape)
    rp = (C.c_size_t * max(len(reps), 1))(*reps)
    oshape = (C.c_size_t * 32)()
    orank = C.c_size_t()
    l = lib()
    l.orc_repmat.restype = C.c_int
    l.orc_repmat.argtypes = [_DP, _SZP, C.c_size_t, _SZP, C.c_size_t, _DP, _SZP, _SZP]
    if l.orc_repmat(_p(fx), sh, len(shape), rp, len(reps), None, oshape, C.byref(orank)):
        raise ValueError("repmat: replication factors must be specified")
    out_shape = tuple(int(oshape[i]) for i in range(orank.value))
    out = np.empty(int(np.prod(out_shape, dtype=np.int64)), dtype=np.float64)
    l.orc_repmat(_p(fx), sh, len(shape), rp, len(reps), _p(out), oshape, C.byref(orank))
    return out.reshape(out_shape, order="F")


def permute(x: np.ndarray, order) -> np.ndarray:
    """permute_data (simple_provider.rs:1645-1740); zero-based order."""
    x = np.asarray(x, dtype=np.float64)
    fx = _f(x)
    order = [int(o) for o in order]
    sh = (C.c_size_t * max(x.ndim, 1))(*x.shape)
    od = (C.c_size_t * max(len(order), 1))(*order)
    oshape = (C.c_size_t * 32)()
    n = max(len(order), x.ndim)
    out = np.empty(fx.size, dtype=np.float64)
    l = lib()
    l.orc_permute.restype = C.c_int
    l.orc_permute.argtypes = [_DP, _SZP, C.c_size_t, _SZP, C.c_size_t, _DP, _SZP]
    rc = l.orc_permute(_p(fx), sh, x.ndim, od, len(order), _p(out), oshape)
    if rc:
        raise ValueError({1: "permute: order must not be empty", 2: "permute: order length must be at least the number of dimensions",
                          3: "permute: invalid dimension index", 4: "permute: duplicate dimension index"}.get(rc, "permute"))
    return out.reshape(tuple(int(oshape[i]) for i in range(len(order))), order="F")


def gather_linear(x: np.ndarray, indices, out_shape) -> np.ndarray:
    fx = _f(np.asarray(x, dtype=np.float64))
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    out = np.empty(idx.size, dtype=np.float64)
    l = lib()
    l.orc_gather_linear.restype = C.c_size_t
    l.orc_gather_linear.argtypes = [_DP, C.c_size_t, _U32P, C.c_size_t, _DP]
    bad = l.orc_gather_linear(_p(fx), fx.size, idx.ctypes.data_as(_U32P), idx.size, _p(out))
    if bad:
        raise IndexError(f"gather_linear: index at position {bad - 1} out of bounds")
    return out.reshape(tuple(out_shape), order="F")


def scatter_linear(target: np.ndarray, indices, values) -> np.ndarray:
    t = np.asarray(target, dtype=np.float64)
    ft = _f(t).copy()
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    fv = _f(np.asarray(values, dtype=np.float64))
    l = lib()
    l.orc_scatter_linear.restype = C.c_size_t
    l.orc_scatter_linear.argtypes = [_DP, C.c_size_t, _U32P, C.c_size_t, _DP]
    bad = l.orc_scatter_linear(_p(ft), ft.size, idx.ctypes.data_as(_U32P), idx.size, _p(fv))
    if bad:
        raise IndexError(f"scatter_linear: index at position {bad - 1} out of bounds")
    return ft.reshape(t.shape, order="F")


def linspace(start: float, stop: float, count: int) -> np.ndarray:
    out = np.empty(int(count), dtype=np.float64)
    l = lib()
    l.orc_linspace.restype = None
    l.orc_linspace.argtypes = [C.c_double, C.c_double, C.c_size_t, _DP]
    l.orc_linspace(float(start), float(stop), int(count), _p(out))
    return out.reshape((1, int(count)))


def eye(shape) -> np.ndarray:
    shape = tuple(int(x) for x in shape)
    shape = (1, 1) if len(shape) == 0 else (shape[0], shape[0]) if len(shape) == 1 else shape  # normalize_shape, simple_provider.rs:779-788
    out = np.empty(int(np.prod(shape, dtype=np.int64)), dtype=np.float64)
    l = lib()
    l.orc_eye.restype = None
    l.orc_eye.argtypes = [_SZP, C.c_size_t, _DP]
    l.orc_eye(_shape(shape), len(shape), _p(out))
    return out.reshape(shape, order="F")


def flip(x: np.ndarray, axes) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape)
    for a in axes:
        while len(shape) <= a:
            shape.append(1)
    flags = np.zeros(len(shape), dtype=np.int32)
    for a in axes:
        flags[a] ^= 1
    fx = _f(x)
    out = np.empty_like(fx)
    l = lib()
    l.orc_flip.restype = None
    l.orc_flip.argtypes = [_DP, _SZP, C.c_size_t, C.POINTER(C.c_int), _DP]
    l.orc_flip(_p(fx), _shape(shape), len(shape), flags.ctypes.data_as(C.POINTER(C.c_int)), _p(out))
    return out.reshape(x.shape, order="F")


def circshift(x: np.ndarray, shifts) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape) + [1] * max(0, len(shifts) - x.ndim)
    full = np.zeros(len(shape), dtype=np.int64)
    full[:len(shifts)] = shifts
    fx = _f(x)
    out = np.empty_like(fx)
    l = lib()
    l.orc_circshift.restype = None
    l.orc_circshift.argtypes = [_DP, _SZP, C.c_size_t, C.POINTER(C.c_longlong), _DP]
    l.orc_circshift(_p(fx), _shape(shape), len(shape), full.ctypes.data_as(C.POINTER(C.c_longlong)), _p(out))
    return out.reshape(x.shape, order="F")


def tri(x: np.ndarray, upper: bool, offset: int) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    fx = _f(x)
    out = np.empty_like(fx)
    l = lib()
    l.orc_tri.restype = None
    l.orc_tri.argtypes = [_DP, _SZP, C.c_size_t, C.c_int, C.c_longlong, _DP]
    l.orc_tri(_p(fx), _shape(x.shape), x.ndim, 1 if upper else 0, int(offset), _p(out))
    return out.reshape(x.shape, order="F")



def _pre_len_post(shape, dim):
    shape = list(shape)
    pre = int(np.prod(shape[:dim], dtype=np.int64)) if dim > 0 else 1
    post = int(np.prod(shape[dim + 1:], dtype=np.int64)) if dim + 1 < len(shape) else 1
    return pre, shape[dim], post


def cumextreme(x: np.ndarray, dim: int, is_max: bool, reverse: bool = False, omit_nan: bool = False):
    """cummin / cummax along zero-based `dim` (< ndim): (values, 1-based indices)  (cummin.rs:719-876)."""
    x = np.asarray(x, dtype=np.float64)
    pre, ln, post = _pre_len_post(x.shape, dim)
    fx = _f(x)
    vals, idx = np.empty_like(fx), np.empty_like(fx)
    l = lib()
    l.orc_cumextreme.restype = None
    l.orc_cumextreme.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, _DP, _DP]
    l.orc_cumextreme(_p(fx), pre, ln, post, int(is_max), int(reverse), int(omit_nan), _p(vals), _p(idx))
    return vals.reshape(x.shape, order="F"), idx.reshape(x.shape, order="F")


def diff(x: np.ndarray, order: int, dim: int, column_major: bool = False) -> np.ndarray:
    """diff_tensor_host with an explicit zero-based dim (diff.rs:439-507): `order` first differences, each in the reference's output
    order (k fastest; see orc_diff_once) unless column_major.  The result is returned as a flat buffer plus its shape when the layout is
    the reference's (it is not a column-major array for dim >= 1 with leading extents > 1), as an array otherwise."""
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape)
    while len(shape) <= dim:
        shape.append(1)
    cur = _f(x)
    l = lib()
    l.orc_diff_once.restype = None
    l.orc_diff_once.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _DP]
    for _ in range(order):
        pre, ln, post = _pre_len_post(shape, dim)
        oshape = list(shape)
        oshape[dim] = max(ln, 1) - 1
        if ln <= 1 or cur.size == 0:
            cur, shape = np.zeros(0), oshape
            break
        out = np.empty(pre * (ln - 1) * post)
        l.orc_diff_once(_p(cur), pre, ln, post, int(column_major), _p(out))
        cur, shape = out, oshape
    return cur, shape


def sort_dim(x: np.ndarray, dim: int, descend: bool = False, by_abs: bool = False):
    """Stable sort of every line along zero-based dim: (sorted, 1-based original positions)  (sort.rs:413-468, 538-574)."""
    x = np.asarray(x, dtype=np.float64)
    pre, ln, post = _pre_len_post(x.shape, dim)
    fx = _f(x)
    vals, idx = np.empty_like(fx), np.empty_like(fx)
    l = lib()
    l.orc_sort_dim.restype = None
    l.orc_sort_dim.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _DP, _DP]
    l.orc_sort_dim(_p(fx), pre, ln, post, int(descend), int(by_abs), _p(vals), _p(idx))
    return vals.reshape(x.shape, order="F"), idx.reshape(x.shape, order="F")


def median_dim(x: np.ndarray, dim: int) -> np.ndarray:
    """Include-NaN median along zero-based dim (median.rs:644-741)."""
    x = np.asarray(x, dtype=np.float64)
    pre, ln, post = _pre_len_post(x.shape, dim)
    fx = _f(x)
    oshape = list(x.shape)
    oshape[dim] = 1
    out = np.empty(pre * post)
    l = lib()
    l.orc_median_dim.restype = None
    l.orc_median_dim.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, _DP]
    l.orc_median_dim(_p(fx), pre, ln, post, _p(out))
    return out.reshape(oshape, order="F")


def median_all(x: np.ndarray, successive: bool = False) -> float:
    """median(x, 'all'): the provider hook (reduce_median: simple_provider.rs:7167-7193) sorts ALL elements; the host path
    (median.rs:531-541) takes medians dimension after dimension - the two differ on general inputs."""
    x = np.asarray(x, dtype=np.float64)
    if not successive:
        return float(median_dim(x.reshape(-1, 1, order="F"), 0)[0, 0]) if x.size else float("nan")
    cur = x
    for d in range(x.ndim):
        cur = median_dim(cur, d)
    return float(cur.reshape(-1)[0])


def diag_from_vector(v: np.ndarray, offset: int, rows: int = None, cols: int = None) -> np.ndarray:
    """diag_from_vector(_sized) (simple_provider.rs:3222-3281): square of size len + |offset| unless rows / cols are given."""
    v = np.asarray(v, dtype=np.float64).ravel(order="F")
    if rows is None:
        rows = cols = v.size + abs(int(offset))
    out = np.empty(rows * cols)
    l = lib()
    l.orc_diag_from_vector.restype = None
    l.orc_diag_from_vector.argtypes = [_DP, C.c_size_t, C.c_longlong, C.c_size_t, C.c_size_t, _DP]
    l.orc_diag_from_vector(_p(np.ascontiguousarray(v)), v.size, int(offset), rows, cols, _p(out))
    return out.reshape((rows, cols), order="F")


def kron(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    rank = max(a.ndim, b.ndim, 1)
    sa, sb = list(a.shape) + [1] * (rank - a.ndim), list(b.shape) + [1] * (rank - b.ndim)
    so = [x * y for x, y in zip(sa, sb)]
    out = np.empty(int(np.prod(so, dtype=np.int64)))
    l = lib()
    l.orc_kron.restype = None
    l.orc_kron.argtypes = [_DP, _SZP, _DP, _SZP, C.c_size_t, _DP]
    fa, fb = _f(a), _f(b)
    if out.size:
        l.orc_kron(_p(fa), _shape(sa), _p(fb), _shape(sb), rank, _p(out))
    return out.reshape(so, order="F")


def cross(a: np.ndarray, b: np.ndarray, dim_one_based: int = None) -> np.ndarray:
    """cross.rs:332-364, 443-467: along the given 1-based dimension (length 3) or the first of length 3."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    if dim_one_based is None:
        dim_one_based = list(a.shape).index(3) + 1
    assert a.shape[dim_one_based - 1] == 3
    pre, _, post = _pre_len_post(a.shape, dim_one_based - 1)
    out = np.empty(a.size)
    l = lib()
    l.orc_cross.restype = None
    l.orc_cross.argtypes = [_DP, _DP, C.c_size_t, C.c_size_t, _DP]
    l.orc_cross(_p(_f(a)), _p(_f(b)), pre, post, _p(out))
    return out.reshape(a.shape, order="F")


def gradient(x: np.ndarray, dim: int, spacing: float = 1.0, coords=None) -> np.ndarray:
    """gradient.rs:650-720 along zero-based dim (beyond the rank: extent one -> zeros)."""
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape) + [1] * max(0, dim + 1 - x.ndim)
    pre, ln, post = _pre_len_post(shape, dim)
    out = np.empty(x.size)
    l = lib()
    l.orc_gradient.restype = None
    l.orc_gradient.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, _DP, _DP]
    cp = None
    if coords is not None:
        cc = np.ascontiguousarray(np.asarray(coords, dtype=np.float64).ravel())
        cp = _p(cc)
    l.orc_gradient(_p(_f(x)), pre, ln, post, float(spacing), cp, _p(out))
    return out.reshape(x.shape, order="F")


def issymmetric(a: np.ndarray, skew: bool = False, tol: float = 0.0) -> bool:
    a = np.asarray(a, dtype=np.float64)
    rows, cols = (a.shape[0], a.shape[1]) if a.ndim >= 2 else (a.size, 1)
    l = lib()
    l.orc_issymmetric.restype = C.c_int
    l.orc_issymmetric.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_int, C.c_double]
    return bool(l.orc_issymmetric(_p(_f(a)), rows, cols, int(skew), float(tol)))


def sort_rows(a, columns, comparison: str = "auto"):
    """sort_rows_host, runmat-accelerate/src/sortrows_host.rs:11-140: a stable sort of the row indices under `compare_rows`; columns as
    (zero-based index, "ascend" | "descend") pairs -> (sorted matrix, 1-based source rows [rows, 1])."""
    import functools
    a = np.asarray(a, dtype=np.float64)
    if a.ndim >= 2:
        m = a.reshape(a.shape[0], a.shape[1], order="F")
    else:
        m = a.reshape(max(a.size, 1), 1)
    rows, cols = m.shape

    def scalar(x, y, descend):
        if x != x or y != y:
            if x != x and y != y:
                return 0
            gt = 1 if x != x else -1  # NaN is the greater one ascending ...
            return -gt if descend else gt  # ... and comes first descending
        if comparison == "abs" and abs(x) != abs(y):
            c = -1 if abs(x) < abs(y) else 1
            return -c if descend else c
        c = int(x > y) - int(x < y)
        return -c if descend else c

    def cmp(r, s):
        for idx, order in columns:
            if idx >= cols:
                continue
            c = scalar(m[r, idx], m[s, idx], order == "descend")
            if c:
                return c
        return 0

    order = list(range(rows))
    if rows > 1 and cols > 0 and columns:
        order.sort(key=functools.cmp_to_key(cmp))
    return m[order, :].reshape(a.shape, order="F") if a.size else a.copy(), (np.array(order, dtype=np.float64) + 1.0).reshape(-1, 1)


def _canon_key(v: float):
    """canonicalize_f64, unique.rs:1347-1355 / ismember.rs:814-822: every NaN one key, both zeros one key, otherwise the bits."""
    if v != v:
        return "nan"
    if v == 0.0:
        return 0.0
    return np.float64(v).tobytes()


def unique(x, order: str = "sorted", occurrence: str = "first"):
    """unique_numeric_elements, unique.rs:473-556 -> (values [g, 1], ia [g, 1], ic [n, 1]) with 1-based indices."""
    data = np.asarray(x, dtype=np.float64).ravel(order="F")
    entries, index, entry_of = [], {}, []
    for i, v in enumerate(data):
        k = _canon_key(v)
        if k in index:
            entries[index[k]][2] = i
        else:
            index[k] = len(entries)
            entries.append([v, i, i])
        entry_of.append(index[k])
    ordering = list(range(len(entries)))
    if order == "sorted":  # compare_f64 (unique.rs:1357-1369): NaN last; a stable sort over distinct keys
        ordering.sort(key=lambda e: (entries[e][0] != entries[e][0], 0.0 if entries[e][0] != entries[e][0] else entries[e][0]))
    position = {e: p for p, e in enumerate(ordering)}
    values = np.array([entries[e][0] for e in ordering], dtype=np.float64).reshape(-1, 1)
    ia = np.array([entries[e][1 if occurrence == "first" else 2] + 1 for e in ordering], dtype=np.float64).reshape(-1, 1)
    ic = np.array([position[e] + 1 for e in entry_of], dtype=np.float64).reshape(-1, 1)
    return values, ia, ic


def union(a, b, order: str = "sorted"):
    """union_numeric_elements + assemble_numeric_union, union.rs:491-544, 1238-1279 -> (values, ia, ib), each [g, 1]."""
    da, db = (np.asarray(v, dtype=np.float64).ravel(order="F") for v in (a, b))
    values, first, _ = unique(np.concatenate([da, db]), order, "first")
    first = first.ravel()
    ia = first[first <= da.size]
    ib = first[first > da.size] - da.size
    return values, ia.reshape(-1, 1), ib.reshape(-1, 1)


def setdiff(a, b, order: str = "sorted"):
    """setdiff_numeric_elements, setdiff.rs:463-496 -> (values, ia)."""
    values, ia, _ = unique(a, order, "first")
    mask, _ = ismember(values.ravel(), np.asarray(b, dtype=np.float64).ravel(order="F")) if values.size else (np.zeros(0, dtype=np.uint8), None)
    keep = mask == 0
    return values.ravel()[keep].reshape(-1, 1), ia.ravel()[keep].reshape(-1, 1)


def ismember(a, b):
    """ismember_numeric_elements, ismember.rs:413-438 -> (mask uint8, loc) in a's shape."""
    a = np.asarray(a, dtype=np.float64)
    first = {}
    for i, v in enumerate(np.asarray(b, dtype=np.float64).ravel(order="F")):
        first.setdefault(_canon_key(v), i + 1)
    flat = a.ravel(order="F")
    loc = np.array([first.get(_canon_key(v), 0) for v in flat], dtype=np.float64)
    return (loc > 0).astype(np.uint8).reshape(a.shape, order="F"), loc.reshape(a.shape, order="F")


def covariance_to_correlation(m):
    """provider_validate_covariance_matrix + provider_covariance_to_correlation, simple_provider.rs:896-975 -> (correlation, sigma [n, 1]); raises
    ValueError with the CPU's message when the matrix is refused."""
    m = np.asarray(m, dtype=np.float64)
    n = m.shape[0]
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise ValueError("covariance matrix must be square")
    if np.any(~np.isnan(m) & ~np.isfinite(m)):
        raise ValueError("covariance matrix must contain finite values or NaN")
    if any(not (m[i, i] >= 0.0) for i in range(n)):
        raise ValueError("covariance matrix diagonal entries must be nonnegative")
    for col in range(n):
        for row in range(col):
            a, b = m[row, col], m[col, row]
            if np.isnan(a) and np.isnan(b):
                continue
            if np.isnan(a) or np.isnan(b) or not abs(a - b) <= 1e-10 * max(abs(a), abs(b), 1.0):
                raise ValueError("covariance matrix must be symmetric")
            vr, vc = m[row, row], m[col, col]
            if np.isnan(vr) or np.isnan(vc):
                continue
            mc = np.sqrt(vr * vc)
            if not abs(a) <= mc + 1e-10 * max(mc, abs(a), 1.0):
                raise ValueError("covariance magnitude exceeds variance bounds")
    sigma = np.sqrt(np.diag(m)) if n else np.zeros(0)
    den = np.outer(sigma, sigma)
    with np.errstate(invalid="ignore", divide="ignore"):
        corr = np.where(den == 0.0, np.nan, m / np.where(den == 0.0, 1.0, den))
    return corr, sigma.reshape(-1, 1)


def _eps_like(v: float) -> float:
    """common/linalg.rs:218-228: the gap from |v| to the next double."""
    a = abs(float(v))
    return float(np.nextafter(a, np.inf) - a)


def svd_default_tolerance(sv, rows: int, cols: int) -> float:
    """common/linalg.rs:209-215."""
    return max(rows, cols) * _eps_like(max([abs(v) for v in sv], default=0.0))


def rank(a, tolerance=None) -> int:
    """rank_real_tensor_impl, rank.rs:280-295: singular values (nalgebra's SVD on the CPU - LAPACK's here, same values to rounding) above
    the tolerance."""
    a = np.asarray(a, dtype=np.float64)
    a = a.reshape(a.shape[0], a.shape[1]) if a.ndim >= 2 else a.reshape(-1, 1)
    if a.shape[0] == 0 or a.shape[1] == 0:
        return 0
    sv = np.linalg.svd(a, compute_uv=False)
    cut = svd_default_tolerance(sv, *a.shape) if tolerance is None else tolerance
    return int(sum(1 for v in sv if np.isinf(v) or v > cut))


def cond2(a) -> float:
    """cond_real_tensor / singular_value_cond, cond.rs:276-330, 448-467 (2-norm)."""
    a = np.asarray(a, dtype=np.float64)
    a = a.reshape(a.shape[0], a.shape[1]) if a.ndim >= 2 else a.reshape(-1, 1)
    if a.shape[0] == 0 or a.shape[1] == 0:
        return 0.0
    if a.size == 1:
        return float("inf") if a.ravel()[0] == 0.0 else 1.0
    sv = np.abs(np.linalg.svd(a, compute_uv=False))
    if not np.isfinite(sv).all():
        return float("inf")
    return float("inf") if sv.min() == 0.0 else float(sv.max() / sv.min())


def pinv(a, tolerance=None) -> np.ndarray:
    """pseudoinverse_real, pinv.rs:276-285: V diag(1 / s_i, s_i > cutoff) U'."""
    a = np.asarray(a, dtype=np.float64)
    a = a.reshape(a.shape[0], a.shape[1]) if a.ndim >= 2 else a.reshape(-1, 1)
    if a.shape[0] == 0 or a.shape[1] == 0:
        return np.zeros((a.shape[1], a.shape[0]))
    u, sv, vt = np.linalg.svd(a, full_matrices=False)
    cut = svd_default_tolerance(sv, *a.shape) if tolerance is None else tolerance
    inv = np.array([1.0 / v if (np.isinf(v) or v > cut) else 0.0 for v in sv])
    return (vt.T * inv) @ u.T


def peaks_xy(x, y) -> np.ndarray:
    """peaks_at, peaks.rs:546-550 (powi(2/3/5) as the repeated products they compile to), elementwise."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    x2, y2 = x * x, y * y
    a, yp, xp = 1.0 - x, y + 1.0, x + 1.0
    t1 = 3.0 * (a * a) * np.exp(-x2 - yp * yp)
    t2 = 10.0 * (x / 5.0 - x2 * x - (y2 * y2) * y) * np.exp(-x2 - y2)
    t3 = 1.0 / 3.0 * np.exp(-(xp * xp) - y2)
    return t1 - t2 - t3


def peaks(n: int) -> np.ndarray:
    """make_axis + make_grids, peaks.rs:511-541: X(row, col) = axis[col], Y(row, col) = axis[row]."""
    axis = np.array([3.0]) if n == 1 else np.array([-3.0 + 6.0 * i / (n - 1) for i in range(n)], dtype=np.float64).reshape(-1) if n else np.zeros(0)
    return peaks_xy(np.tile(axis.reshape(1, -1), (n, 1)), np.tile(axis.reshape(-1, 1), (1, n)))


def corrcoef(matrix, normalization: str = "unbiased") -> np.ndarray:
    """corrcoef_dense + column_pair_corr + divide_covariance + clamp_correlation, corrcoef.rs:720-926 (rows == All): sequential sums in the
    CPU's order; the result is written on and above the diagonal and mirrored (set_entry, :928-935)."""
    m = np.asarray(matrix, dtype=np.float64)
    m = m.reshape(m.shape[0], -1, order="F") if m.ndim >= 2 else m.reshape(-1, 1)
    rows, cols = m.shape
    out = np.full((cols, cols), np.nan)
    if cols == 0 or rows == 0:
        return out
    denom = float(rows) - 1.0 if normalization == "unbiased" else float(rows)
    if denom <= 0.0:
        return out
    means = np.full(cols, np.nan)
    for c in range(cols):
        fin = m[np.isfinite(m[:, c]), c]
        if fin.size:
            s = 0.0
            for v in fin:
                s += v
            means[c] = s / fin.size
    for c in range(cols):
        if not np.isfinite(means[c]) or not np.isfinite(m[:, c]).all():
            continue
        var = 0.0
        for v in m[:, c]:
            d = v - means[c]
            var += d * d
        var /= denom
        out[c, c] = 1.0 if np.sqrt(max(var, 0.0) if -1e-12 < var < 0 else var) > 0.0 else np.nan
        for o in range(c + 1, cols):
            if not np.isfinite(means[o]) or not np.isfinite(m[:, o]).all():
                continue
            vx = vy = cv = 0.0
            for a, b in zip(m[:, c], m[:, o]):
                dx, dy = a - means[c], b - means[o]
                vx += dx * dx
                vy += dy * dy
                cv += dx * dy
            vx, vy, cv = vx / denom, vy / denom, cv / denom
            r = np.nan
            if np.isfinite(vx) and np.isfinite(vy) and vx > 0 and vy > 0:
                r = cv / (np.sqrt(vx) * np.sqrt(vy))
                if r > 1.0 and r - 1.0 < 1e-12:
                    r = 1.0
                elif r < -1.0 and -1.0 - r < 1e-12:
                    r = -1.0
            out[c, o] = out[o, c] = r
    return out


def imfilter(image, kernel, padding="constant", shape: str = "same", mode: str = "correlation") -> np.ndarray:
    """build_imfilter_plan + evaluate_filter + sample_with_padding, builtins/image/filters/imfilter.rs:476-545, 620-783.  padding: a number
    (constant fill) | "replicate" | "symmetric" | "circular".  The sum runs over the kernel's points in storage order (first dimension
    fastest), vectorised over the outputs (numpy applies the same rounded operation to each)."""
    img, ker = np.asarray(image, dtype=np.float64), np.asarray(kernel, dtype=np.float64)
    ishape = list(img.shape) if img.ndim else [1, 1]
    kshape = list(ker.shape) if ker.ndim else [1, 1]
    rank = max(len(ishape), len(kshape))
    iext, kext = ishape + [1] * (rank - len(ishape)), kshape + [1] * (rank - len(kshape))
    origin = [k // 2 for k in kext]
    if shape == "full":
        oext, base = [i + k - 1 for i, k in zip(iext, kext)], [o - (k - 1) for o, k in zip(origin, kext)]
    elif shape == "valid":
        oext, base = [i - k + 1 if i >= k else 0 for i, k in zip(iext, kext)], list(origin)
    else:
        oext, base = list(iext), [0] * rank
    final = list(oext)
    while len(final) > len(ishape) and final[-1] == 1:
        final.pop()
    im = img.reshape(iext, order="F")
    kr = ker.reshape(kext, order="F")
    out = np.zeros(oext)
    if out.size == 0:
        return out.reshape(final, order="F")
    grids = np.meshgrid(*[np.arange(n) for n in oext], indexing="ij")
    const = padding if not isinstance(padding, str) else 0.0

    def resolve(coord, n):
        inside = (coord >= 0) & (coord < n)
        if not isinstance(padding, str) or padding == "constant":
            return np.where(inside, coord, 0), inside
        if padding == "replicate":
            return np.clip(coord, 0, n - 1), np.ones_like(inside)
        if padding == "circular":
            return np.mod(coord, n), np.ones_like(inside)
        if n == 1:
            return np.zeros_like(coord), np.ones_like(inside)
        period = 2 * n - 2
        v = np.mod(coord, period)
        return np.where(v >= n, period - v, v), np.ones_like(inside)

    for kidx in np.ndindex(*kext[::-1]):            # storage order: the first dimension fastest
        kidx = kidx[::-1]
        src = tuple(kext[d] - 1 - kidx[d] for d in range(rank)) if mode == "convolution" else kidx
        kv = kr[src]
        ok = np.ones(oext, dtype=bool)
        coords = []
        for d in range(rank):
            c, inside = resolve(grids[d] + base[d] + kidx[d] - origin[d], iext[d])
            coords.append(c)
            ok &= inside
        sample = np.where(ok, im[tuple(coords)], const)
        out = out + kv * sample
    return out.reshape(final, order="F")


def interp1(x, y, xq, method: str = "linear", extrapolation="nan") -> np.ndarray:
    """interp1_value / interp1_interval_index, simple_provider.rs:1396-1472: y as [sample_len, series]; returns [query_len, series].
    extrapolation: "nan" | "extrapolate" | a fill value."""
    import bisect
    x = np.asarray(x, dtype=np.float64).ravel()
    y = np.asarray(y, dtype=np.float64).reshape(x.size, -1, order="F")
    q = np.asarray(xq, dtype=np.float64).ravel(order="F")
    extrap = extrapolation == "extrapolate"
    oor = np.nan if extrapolation in ("nan", "extrapolate") else float(extrapolation)
    xs = x.tolist()
    last = x.size - 1
    out = np.empty((q.size, y.shape[1]))
    for s in range(y.shape[1]):
        ys = y[:, s]
        for i, v in enumerate(q):
            if not np.isfinite(v):
                out[i, s] = np.nan
                continue
            if method == "linear":
                if v < x[0]:
                    piece = 0 if extrap else None
                elif v > x[last]:
                    piece = last - 1 if extrap else None
                elif v == x[last]:
                    piece = last - 1
                else:
                    idx = bisect.bisect_left(xs, v)
                    piece = min(idx, last - 1) if idx < x.size and x[idx] == v else (idx - 1 if 0 < idx < x.size else None)
                if piece is None:
                    out[i, s] = oor
                else:
                    h = x[piece + 1] - x[piece]
                    t = (v - x[piece]) / h
                    out[i, s] = ys[piece] + t * (ys[piece + 1] - ys[piece])
            elif v < x[0]:
                out[i, s] = ys[0] if extrap else oor
            elif v > x[last]:
                out[i, s] = ys[last] if extrap else oor
            else:
                idx = bisect.bisect_left(xs, v)
                if idx < x.size and x[idx] == v:
                    out[i, s] = ys[idx]
                else:
                    left, right = max(idx - 1, 0), min(idx, last)
                    out[i, s] = ys[left] if abs(v - x[left]) <= abs(x[right] - v) else ys[right]
    return out


def iir_filter(b, a, x, dim: int = 0, zi=None):
    """filter_host, filter.rs:1119-1222, on real data: direct form II transposed per channel, all channels advanced together (numpy applies
    the same rounded operation to each).  Coefficients normalised as `Complex /= a0` rounds them: (c * a0 + 0) / (a0 * a0 + 0).
    Returns (y in x's shape, final states with `dim` of extent max(nb, na) - 1)."""
    b, a = (np.asarray(v, dtype=np.float64).ravel(order="F") for v in (b, a))
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape) + [1] * max(0, dim + 1 - x.ndim)
    order = max(b.size, a.size)
    a0 = a[0]
    den = a0 * a0 + 0.0
    bn, an = np.zeros(order), np.zeros(order)
    bn[:b.size] = (b * a0 + 0.0) / den
    an[0] = 1.0
    an[1:a.size] = (a[1:] * a0 + 0.0) / den
    xs = np.moveaxis(x.reshape(shape), dim, 0)
    st = np.zeros((max(order - 1, 0),) + xs.shape[1:])
    if zi is not None and order > 1:
        zshape = list(shape)
        zshape[dim] = order - 1
        st = np.moveaxis(np.asarray(zi, dtype=np.float64).reshape(zshape), dim, 0).copy()
    y = np.empty_like(xs)
    if order == 1:
        y = bn[0] * xs
    else:
        for n in range(xs.shape[0]):
            xn = xs[n]
            yv = bn[0] * xn + st[0]
            y[n] = yv
            for i in range(1, order):
                nxt = st[i] if i < order - 1 else 0.0
                st[i - 1] = (bn[i] * xn + nxt) - an[i] * yv
    return np.moveaxis(y, 0, dim).reshape(x.shape), np.moveaxis(st, 0, dim)


def _poly_orientation(shape) -> str:
    """poly_orientation_from_shape, runmat-accelerate/src/simple_provider.rs:320-338."""
    kinds = ["column" if d == 0 else "row" for d, n in enumerate(shape) if n > 1]
    if len(kinds) > 1:
        raise ValueError("polyder: coefficient inputs must be vectors")
    return kinds[0] if kinds else "scalar"


def _poly_shaped(values, orientation):
    """allocate_polynomial / poly_shape_for_len, simple_provider.rs:341-348, 763-770 -> (values, shape)."""
    v = np.array(values, dtype=np.float64)
    return v, ([1, 1] if v.size <= 1 else [v.size, 1] if orientation == "column" else [1, v.size])


def _poly_load(coefficients):
    c = np.asarray(coefficients, dtype=np.float64)
    flat = [float(x) for x in c.ravel(order="F")]
    return (flat if flat else [0.0]), _poly_orientation(c.shape)  # load_polynomial, simple_provider.rs:750-761


def _poly_raw_derivative(c):
    """simple_provider.rs:361-372."""
    if len(c) <= 1:
        return [0.0]
    return [float(np.float64(c[i]) * np.float64(len(c) - 1 - i)) for i in range(len(c) - 1)]


def _poly_trim(c):
    """poly_trim_slice, simple_provider.rs:350-359 (POLYDER_EPS = 1e-12, :72; a NaN is not above it)."""
    for i, v in enumerate(c):
        if abs(v) > 1.0e-12:
            return list(c[i:])
    return [0.0]


def _poly_convolve(a, b):
    """poly_convolve_real, simple_provider.rs:544-555: result[i + j] += a[i] * b[j], i outer."""
    r = [np.float64(0.0)] * (len(a) + len(b) - 1)
    with np.errstate(all="ignore"):
        for i, ai in enumerate(a):
            for j, bj in enumerate(b):
                r[i + j] = r[i + j] + np.float64(ai) * np.float64(bj)
    return [float(x) for x in r]


def _poly_combine(a, b, sign):
    """poly_add_real / poly_sub_real, simple_provider.rs:557-579: both right-aligned and added to (subtracted from) zeros."""
    n = max(len(a), len(b))
    r = [np.float64(0.0)] * n
    with np.errstate(all="ignore"):
        for i, v in enumerate(a):
            r[n - len(a) + i] = r[n - len(a) + i] + np.float64(v)
        for i, v in enumerate(b):
            r[n - len(b) + i] = r[n - len(b) + i] + np.float64(v) if sign > 0 else r[n - len(b) + i] - np.float64(v)
    return [float(x) for x in r]


def polyder_single(polynomial):
    """simple_provider.rs:3137-3147 -> (values, shape)."""
    c, o = _poly_load(polynomial)
    return _poly_shaped(_poly_trim(_poly_raw_derivative(c)), o)


def polyder_product(p, q):
    """simple_provider.rs:3149-3165: p' q + p q', in p's orientation."""
    pc, o = _poly_load(p)
    qc, _ = _poly_load(q)
    s = _poly_combine(_poly_convolve(_poly_raw_derivative(pc), qc), _poly_convolve(pc, _poly_raw_derivative(qc)), +1)
    return _poly_shaped(_poly_trim(s), o)


def polyder_quotient(u, v):
    """simple_provider.rs:3167-3188 -> ((numerator, shape), (denominator, shape)): u' v - u v' in u's orientation, v v in v's."""
    uc, ou = _poly_load(u)
    vc, ov = _poly_load(v)
    num = _poly_combine(_poly_convolve(_poly_raw_derivative(uc), vc), _poly_convolve(uc, _poly_raw_derivative(vc)), -1)
    return _poly_shaped(_poly_trim(num), ou), _poly_shaped(_poly_trim(_poly_convolve(vc, vc)), ov)


def polyint(polynomial, constant=0.0):
    """poly_integral_real, simple_provider.rs:374-390, 3190-3204 -> (values, shape)."""
    c = np.asarray(polynomial, dtype=np.float64)
    o = _poly_orientation(c.shape)
    flat = c.ravel(order="F")
    with np.errstate(all="ignore"):
        out = [float(flat[i] / np.float64(flat.size - i)) for i in range(flat.size)] + [float(constant)]
    return _poly_shaped(out, o)


def polyval(coefficients, points, mu=None) -> np.ndarray:
    """polyval.rs:886-905 restated on real data: the CPU evaluates acc = acc * x + c in `Complex64` (num-complex: re = a.re * b.re - a.im *
    b.im, rounded product by product), whose real part for real operands is the real recurrence with the product rounded before the sum;
    `mu` first maps x to ((x - mean) * scale) / (scale * scale) - the real part of its complex division by (scale + 0i)."""
    c = np.asarray(coefficients, dtype=np.float64).ravel(order="F")
    x = np.asarray(points, dtype=np.float64)
    v = x.copy()
    if mu is not None:
        mean, scale = float(mu[0]), float(mu[1])
        v = ((v - mean) * scale + 0.0) / (scale * scale + 0.0)
    acc = np.zeros_like(v)
    for cj in c:
        acc = acc * v + cj
    return acc


def meshgrid(axes):
    """ops/constructors.rs:230-308: X(iy, ix[, iz]) = x[ix], Y = y[iy], Z = z[iz]; a one-point Z axis keeps the grids two-dimensional."""
    x, y = (np.asarray(a, dtype=np.float64).ravel() for a in axes[:2])
    z = np.asarray(axes[2], dtype=np.float64).ravel() if len(axes) == 3 else None
    nz = z.size if z is not None else 1
    shape = (y.size, x.size) if nz == 1 else (y.size, x.size, nz)
    iy, ix, iz = np.meshgrid(np.arange(y.size), np.arange(x.size), np.arange(nz), indexing="ij")
    outs = [x[ix].reshape(shape), y[iy].reshape(shape)]
    if z is not None:
        outs.append(z[iz].reshape(shape))
    return outs


_MOVING_OPS = {"sum": 0, "mean": 1, "prod": 2, "min": 3, "max": 4, "median": 5, "std": 6, "var": 7}


def moving_window(x, dim: int, before: int, after: int, op: str, endpoints="shrink", nan_mode: str = "include", normalization: str = "sample") -> np.ndarray:
    """moving.rs:737-825 for count windows; endpoints: "shrink" | "discard" | a fill value."""
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape) + [1] * max(0, dim + 1 - x.ndim)
    pre, ln, post = int(np.prod(shape[:dim], dtype=np.int64)), shape[dim], int(np.prod(shape[dim + 1:], dtype=np.int64))
    ep, fill = (0, 0.0) if endpoints == "shrink" else (1, 0.0) if endpoints == "discard" else (2, float(endpoints))
    oshape = list(shape)
    if ep == 1:
        oshape[dim] = max(0, ln - before - after)
    out = np.zeros(max(1, int(np.prod(oshape, dtype=np.int64))))
    l = lib()
    l.orc_moving_window.restype = None
    l.orc_moving_window.argtypes = [_DP] + [C.c_size_t] * 6 + [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _DP]
    l.orc_moving_window(_p(_f(x.reshape(shape))), pre, ln, post, oshape[dim], before, after, _MOVING_OPS[op], ep, fill, 0 if nan_mode == "include" else 1,
                        0 if normalization == "sample" else 1, _p(out))
    return out[:int(np.prod(oshape, dtype=np.int64))].reshape(oshape, order="F").copy()


_CONV_MODES = {"full": 0, "same": 1, "valid": 2}


def conv1d(a, b, mode: str = "full") -> np.ndarray:
    """conv.rs:481-517 on the flattened operands -> the 1-D result (the caller's orientation only shapes it)."""
    a, b = _f(np.asarray(a, dtype=np.float64).ravel(order="F")), _f(np.asarray(b, dtype=np.float64).ravel(order="F"))
    out = np.zeros(max(1, a.size + b.size))
    l = lib()
    l.orc_conv1d.restype = C.c_size_t
    l.orc_conv1d.argtypes = [_DP, C.c_size_t, _DP, C.c_size_t, C.c_int, _DP]
    n = l.orc_conv1d(_p(a), a.size, _p(b), b.size, _CONV_MODES[mode], _p(out))
    return out[:n].copy()


def conv2d(a, b, mode: str = "full") -> np.ndarray:
    """conv2.rs:595-640 on 2-D operands."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    two = lambda x: x.reshape(1, 1) if x.ndim == 0 else x.reshape(x.shape[0], 1) if x.ndim == 1 else x.reshape(x.shape[0], x.shape[1], order="F")
    a2, b2 = two(a), two(b)
    out = np.zeros(max(1, (a2.shape[0] + b2.shape[0]) * (a2.shape[1] + b2.shape[1])))
    rows, cols = C.c_size_t(), C.c_size_t()
    l = lib()
    l.orc_conv2d.restype = C.c_size_t
    l.orc_conv2d.argtypes = [_DP, C.c_size_t, C.c_size_t, _DP, C.c_size_t, C.c_size_t, C.c_int, _DP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    n = l.orc_conv2d(_p(_f(a2)), a2.shape[0], a2.shape[1], _p(_f(b2)), b2.shape[0], b2.shape[1], _CONV_MODES[mode], _p(out), C.byref(rows), C.byref(cols))
    return out[:n].reshape((rows.value, cols.value), order="F").copy()


def window(kind: str, length: int, periodic: bool = False) -> np.ndarray:
    """simple_provider.rs:95-120 -> [length, 1]."""
    out = np.zeros(max(1, length))
    l = lib()
    l.orc_window.restype = None
    l.orc_window.argtypes = [C.c_int, C.c_size_t, C.c_int, _DP]
    l.orc_window({"hann": 0, "hamming": 1, "blackman": 2}[kind], length, int(periodic), _p(out))
    return out[:length].reshape(length, 1).copy()


def fft_dim(x: np.ndarray, length=None, dim: int = 0, inverse: bool = False) -> np.ndarray:
    """The transform of every line along zero-based `dim`, padded / truncated to `length` points, by direct evaluation of the DFT in
    long double (orc_dft_dim); real or complex input, complex128 output of the reference's shape rule (the shape extended to
    dim + 1 axes, that axis set to the length; ops/fft/fallback.rs:30-52)."""
    x = np.asarray(x)
    cplx = np.iscomplexobj(x)
    x = x.astype(np.complex128 if cplx else np.float64)
    shape = list(x.shape) if x.ndim else [x.size]
    while len(shape) <= dim:
        shape.append(1)
    cur = shape[dim]
    n = cur if length is None else int(length)
    inner = int(np.prod(shape[:dim], dtype=np.int64))
    outer = int(np.prod(shape[dim + 1:], dtype=np.int64))
    oshape = list(shape)
    oshape[dim] = n
    flat = np.ascontiguousarray(x.reshape(-1, order="F"))
    out = np.zeros(inner * n * outer, dtype=np.complex128)
    if out.size and cur:
        l = lib()
        l.orc_dft_dim.restype = None
        l.orc_dft_dim.argtypes = [_DP, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _DP]
        l.orc_dft_dim(flat.view(np.float64).ctypes.data_as(_DP), int(cplx), inner, cur, outer, n, int(inverse), out.view(np.float64).ctypes.data_as(_DP))
    return out.reshape(oshape, order="F")


def hilbert(x, length=None, dim: int = 0) -> np.ndarray:
    """hilbert.rs:349-412 on the oracle's DFT: ifft(fft(x, n, dim) .* mask, n, dim), mask = analytic_signal_multiplier."""
    spec = fft_dim(x, length, dim)
    n = spec.shape[dim]
    mask = np.zeros(n)
    for f in range(n):
        mask[f] = 1.0 if f == 0 else (2.0 if f < n // 2 else (1.0 if f == n // 2 else 0.0)) if n % 2 == 0 else (2.0 if f <= n // 2 else 0.0)
    shape = [1] * spec.ndim
    shape[dim] = n
    return fft_dim(spec * mask.reshape(shape), None, dim, True)


def ishermitian(a: np.ndarray, skew: bool = False, tol: float = 0.0) -> bool:
    """ishermitian.rs:455-482 for real data."""
    a = np.asarray(a, dtype=np.float64)
    rows, cols = (a.shape[0], a.shape[1]) if a.ndim >= 2 else (a.size, 1)
    l = lib()
    l.orc_ishermitian.restype = C.c_int
    l.orc_ishermitian.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_int, C.c_double]
    return bool(l.orc_ishermitian(_p(_f(a)), rows, cols, int(skew), float(tol)))


def bandwidth(a: np.ndarray):
    """(lower, upper), bandwidth.rs:303-318 (a rank-1 shape is a row) and :341-365."""
    a = np.asarray(a, dtype=np.float64)
    rows, cols = (a.shape[0], a.shape[1]) if a.ndim >= 2 else (1, a.size)
    l = lib()
    l.orc_bandwidth.restype = None
    l.orc_bandwidth.argtypes = [_DP, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lo, up = C.c_size_t(), C.c_size_t()
    l.orc_bandwidth(_p(_f(a)), rows, cols, C.byref(lo), C.byref(up))
    return int(lo.value), int(up.value)


def inv(a: np.ndarray):
    """The inverse of a square matrix by LU with partial pivoting (the published algorithm of nalgebra's try_inverse, inv.rs:224-228), or
    None when a pivot is exactly zero."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    out = np.empty(n * n)
    l = lib()
    l.orc_inv.restype = C.c_int
    l.orc_inv.argtypes = [_DP, C.c_size_t, _DP]
    if l.orc_inv(_p(_f(a)), n, _p(out)):
        return None
    return out.reshape((n, n), order="F")


def find(x: np.ndarray, limit=None, last: bool = False):
    """find.rs:593-633 / simple_provider.rs:7500-7575: (linear, rows, cols, values), each [count, 1]; limit None = all (first) or 1 (last)."""
    x = np.asarray(x, dtype=np.float64)
    fx = _f(x)
    cap = fx.size if (limit is None and not last) else (1 if limit is None else int(limit))
    cap = min(cap, fx.size)
    outs = [np.empty(max(cap, 1)) for _ in range(4)]
    l = lib()
    l.orc_find.restype = C.c_size_t
    l.orc_find.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, _DP, _DP, _DP, _DP]
    n = l.orc_find(_p(fx), fx.size, max(x.shape[0] if x.ndim else 1, 1), cap, int(last), *[_p(o) for o in outs])
    return tuple(o[:n].reshape(-1, 1) for o in outs)


def trapz(x: np.ndarray, dim: int, spacing=None, cumulative: bool = False) -> np.ndarray:
    """trapz / cumtrapz along zero-based dim (simple_provider.rs:2421-2598): spacing None (unit), a float, a vector of the dimension's
    extent, or a tensor of x's shape."""
    x = np.asarray(x, dtype=np.float64)
    shape = list(x.shape) + [1] * max(0, dim + 1 - x.ndim)
    pre, ln, post = _pre_len_post(shape, dim)
    kind, scalar, sp = 0, 0.0, None
    if spacing is not None:
        if np.isscalar(spacing):
            kind, scalar = 1, float(spacing)
        else:
            sa = np.asarray(spacing, dtype=np.float64)
            kind = 4 if sa.size == x.size and sa.size != ln else 3
            sp = np.ascontiguousarray(_f(sa))
    oshape = list(shape)
    if not cumulative:
        oshape[dim] = 1
    out = np.empty(int(np.prod(oshape, dtype=np.int64)))
    l = lib()
    l.orc_trapz.restype = None
    l.orc_trapz.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_double, _DP, C.c_int, _DP]
    l.orc_trapz(_p(_f(x)), pre, ln, post, kind, scalar, _p(sp) if sp is not None else None, int(cumulative), _p(out))
    return out.reshape(oshape, order="F")


# ---- subscript / grid / slice-write hooks (numpy restatements of integer / copy work; file:line of what each follows) ----

def ndgrid(axes, output_shape, output_count):
    """simple_provider.rs:2784-2855: out_d[i] = axis_d[(i / stride_d) % extent_d]."""
    total = int(np.prod(output_shape, dtype=np.int64))
    i = np.arange(total, dtype=np.int64)
    outs, stride = [], 1
    for d in range(output_count):
        ext = output_shape[d] if d < len(output_shape) else 1
        ax = np.asarray(axes[d], dtype=np.float64).ravel(order="F")
        assert ax.size == ext
        outs.append(ax[(i // stride) % max(ext, 1)].reshape(output_shape, order="F") if total else np.zeros(output_shape))
        stride *= ext
    return outs


def _round_half_away(v):
    """f64::round: to the nearest integer, halves away from zero - exactly (trunc + comparison of the exact fraction, no v + 0.5)."""
    v = np.asarray(v, dtype=np.float64)
    t = np.trunc(v)
    with np.errstate(invalid="ignore"):
        return np.where(np.abs(v - t) >= 0.5, t + np.copysign(1.0, v), t)


def _coerce_index(v, upper):
    """coerce_sub2ind_value / coerce_linear_index (simple_provider.rs:2268-2291, ind2sub.rs:326-353): None when refused."""
    if not np.isfinite(v):
        return None
    r = _round_half_away(np.float64(v))
    if abs(r - v) > np.finfo(float).eps or r < 1.0 or r > upper:
        return None
    return int(r)


def sub2ind(dims, strides, inputs, scalar_mask, length, output_shape):
    """simple_provider.rs:8340-8420: the linear indices, or (i, d) of the first refused subscript in (element, dimension) order."""
    cols = [np.asarray(x, dtype=np.float64).ravel(order="F") for x in inputs]
    out = np.empty(length)
    for i in range(length):
        off = 0
        for d, (dim, st) in enumerate(zip(dims, strides)):
            c = _coerce_index(cols[d][0 if scalar_mask[d] else i], dim)
            if c is None:
                return (i, d)
            off += (c - 1) * st
        out[i] = off + 1
    return out.reshape(output_shape, order="F")


def ind2sub(dims, strides, indices, total):
    """ind2sub.rs:289-324: one subscript array per dimension, or the position of the first refused index."""
    idx = np.asarray(indices, dtype=np.float64)
    flat = idx.ravel(order="F")
    outs = [np.empty(flat.size) for _ in dims]
    for i, v in enumerate(flat):
        c = _coerce_index(v, total)
        if c is None:
            return i
        for d, (dim, st) in enumerate(zip(dims, strides)):
            outs[d][i] = ((c - 1) // st) % dim + 1
    return [o.reshape(idx.shape, order="F") for o in outs]


def scatter_line(matrix, is_column, index, values):
    """write_slice.rs:680-709: the matrix with one whole column / row replaced."""
    m = np.array(matrix, dtype=np.float64, order="F", copy=True)
    v = np.asarray(values, dtype=np.float64).ravel(order="F")
    if is_column:
        m[:, index] = v
    else:
        m[index, :] = v
    return m


def pow2_scale(m, e):
    """simple_provider.rs:5838-5840: m * exp2(e)."""
    return np.asarray(m, dtype=np.float64) * np.exp2(np.asarray(e, dtype=np.float64))


def powi10(b: int) -> float:
    """Rust f64::powi (compiler-rt __powidf2): square-and-multiply, reciprocal last."""
    recip, e, a, r = b < 0, abs(int(b)), np.float64(10.0), np.float64(1.0)
    with np.errstate(over="ignore"):
        while True:
            if e & 1:
                r = r * a
            e //= 2
            if e == 0:
                break
            a = a * a
        return float(np.float64(1.0) / r) if recip else float(r)


def round_decimals(x, digits: int):
    """round_decimals, simple_provider.rs:5366-5378."""
    x = np.asarray(x, dtype=np.float64)
    f = powi10(digits)
    rnd = _round_half_away
    out = x.copy()
    fin = np.isfinite(x)
    if digits == 0:
        out[fin] = rnd(x[fin])
    elif np.isfinite(f) and f != 0.0:
        with np.errstate(over="ignore", invalid="ignore"):
            out[fin] = rnd(x[fin] * f) / f
    return out


def angle_real(x):
    """0.0f64.atan2(x), simple_provider.rs:5502."""
    return np.arctan2(np.zeros_like(np.asarray(x, dtype=np.float64)), np.asarray(x, dtype=np.float64))


def chol(a: np.ndarray):
    """(upper factor R with A = R'R, info) as chol.rs:374-433 computes them (info > 0: rows from info - 1 on are zero)."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    out = np.empty(max(n * n, 1))
    l = lib()
    l.orc_chol.restype = C.c_uint
    l.orc_chol.argtypes = [_DP, C.c_size_t, _DP]
    info = l.orc_chol(_p(_f(a)), n, _p(out))
    return out[:n * n].reshape((n, n), order="F"), int(info)


NORM_ORDERS = {"one": 1, "two": 2, "inf": 3, "-inf": 4, "zero": 5, "fro": 6, "nuc": 7, "p": 8}


def norm(x: np.ndarray, order="two", p: float = 2.0):
    """norm.rs:269-529 for real data: the value, or None where the CPU refuses / needs an SVD (matrix 2-norm, nuclear, vector p < 1)."""
    x = np.asarray(x, dtype=np.float64)
    rows = x.shape[0] if x.ndim >= 1 else 1
    cols = x.shape[1] if x.ndim >= 2 else 1
    is_matrix = not (x.ndim <= 1 or rows <= 1 or cols <= 1)
    l = lib()
    l.orc_norm.restype = C.c_double
    l.orc_norm.argtypes = [_DP, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
    refused = C.c_int(0)
    v = l.orc_norm(_p(_f(x)), rows if is_matrix else x.size, cols if is_matrix else 1, int(is_matrix), NORM_ORDERS[order], float(p), C.byref(refused))
    return None if refused.value else float(v)
