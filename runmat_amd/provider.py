"""HipProvider -- host-side mirror of the reference's `trait AccelProvider`
(crates/runmat-accelerate-api/src/lib.rs:1386-3151) for the dense-array hot path, calling the C ABI
of librmhip.so (include/rmhip.h) through ctypes.

Method names, argument meaning and error behaviour follow the trait so the parity tests read like
the reference's own provider tests (crates/runmat-runtime-integration-tests/tests/gpu.rs,
crates/runmat-accelerate/tests/*.rs): every op returns a NEW handle, inputs are never mutated,
unsupported requests raise ProviderError (the reference's callers treat a provider Err as "fall
back to the CPU builtin", mtimes.rs:212-216) -- and there is no CPU fallback in here.

The Rust binding a RunMat maintainer would add is the same sequence of calls
(shim/hip_provider.rs).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

# op codes: the enums of include/rmhip.h, via the generated table (runmat_amd/_abi.py)
def _ops(enum: str, prefix: str) -> dict:
    return {k[len(prefix):].lower(): v for k, v in _lib.ENUMS[enum].items() if not k.endswith("_COUNT")}


BINARY_OPS = _ops("rmhip_binary_op", "RMHIP_")
UNARY_OPS = _ops("rmhip_unary_op", "RMHIP_")
SCALAR_OPS = _ops("rmhip_scalar_op", "RMHIP_S")
REDUCE_OPS = _ops("rmhip_reduce_op", "RMHIP_R")
# trait method -> unary op code name (lib.rs:2077-2331, 2053-2068, 2980-2988)
UNARY_HOOKS = {f"unary_{n}": n for n in UNARY_OPS}  # unary_sin ... (plus unary_<op> for every op code without a trait method of that name)
UNARY_HOOKS.update({"unary_pow2": "exp2", "logical_not": "not", "logical_isnan": "isnan", "logical_isinf": "isinf",
                    "logical_isfinite": "isfinite", "map_nan_to_zero": "nan_to_zero", "not_nan_mask": "not_nan"})


class ProviderError(RuntimeError):
    """`anyhow::Error` of a provider call. `.code` is the RMHIP_ERR_* status."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


@dataclass(frozen=True)
class GpuTensorHandle:
    """lib.rs:260-264 -- plain data; the provider owns the buffer until `free`."""
    shape: Tuple[int, ...]
    device_id: int
    buffer_id: int


@dataclass(frozen=True)
class ReductionFlavor:
    """lib.rs:865-890"""
    kind: str  # "sum" | "mean" | "custom"
    scale: float = 1.0

    @staticmethod
    def Sum() -> "ReductionFlavor":
        return ReductionFlavor("sum")

    @staticmethod
    def Mean() -> "ReductionFlavor":
        return ReductionFlavor("mean")

    @staticmethod
    def CustomScale(s: float) -> "ReductionFlavor":
        return ReductionFlavor("custom", float(s))


@dataclass
class ProviderCholResult:
    """`ProviderCholResult` (lib.rs:658-662)."""
    factor: "GpuTensorHandle"
    info: int


@dataclass
class ProviderFindResult:
    """`ProviderFindResult` (lib.rs:623-628); `values` is always present here."""
    linear: "GpuTensorHandle"
    rows: "GpuTensorHandle"
    cols: "GpuTensorHandle"
    values: "GpuTensorHandle"


@dataclass
class SortResult:
    """`SortResult` (lib.rs:1085-1088): host tensors."""
    values: np.ndarray
    indices: np.ndarray


@dataclass
class ReduceDimResult:
    """`ReduceDimResult` (lib.rs:513-517)."""
    values: "GpuTensorHandle"
    indices: "GpuTensorHandle"


@dataclass
class ProviderLuResult:
    """lib.rs:649-698"""
    combined: GpuTensorHandle
    lower: GpuTensorHandle
    upper: GpuTensorHandle
    perm_matrix: GpuTensorHandle
    perm_vector: GpuTensorHandle


@dataclass
class ProviderLinsolveOptions:
    """lib.rs:679-690"""
    lower: bool = False
    upper: bool = False
    rectangular: bool = False
    transposed: bool = False
    conjugate: bool = False
    symmetric: bool = False
    posdef: bool = False
    need_rcond: bool = False
    rcond: Optional[float] = None


@dataclass
class ProviderLinsolveResult:
    """lib.rs:692-696"""
    solution: GpuTensorHandle
    reciprocal_condition: float


def _shape_array(shape: Sequence[int]):
    arr = (C.c_size_t * max(len(shape), 1))(*[int(s) for s in shape])
    return arr, len(shape)


class HipProvider:
    """One provider per GPU (one process per GPU in multi-GPU jobs)."""

    _next_device_id = 1  # next_device_id(), lib.rs:3279

    def __init__(self, device_ordinal: int = 0, precision: str = "F64"):
        """`precision`: "F64" (default) or "F32" -- `ProviderPrecision` (lib.rs:815-818), fixed for the provider's
        lifetime.  At F32 tensors live in HBM as f32; the host boundary stays f64 (upload rounds, download widens)."""
        if precision not in ("F64", "F32"):
            raise ValueError("precision must be 'F64' or 'F32'")
        self._lib = _lib.load()
        ctx = C.c_void_p()
        rc = self._lib.rmhip_init(int(device_ordinal), C.byref(ctx))
        if rc != _lib.OK:
            raise ProviderError(rc, _lib.last_error())
        self._ctx = ctx
        self._precision = precision
        if precision == "F32":
            self._check(self._lib.rmhip_set_precision(self._ctx, 32))
        self._device_id = HipProvider._next_device_id
        HipProvider._next_device_id += 1
        self.device_ordinal = device_ordinal

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != _lib.OK:
            raise ProviderError(rc, _lib.last_error())

    def _handle(self, buffer_id: int, shape: Optional[Sequence[int]] = None) -> GpuTensorHandle:
        if shape is None:
            rank = C.c_size_t(16)
            buf = (C.c_size_t * 16)()
            self._check(self._lib.rmhip_shape(self._ctx, buffer_id, C.byref(rank), buf))
            shape = tuple(int(buf[i]) for i in range(rank.value))
        return GpuTensorHandle(tuple(int(s) for s in shape), self._device_id, int(buffer_id))

    def _shader_bytes(self, shader: str) -> bytes:
        """One bytes object per shader text: the library recognises a repeated (pointer, length) before it hashes the text
        (RunMat hands it the plan's cached `Arc<str>` the same way)."""
        cache = self.__dict__.setdefault("_shader_cache", {})
        b = cache.get(shader)
        if b is None:
            if len(cache) > 4096:
                cache.clear()
            b = cache[shader] = shader.encode()
        return b

    def _id(self, h: GpuTensorHandle) -> int:
        if h.device_id != self._device_id:  # io.rs:269-275: foreign handles are an error
            raise ProviderError(_lib.ERR_INVALID, f"handle belongs to device {h.device_id}, not {self._device_id}")
        return h.buffer_id

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.rmhip_shutdown(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self) -> None:
        self._check(self._lib.rmhip_synchronize(self._ctx))

    # -- identity -------------------------------------------------------------------------------
    def device_id(self) -> int:
        return self._device_id

    def precision(self) -> str:
        return self._precision  # ProviderPrecision (lib.rs:815-818); the fusion emitter picks `scalar_ty` from it

    def scalar_ty(self) -> str:
        """WGSL scalar type the planner emits for this provider (fusion.rs:1525-1533)."""
        return "f32" if self._precision == "F32" else "f64"

    def buffer_bits(self, h: GpuTensorHandle) -> int:
        bits = C.c_int()
        self._check(self._lib.rmhip_buffer_bits(self._ctx, self._id(h), C.byref(bits)))
        return bits.value

    def device_info_struct(self) -> dict:
        info = _lib.DeviceInfo()
        self._check(self._lib.rmhip_device_info(self._ctx, C.byref(info)))
        return {f: (getattr(info, f).decode() if isinstance(getattr(info, f), bytes) else getattr(info, f))
                for f, _ in info._fields_}

    def device_info(self) -> str:
        """`device_info` (lib.rs:1390): the one-line description."""
        i = self.device_info_struct()
        return f"{i['name']} ({i['arch']}, {i['backend']})"

    def default_reduction_workgroup_size(self) -> int:
        return int(self.device_info_struct()["reduction_workgroup_size"])  # lib.rs:3048

    def two_pass_threshold(self) -> int:
        return int(self.device_info_struct()["two_pass_threshold"])  # lib.rs:3053

    def fused_cache_counters(self) -> Tuple[int, int]:
        """(hits, misses) of the fused-kernel cache (lib.rs:3014)."""
        t = _lib.Telemetry()
        self._check(self._lib.rmhip_telemetry(self._ctx, C.byref(t)))
        return int(t.fusion_cache_hits), int(t.fusion_cache_misses)

    # -- memory ---------------------------------------------------------------------------------
    def upload(self, data, shape: Optional[Sequence[int]] = None) -> GpuTensorHandle:
        """`upload(&HostTensorView{data,shape})`: column-major f64 (lib.rs:3362-3372)."""
        arr = np.asarray(data, dtype=np.float64)
        if shape is None and arr.ndim == 2 and arr.flags.c_contiguous and not arr.flags.f_contiguous and arr.size >= 1 << 16:
            # A row-major numpy matrix is the column-major storage of its transpose: hand the bytes over as they are
            # and let the device transpose (numpy's strided gather runs at ~0.15 GB/s on an 8192x8192 matrix).
            # RunMat itself always passes column-major data; this is a convenience of the Python mirror.
            base = self.upload(arr.reshape(-1), (arr.shape[1], arr.shape[0]))
            out = self.transpose(base)
            self.free(base)
            return out
        if shape is None:
            shape = arr.shape if arr.ndim >= 2 else (arr.size, 1) if arr.ndim == 1 else (1, 1)
            flat = np.ascontiguousarray(arr.reshape(-1, order="F"))
        else:
            flat = np.ascontiguousarray(arr.reshape(-1))
        if flat.size != int(np.prod(shape, dtype=np.int64)):
            raise ProviderError(_lib.ERR_SHAPE, "upload: data length does not match shape")
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_upload(self._ctx, flat.ctypes.data_as(C.POINTER(C.c_double)), sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def download(self, h: GpuTensorHandle) -> np.ndarray:
        """Returns the column-major flat data (`HostTensorOwned.data`); a complex-interleaved tensor (`storage`, lib.rs:3362-3366)
        comes back as a complex128 array of the logical length (its re, im pairs viewed as complex numbers)."""
        n = int(np.prod(h.shape, dtype=np.int64)) if len(h.shape) else 1
        if self.is_complex(h):
            out = np.empty(2 * n, dtype=np.float64)
            self._check(self._lib.rmhip_download(self._ctx, self._id(h), out.ctypes.data_as(C.POINTER(C.c_double)), 2 * n))
            return out.view(np.complex128)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.rmhip_download(self._ctx, self._id(h), out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def is_complex(self, h: GpuTensorHandle) -> bool:
        """`handle_storage(h) == GpuTensorStorage::ComplexInterleaved` (lib.rs:588-594), asked of the library."""
        res = C.c_int()
        self._check(self._lib.rmhip_storage(self._ctx, self._id(h), C.byref(res)))
        return bool(res.value)

    def download_matrix(self, h: GpuTensorHandle) -> np.ndarray:
        """Convenience: data reshaped (Fortran order) to the handle's shape."""
        return self.download(h).reshape(h.shape, order="F")

    def free(self, h: GpuTensorHandle) -> None:
        self._check(self._lib.rmhip_free(self._ctx, self._id(h)))

    def fill(self, shape: Sequence[int], value: float) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_fill(self._ctx, float(value), sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def zeros(self, shape: Sequence[int]) -> GpuTensorHandle:
        return self.fill(shape, 0.0)

    def ones(self, shape: Sequence[int]) -> GpuTensorHandle:
        return self.fill(shape, 1.0)

    def fill_uniform(self, seed: int, lo: float, hi: float, shape: Sequence[int]) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_fill_uniform(self._ctx, int(seed), float(lo), float(hi), sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def reshape(self, h: GpuTensorHandle, shape: Sequence[int]) -> GpuTensorHandle:
        """`reshape` (lib.rs:2676-2684): the SAME buffer with a new shape - the returned handle carries `h`'s
        buffer_id, and one `free` releases the storage (a transpose view is materialised first)."""
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_reshape(self._ctx, self._id(h), sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def wrap_external(self, device_ptr: int, shape: Sequence[int]) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_wrap_external(self._ctx, C.c_void_p(device_ptr), sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def device_ptr(self, h: GpuTensorHandle) -> int:
        p = self._lib.rmhip_device_ptr(self._ctx, self._id(h))
        return int(p or 0)

    def set_stream(self, stream_ptr: int) -> None:
        self._check(self._lib.rmhip_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    # -- fused kernels --------------------------------------------------------------------------
    def fused_elementwise(self, shader: str, inputs: Sequence[GpuTensorHandle], output_shape: Sequence[int],
                          length: int) -> GpuTensorHandle:
        return self.fused_elementwise_multi(shader, inputs, output_shape, length, 1)[0]

    def fused_elementwise_multi(self, shader: str, inputs: Sequence[GpuTensorHandle], output_shape: Sequence[int],
                                length: int, num_outputs: int) -> List[GpuTensorHandle]:
        ids = (C.c_uint64 * max(len(inputs), 1))(*[self._id(h) for h in inputs])
        sh, rank = _shape_array(output_shape)
        outs = (C.c_uint64 * max(num_outputs, 1))()
        self._check(self._lib.rmhip_fused_elementwise(self._ctx, self._shader_bytes(shader), ids, len(inputs), sh, rank,
                                                      int(length), int(num_outputs), outs))
        return [self._handle(outs[k], output_shape) for k in range(num_outputs)]

    def fused_reduction(self, shader: str, inputs: Sequence[GpuTensorHandle], output_shape: Sequence[int],
                        reduce_len: int, num_slices: int, workgroup_size: int,
                        flavor: ReductionFlavor) -> GpuTensorHandle:
        ids = (C.c_uint64 * max(len(inputs), 1))(*[self._id(h) for h in inputs])
        sh, rank = _shape_array(output_shape)
        out = C.c_uint64()
        code = {"sum": 0, "mean": 1, "custom": 2}[flavor.kind]
        self._check(self._lib.rmhip_fused_reduction(self._ctx, self._shader_bytes(shader), ids, len(inputs), sh, rank,
                                                    int(reduce_len), int(num_slices), int(workgroup_size), code,
                                                    float(flavor.scale), C.byref(out)))
        return self._handle(out.value, output_shape)

    # -- per-op kernels -------------------------------------------------------------------------
    def _binary(self, op: str, a: GpuTensorHandle, b: GpuTensorHandle) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_binary(self._ctx, BINARY_OPS[op], self._id(a), self._id(b), C.byref(out)))
        return self._handle(out.value)

    def _unary(self, op: str, a: GpuTensorHandle) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_unary(self._ctx, UNARY_OPS[op], self._id(a), C.byref(out)))
        return self._handle(out.value, a.shape)

    def _scalar(self, op: str, a: GpuTensorHandle, s: float) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_scalar(self._ctx, SCALAR_OPS[op], self._id(a), float(s), C.byref(out)))
        return self._handle(out.value, a.shape)

    def elem_add(self, a, b): return self._binary("add", a, b)
    def elem_sub(self, a, b): return self._binary("sub", a, b)
    def elem_mul(self, a, b): return self._binary("mul", a, b)
    def elem_div(self, a, b): return self._binary("div", a, b)
    def elem_pow(self, a, b): return self._binary("pow", a, b)
    def elem_max(self, a, b): return self._binary("max", a, b)
    def elem_min(self, a, b): return self._binary("min", a, b)
    def elem_hypot(self, a, b): return self._binary("hypot", a, b)
    def elem_atan2(self, a, b): return self._binary("atan2", a, b)
    # comparisons / logicals (lib.rs:1939-2068): 1.0 / 0.0 tensors
    def elem_eq(self, a, b): return self._binary("eq", a, b)
    def elem_ne(self, a, b): return self._binary("ne", a, b)
    def elem_lt(self, a, b): return self._binary("lt", a, b)
    def elem_le(self, a, b): return self._binary("le", a, b)
    def elem_gt(self, a, b): return self._binary("gt", a, b)
    def elem_ge(self, a, b): return self._binary("ge", a, b)
    def logical_and(self, a, b): return self._binary("and", a, b)
    def logical_or(self, a, b): return self._binary("or", a, b)
    def logical_xor(self, a, b): return self._binary("xor", a, b)

    def scalar_add(self, a, s): return self._scalar("add", a, s)
    def scalar_sub(self, a, s): return self._scalar("sub", a, s)
    def scalar_mul(self, a, s): return self._scalar("mul", a, s)
    def scalar_div(self, a, s): return self._scalar("div", a, s)
    def scalar_rsub(self, a, s): return self._scalar("rsub", a, s)
    def scalar_rdiv(self, a, s): return self._scalar("rdiv", a, s)
    def scalar_max(self, a, s): return self._scalar("max", a, s)
    def scalar_min(self, a, s): return self._scalar("min", a, s)

    # unary_sin, unary_cos, ... logical_not / isnan / isinf / isfinite, map_nan_to_zero, not_nan_mask (lib.rs:2077-2331, 2053-2068,
    # 2980-2988) are attached below the class from UNARY_HOOKS: one method per trait hook, all through rmhip_unary.

    # -- shape / indexing hooks -------------------------------------------------------------------
    def repmat(self, a: GpuTensorHandle, reps: Sequence[int]) -> GpuTensorHandle:
        """`repmat` (lib.rs:2689-2695).  The result is a view of `a`'s storage; elem_* / fused_elementwise read it in
        place, anything else materialises it on first use (include/rmhip.h)."""
        arr, n = _shape_array(reps)
        out = C.c_uint64()
        self._check(self._lib.rmhip_repmat(self._ctx, self._id(a), arr, n, C.byref(out)))
        return self._handle(out.value)

    def permute(self, a: GpuTensorHandle, order: Sequence[int]) -> GpuTensorHandle:
        """`permute` (lib.rs:2579-2585): `order` is zero-based."""
        arr, n = _shape_array(order)
        out = C.c_uint64()
        self._check(self._lib.rmhip_permute(self._ctx, self._id(a), arr, n, C.byref(out)))
        return self._handle(out.value)

    def fill_like(self, prototype: GpuTensorHandle, value: float) -> GpuTensorHandle:
        """`fill_like` (lib.rs:1524-1545)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_fill_like(self._ctx, self._id(prototype), float(value), C.byref(out)))
        return self._handle(out.value, prototype.shape)

    def zeros_like(self, prototype): return self.fill_like(prototype, 0.0)  # lib.rs:1497
    def ones_like(self, prototype): return self.fill_like(prototype, 1.0)   # lib.rs:1547

    def read_scalar(self, h: GpuTensorHandle, linear_index: int) -> float:
        """`read_scalar` (lib.rs:1463): zero-based column-major index; out of range raises."""
        if linear_index < 0:
            raise ProviderError(_lib.ERR_INVALID, "read_scalar: negative index")
        out = C.c_double()
        self._check(self._lib.rmhip_read_scalar(self._ctx, self._id(h), int(linear_index), C.byref(out)))
        return out.value

    def gather_linear(self, source: GpuTensorHandle, indices: Sequence[int], output_shape: Sequence[int]) -> GpuTensorHandle:
        """`gather_linear` (lib.rs:1423-1430): zero-based u32 linear indices."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        sh, rank = _shape_array(output_shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_gather_linear(self._ctx, self._id(source), idx.ctypes.data_as(C.POINTER(C.c_uint32)), idx.size, sh,
                                                  rank, C.byref(out)))
        return self._handle(out.value, output_shape)

    def scatter_linear(self, target: GpuTensorHandle, indices: Sequence[int], values: GpuTensorHandle) -> None:
        """`scatter_linear` (lib.rs:1438-1445): updates `target` in place."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        self._check(self._lib.rmhip_scatter_linear(self._ctx, self._id(target), idx.ctypes.data_as(C.POINTER(C.c_uint32)), idx.size,
                                                   self._id(values)))

    def eye(self, shape: Sequence[int]) -> GpuTensorHandle:
        """`eye` (lib.rs:1552): identity on the first two dimensions of every page; [n] means [n, n]."""
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_eye(self._ctx, sh, rank, C.byref(out)))
        return self._handle(out.value)

    def eye_like(self, prototype: GpuTensorHandle) -> GpuTensorHandle:
        return self.eye(prototype.shape)  # lib.rs:1557

    def flip(self, a: GpuTensorHandle, axes: Sequence[int]) -> GpuTensorHandle:
        """`flip` (lib.rs:2586): zero-based axes."""
        arr, n = _shape_array(axes)
        out = C.c_uint64()
        self._check(self._lib.rmhip_flip(self._ctx, self._id(a), arr, n, C.byref(out)))
        return self._handle(out.value, a.shape)

    def circshift(self, a: GpuTensorHandle, shifts: Sequence[int]) -> GpuTensorHandle:
        """`circshift` (lib.rs:2589-2595): signed shifts per dimension."""
        arr = (C.c_longlong * max(len(shifts), 1))(*[int(v) for v in shifts])
        out = C.c_uint64()
        self._check(self._lib.rmhip_circshift(self._ctx, self._id(a), arr, len(shifts), C.byref(out)))
        return self._handle(out.value, a.shape)

    def _tri(self, a: GpuTensorHandle, upper: bool, offset: int) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_tri(self._ctx, self._id(a), 1 if upper else 0, int(offset), C.byref(out)))
        return self._handle(out.value, a.shape)

    def tril(self, a, offset: int = 0): return self._tri(a, False, offset)  # lib.rs:1635
    def triu(self, a, offset: int = 0): return self._tri(a, True, offset)   # lib.rs:1644

    def cat(self, dim: int, inputs: Sequence[GpuTensorHandle]) -> GpuTensorHandle:
        """`cat` (lib.rs:2686): ONE-based dimension."""
        ids = (C.c_uint64 * max(len(inputs), 1))(*[self._id(h) for h in inputs])
        out = C.c_uint64()
        self._check(self._lib.rmhip_cat(self._ctx, int(dim), ids, len(inputs), C.byref(out)))
        return self._handle(out.value)

    def linspace(self, start: float, stop: float, count: int) -> GpuTensorHandle:
        """`linspace` (lib.rs:1887) -> [1, count]."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_linspace(self._ctx, float(start), float(stop), int(count), C.byref(out)))
        return self._handle(out.value, (1, int(count)))

    # -- reductions -----------------------------------------------------------------------------
    def _reduce(self, op: str, a: GpuTensorHandle, dim: int, omitnan: bool = False) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_reduce(self._ctx, REDUCE_OPS[op], self._id(a), int(dim), 1 if omitnan else 0,
                                           C.byref(out)))
        return self._handle(out.value)

    def reduce_sum(self, a): return self._reduce("sum", a, -1)
    def reduce_sum_dim(self, a, dim): return self._reduce("sum", a, dim)
    def reduce_mean(self, a): return self._reduce("mean", a, -1)
    def reduce_mean_dim(self, a, dim): return self._reduce("mean", a, dim)
    def reduce_min(self, a): return self._reduce("min", a, -1)
    def reduce_max(self, a): return self._reduce("max", a, -1)

    def _reduce_minmax_dim(self, op: str, a: GpuTensorHandle, dim: int, omitnan: bool) -> "ReduceDimResult":
        values, indices = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_reduce_minmax_dim(self._ctx, REDUCE_OPS[op], self._id(a), int(dim), 1 if omitnan else 0,
                                                      C.byref(values), C.byref(indices)))
        return ReduceDimResult(self._handle(values.value), self._handle(indices.value))

    def reduce_min_dim(self, a, dim, omitnan: bool = False) -> "ReduceDimResult":
        """lib.rs:2864-2870 -> `ReduceDimResult{values, indices}` (:513-517): minimum along zero-based `dim` and its 1-based position
        (first occurrence, -0 below +0, the first NaN of a slice wins unless omitnan: min.rs:1443-1531)."""
        return self._reduce_minmax_dim("min", a, dim, omitnan)

    def reduce_max_dim(self, a, dim, omitnan: bool = False) -> "ReduceDimResult":
        """lib.rs:2877-2883; max.rs:1715-1727."""
        return self._reduce_minmax_dim("max", a, dim, omitnan)

    def reduce_std(self, a, normalization: str = "sample", omitnan: bool = False) -> GpuTensorHandle:
        """lib.rs:2786-2793 (`ProviderStdNormalization::{Sample, Population}`, `ProviderNanMode`): std of all elements -> [1,1]."""
        return self.reduce_std_dim(a, -1, normalization, omitnan)

    def reduce_std_dim(self, a, dim: int, normalization: str = "sample", omitnan: bool = False) -> GpuTensorHandle:
        """lib.rs:2794-2802; CPU semantics std.rs:858-935."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_reduce_std(self._ctx, self._id(a), int(dim), {"sample": 0, "population": 1}[normalization],
                                               1 if omitnan else 0, C.byref(out)))
        return self._handle(out.value)

    def _truth(self, op: int, a, dim: int, omit_nan: bool) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_reduce_truth(self._ctx, op, self._id(a), int(dim), 1 if omit_nan else 0, C.byref(out)))
        return self._handle(out.value)

    def reduce_nnz(self, a): return self._truth(0, a, -1, False)            # lib.rs:2730-2735
    def reduce_nnz_dim(self, a, dim): return self._truth(0, a, dim, False)  # lib.rs:2736-2742
    def reduce_any(self, a, omit_nan: bool = False): return self._truth(1, a, -1, omit_nan)            # lib.rs:2803-2809
    def reduce_any_dim(self, a, dim, omit_nan: bool = False): return self._truth(1, a, dim, omit_nan)  # lib.rs:2810-2817
    def reduce_all(self, a, omit_nan: bool = False): return self._truth(2, a, -1, omit_nan)            # lib.rs:2818-2824
    def reduce_all_dim(self, a, dim, omit_nan: bool = False): return self._truth(2, a, dim, omit_nan)  # lib.rs:2825-2832

    def _cumulative(self, op: int, a, dim: int, reverse: bool, omitnan: bool) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_cumulative(self._ctx, op, self._id(a), int(dim), 1 if reverse else 0, 1 if omitnan else 0, C.byref(out)))
        return self._handle(out.value)

    def cumsum_scan(self, a, dim: int, reverse: bool = False, omitnan: bool = False) -> GpuTensorHandle:
        """lib.rs:2884-2891 (`ProviderScanDirection`, `ProviderNanMode`); zero-based dim; cumsum.rs:559-650."""
        return self._cumulative(0, a, dim, reverse, omitnan)

    def cumprod_scan(self, a, dim: int, reverse: bool = False, omitnan: bool = False) -> GpuTensorHandle:
        """lib.rs:2908-2915; cumprod.rs:581-670."""
        return self._cumulative(1, a, dim, reverse, omitnan)

    def _cumextreme(self, is_max: bool, a, dim: int, reverse: bool, omitnan: bool) -> "ReduceDimResult":
        values, indices = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_cumextreme(self._ctx, 1 if is_max else 0, self._id(a), int(dim), 1 if reverse else 0, 1 if omitnan else 0,
                                               C.byref(values), C.byref(indices)))
        return ReduceDimResult(self._handle(values.value), self._handle(indices.value))

    def cummin_scan(self, a, dim: int, reverse: bool = False, omitnan: bool = False) -> "ReduceDimResult":
        """lib.rs:2918-2926 -> `ProviderCumminResult{values, indices}` (:520-523); zero-based dim < rank; cummin.rs:719-876."""
        return self._cumextreme(False, a, dim, reverse, omitnan)

    def cummax_scan(self, a, dim: int, reverse: bool = False, omitnan: bool = False) -> "ReduceDimResult":
        """lib.rs:2927-2935; cummax.rs (the mirror image of cummin.rs)."""
        return self._cumextreme(True, a, dim, reverse, omitnan)

    def diff_dim(self, a, order: int, dim: int, column_major: bool = False) -> GpuTensorHandle:
        """lib.rs:2596-2603: `order` first differences along zero-based dim, in the reference's output order (k fastest inside every
        line, diff.rs:493-503) unless column_major; the handle carries the shape diff_tensor_host reports."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_diff_dim(self._ctx, self._id(a), int(order), int(dim), 1 if column_major else 0, C.byref(out)))
        return self._handle(out.value)

    def sort_dim(self, a, dim: int, order: str = "ascend", comparison: str = "auto") -> "SortResult":
        """lib.rs:2358-2366 -> `SortResult{values, indices}` with HOST tensors (:1085-1088): stable sort of every line along zero-based dim
        (`SortOrder::{Ascend, Descend}`, `SortComparison::{Auto, Real, Abs}`); indices are 1-based original positions."""
        if order not in ("ascend", "descend") or comparison not in ("auto", "real", "abs"):
            raise RmhipError(1, f"sort_dim: order {order!r} / comparison {comparison!r}")
        sv, si = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_sort_dim(self._ctx, self._id(a), int(dim), 1 if order == "descend" else 0, 1 if comparison == "abs" else 0,
                                             C.byref(sv), C.byref(si)))
        hv, hi = self._handle(sv.value), self._handle(si.value)
        try:
            return SortResult(self.download_matrix(hv), self.download_matrix(hi))
        finally:
            self.free(hv)
            self.free(hi)

    def sort_rows(self, a, columns: Sequence[Tuple[int, str]], comparison: str = "auto") -> "SortResult":
        """lib.rs:2367-2374: `columns` as (zero-based index, "ascend" | "descend") pairs (`SortRowsColumnSpec`) -> `SortResult` (host tensors)."""
        if comparison not in ("auto", "real", "abs") or any(o not in ("ascend", "descend") for _, o in columns):
            raise RmhipError(1, f"sort_rows: columns {columns!r} / comparison {comparison!r}")
        n = len(columns)
        idx = (C.c_size_t * max(n, 1))(*[int(i) for i, _ in columns])
        desc = (C.c_int * max(n, 1))(*[1 if o == "descend" else 0 for _, o in columns])
        sv, si = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_sort_rows(self._ctx, self._id(a), idx, desc, n, 1 if comparison == "abs" else 0, C.byref(sv), C.byref(si)))
        hv, hi = self._handle(sv.value), self._handle(si.value)
        try:
            return SortResult(self.download_matrix(hv), self.download_matrix(hi))
        finally:
            self.free(hv)
            self.free(hi)

    def find(self, a, limit: Optional[int] = None, direction: str = "first") -> "ProviderFindResult":
        """lib.rs:2937-2944 (`FindDirection::{First, Last}`) -> `ProviderFindResult{linear, rows, cols, values}` (:623-628)."""
        if direction not in ("first", "last"):
            raise RmhipError(1, f"find: direction {direction!r}")
        outs = [C.c_uint64() for _ in range(4)]
        self._check(self._lib.rmhip_find(self._ctx, self._id(a), -1 if limit is None else int(limit), 1 if direction == "last" else 0,
                                         *[C.byref(o) for o in outs]))
        return ProviderFindResult(*[self._handle(o.value) for o in outs])

    def reduce_median(self, a) -> GpuTensorHandle:
        """lib.rs:2833-2838: the median of ALL elements -> [1, 1] (NaN if any element is NaN; simple_provider.rs:7167-7193)."""
        return self.reduce_median_dim(a, -1)

    def reduce_median_dim(self, a, dim: int) -> GpuTensorHandle:
        """lib.rs:2839-2845: include-NaN median along zero-based dim (median.rs:644-741)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_reduce_median(self._ctx, self._id(a), int(dim), C.byref(out)))
        return self._handle(out.value)

    def reduce_prod(self, a): return self._reduce("prod", a, -1)
    def reduce_prod_dim(self, a, dim): return self._reduce("prod", a, dim)

    def reduce_mean_nd(self, a: GpuTensorHandle, dims_zero_based: Sequence[int], omitnan: bool = False) -> GpuTensorHandle:
        """lib.rs:2763-2769: mean over several zero-based dims (reduced extents become 1)."""
        return self._reduce_nd("mean", a, dims_zero_based, omitnan)

    def reduce_moments_nd(self, a: GpuTensorHandle, dims_zero_based: Sequence[int]) -> Tuple[GpuTensorHandle, GpuTensorHandle]:
        """lib.rs:2770-2778 -> `ProviderMoments2 { mean, ex2 }` (lib.rs:1317-1320): E[x] and E[x^2] over the dims."""
        dims = (C.c_size_t * max(len(dims_zero_based), 1))(*[int(d) for d in dims_zero_based])
        mean, ex2 = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_reduce_moments_nd(self._ctx, self._id(a), dims, len(dims_zero_based), C.byref(mean), C.byref(ex2)))
        return self._handle(mean.value), self._handle(ex2.value)

    def _reduce_nd(self, op: str, a: GpuTensorHandle, dims_zero_based: Sequence[int], omitnan: bool = False) -> GpuTensorHandle:
        dims = (C.c_size_t * max(len(dims_zero_based), 1))(*[int(d) for d in dims_zero_based])
        out = C.c_uint64()
        self._check(self._lib.rmhip_reduce_nd(self._ctx, REDUCE_OPS[op], self._id(a), dims, len(dims_zero_based),
                                              1 if omitnan else 0, C.byref(out)))
        return self._handle(out.value)

    def dot(self, a: GpuTensorHandle, b: GpuTensorHandle, dim: Optional[int] = None) -> GpuTensorHandle:
        """`dot(lhs, rhs, dim)` (lib.rs:2722): dim is zero-based, None = first non-singleton."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_dot(self._ctx, self._id(a), self._id(b), -1 if dim is None else int(dim), C.byref(out)))
        return self._handle(out.value)

    # -- linear algebra -------------------------------------------------------------------------
    def matmul(self, a: GpuTensorHandle, b: GpuTensorHandle) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_matmul(self._ctx, self._id(a), self._id(b), C.byref(out)))
        return self._handle(out.value, (a.shape[0], b.shape[1]))

    def matmul_epilogue(self, a: GpuTensorHandle, b: GpuTensorHandle, alpha: float = 1.0, beta: float = 0.0,
                        row_scale: Optional[GpuTensorHandle] = None, col_scale: Optional[GpuTensorHandle] = None,
                        row_op: str = "multiply", col_op: str = "multiply", clamp_min: Optional[float] = None,
                        clamp_max: Optional[float] = None, pow_exponent: Optional[float] = None,
                        diag_output: Optional[GpuTensorHandle] = None) -> GpuTensorHandle:
        """`matmul_epilogue(a, b, &MatmulEpilogue)` (lib.rs:2394-2405, 3498-3560), folded into the dgemm store."""
        ep = _lib.MatmulEpilogue(
            float(alpha), float(beta), self._id(row_scale) if row_scale else 0, self._id(col_scale) if col_scale else 0,
            1 if row_op == "divide" else 0, 1 if col_op == "divide" else 0, int(clamp_min is not None),
            int(clamp_max is not None), int(pow_exponent is not None), float(clamp_min or 0.0), float(clamp_max or 0.0),
            float(pow_exponent or 0.0), self._id(diag_output) if diag_output else 0)
        out = C.c_uint64()
        self._check(self._lib.rmhip_matmul_epilogue(self._ctx, self._id(a), self._id(b), C.byref(ep), C.byref(out)))
        return self._handle(out.value, (a.shape[0], b.shape[1]))

    def lu(self, a: GpuTensorHandle) -> ProviderLuResult:
        outs = (C.c_uint64 * 5)()
        self._check(self._lib.rmhip_lu(self._ctx, self._id(a), outs))
        hs = [self._handle(outs[i]) for i in range(5)]
        return ProviderLuResult(*hs)

    def mldivide(self, lhs: GpuTensorHandle, rhs: GpuTensorHandle) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_mldivide(self._ctx, self._id(lhs), self._id(rhs), C.byref(out)))
        return self._handle(out.value)

    def mrdivide(self, lhs: GpuTensorHandle, rhs: GpuTensorHandle) -> GpuTensorHandle:
        """`mrdivide` (lib.rs:2484-2490): X = lhs / rhs, i.e. X * rhs = lhs."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_mrdivide(self._ctx, self._id(lhs), self._id(rhs), C.byref(out)))
        return self._handle(out.value)

    def chol(self, a: GpuTensorHandle, lower: bool = False) -> "ProviderCholResult":
        """`chol` (lib.rs:2502-2508) -> `ProviderCholResult{factor, info}` (:658-662): the success path only (info == 0); a matrix that is
        not symmetric / positive definite raises (UNSUPPORTED) and chol.rs:331-342 takes its host path."""
        out, info = C.c_uint64(), C.c_uint()
        self._check(self._lib.rmhip_chol(self._ctx, self._id(a), 1 if lower else 0, C.byref(out), C.byref(info)))
        return ProviderCholResult(self._handle(out.value), int(info.value))

    def inv(self, matrix: GpuTensorHandle, options=None) -> GpuTensorHandle:
        """`inv` (lib.rs:2430-2436; `ProviderInvOptions {}`): X = A \\ I on the LU path; SINGULAR -> the caller's CPU path (inv.rs:225-227)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_inv(self._ctx, self._id(matrix), C.byref(out)))
        return self._handle(out.value)

    def stochastic_evolution(self, state: GpuTensorHandle, drift: float, scale: float, steps: int,
                             draws_per_step: int = 0) -> GpuTensorHandle:
        """lib.rs:1759-1769; `draws_per_step` > 0 selects the sharded form (see include/rmhip.h)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_stochastic_evolution_sharded(self._ctx, self._id(state), float(drift), float(scale),
                                                                 int(steps), int(draws_per_step), C.byref(out)))
        return self._handle(out.value, state.shape)

    def linsolve(self, lhs: GpuTensorHandle, rhs: GpuTensorHandle,
                 options: Optional[ProviderLinsolveOptions] = None) -> ProviderLinsolveResult:
        """lib.rs:2422-2429; CPU semantics linsolve.rs:691-726."""
        o = options or ProviderLinsolveOptions()
        co = _lib.LinsolveOptions(int(o.lower), int(o.upper), int(o.rectangular), int(o.transposed), int(o.conjugate),
                                  int(o.symmetric), int(o.posdef), int(o.need_rcond), int(o.rcond is not None),
                                  float(o.rcond) if o.rcond is not None else 0.0)
        out = C.c_uint64()
        rc = C.c_double(float("nan"))
        self._check(self._lib.rmhip_linsolve(self._ctx, self._id(lhs), self._id(rhs), C.byref(co), C.byref(out), C.byref(rc)))
        return ProviderLinsolveResult(self._handle(out.value), rc.value)

    def transpose(self, a: GpuTensorHandle) -> GpuTensorHandle:
        """lib.rs:2532: a view aliasing `a` (consumed in place by matmul / syrk, materialised on any other use)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_transpose(self._ctx, self._id(a), C.byref(out)))
        return self._handle(out.value)

    def matmul_power_step(self, lhs: GpuTensorHandle, rhs: GpuTensorHandle, epsilon: float = 0.0) -> GpuTensorHandle:
        """lib.rs:2414-2421 (PowerStepEpilogue.epsilon): lhs*rhs with every column divided by its 2-norm."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_matmul_power_step(self._ctx, self._id(lhs), self._id(rhs), float(epsilon), C.byref(out)))
        return self._handle(out.value)

    def image_normalize(self, x: GpuTensorHandle, batch: int, height: int, width: int, epsilon: float,
                        gain: Optional[float] = None, bias: Optional[float] = None, gamma: Optional[float] = None,
                        clamp_zero: bool = True) -> GpuTensorHandle:
        """lib.rs:2407-2413 with the fields of ImageNormalizeDescriptor (lib.rs:3563-3577; clamp_zero defaults to true)."""
        d = _lib.ImageNormalize(int(batch), int(height), int(width), float(epsilon), int(gain is not None), int(bias is not None),
                                int(gamma is not None), int(bool(clamp_zero)), float(gain or 0.0), float(bias or 0.0),
                                float(gamma or 0.0))
        out = C.c_uint64()
        self._check(self._lib.rmhip_image_normalize(self._ctx, self._id(x), C.byref(d), C.byref(out)))
        return self._handle(out.value, x.shape)

    def covariance(self, matrix: GpuTensorHandle, second=None, weights=None, biased: bool = False, rows: str = "all") -> GpuTensorHandle:
        """lib.rs:1857-1865: only the dense unweighted form (what the CenteredGram pattern issues) is offloaded."""
        if second is not None or weights is not None or rows != "all":
            raise ProviderError(_lib.ERR_UNSUPPORTED, "covariance: second matrix / weights / row filtering use the CPU path")
        out = C.c_uint64()
        self._check(self._lib.rmhip_covariance(self._ctx, self._id(matrix), int(bool(biased)), C.byref(out)))
        return self._handle(out.value)

    def covariance_to_correlation(self, matrix):
        """lib.rs:1876-1884 -> `ProviderCovarianceToCorrelationResult { correlation, sigma }` as a pair of handles."""
        corr, sig = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_covariance_to_correlation(self._ctx, self._id(matrix), C.byref(corr), C.byref(sig)))
        return self._handle(corr.value), self._handle(sig.value)

    def rank(self, matrix, tolerance: Optional[float] = None) -> GpuTensorHandle:
        """lib.rs:2464-2470 -> a [1, 1] tensor holding the numerical rank."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_rank(self._ctx, self._id(matrix), 0 if tolerance is None else 1, 0.0 if tolerance is None else float(tolerance), C.byref(out)))
        return self._handle(out.value)

    def cond(self, matrix, norm: str = "two") -> GpuTensorHandle:
        """lib.rs:2444-2450 (`ProviderCondNorm::{Two, One, Inf, Fro}`, :736-741): the 2-norm is served."""
        codes = {"two": 0, "one": 1, "inf": 2, "fro": 3}
        if norm not in codes:
            raise RmhipError(1, f"cond: norm {norm!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_cond(self._ctx, self._id(matrix), codes[norm], C.byref(out)))
        return self._handle(out.value)

    def rcond(self, matrix) -> GpuTensorHandle:
        """lib.rs:2471-2476 -> [1, 1]: s_min / s_max of a square matrix."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_rcond(self._ctx, self._id(matrix), C.byref(out)))
        return self._handle(out.value)

    def pinv(self, matrix, tolerance: Optional[float] = None) -> GpuTensorHandle:
        """lib.rs:2437-2443 (`ProviderPinvOptions { tolerance }`) -> [cols, rows]."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_pinv(self._ctx, self._id(matrix), 0 if tolerance is None else 1, 0.0 if tolerance is None else float(tolerance), C.byref(out)))
        return self._handle(out.value)

    def peaks(self, n: int) -> GpuTensorHandle:
        """lib.rs:1781-1785: Z of the peaks surface on the n x n grid over [-3, 3] x [-3, 3]."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_peaks(self._ctx, int(n), 0, 0, C.byref(out)))
        return self._handle(out.value)

    def peaks_xy(self, x, y) -> GpuTensorHandle:
        """lib.rs:1787-1795: the peaks formula at same-shape coordinate tensors."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_peaks(self._ctx, 0, self._id(x), self._id(y), C.byref(out)))
        return self._handle(out.value)

    def corrcoef(self, matrix: GpuTensorHandle, normalization: str = "unbiased", rows: str = "all") -> GpuTensorHandle:
        """lib.rs:1867-1874 (`CorrcoefOptions`, :906-911): only rows == "all" is offloaded (what corrcoef_try_gpu issues, corrcoef.rs:480-483)."""
        if normalization not in ("unbiased", "biased") or rows not in ("all", "complete", "pairwise"):
            raise RmhipError(1, f"corrcoef: normalization {normalization!r} / rows {rows!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_corrcoef(self._ctx, self._id(matrix), 1 if normalization == "biased" else 0, {"all": 0, "complete": 1, "pairwise": 2}[rows], C.byref(out)))
        return self._handle(out.value)

    def diag_extract(self, matrix: GpuTensorHandle, offset: int = 0) -> GpuTensorHandle:
        """lib.rs:1625-1632"""
        out = C.c_uint64()
        self._check(self._lib.rmhip_diag_extract(self._ctx, self._id(matrix), int(offset), C.byref(out)))
        return self._handle(out.value)

    def syrk(self, a: GpuTensorHandle) -> GpuTensorHandle:
        """lib.rs:2383: A' * A."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_syrk(self._ctx, self._id(a), C.byref(out)))
        return self._handle(out.value)

    # -- block-level building blocks (views; used by the multi-GPU solver, they mutate the buffer) ----
    def _view(self, v) -> "_lib.View":
        h, r0, c0, rows, cols = v
        return _lib.View(self._id(h), int(r0), int(c0), int(rows), int(cols))

    def blk_copy(self, view) -> GpuTensorHandle:
        out = C.c_uint64()
        v = self._view(view)
        self._check(self._lib.rmhip_blk_copy(self._ctx, C.byref(v), C.byref(out)))
        return self._handle(out.value, (view[3], view[4]))

    def blk_assign(self, view, src: GpuTensorHandle) -> None:
        v = self._view(view)
        self._check(self._lib.rmhip_blk_assign(self._ctx, C.byref(v), self._id(src)))

    def blk_gemm(self, alpha: float, a, b, beta: float, c) -> None:
        va, vb, vc = self._view(a), self._view(b), self._view(c)
        self._check(self._lib.rmhip_blk_gemm(self._ctx, float(alpha), C.byref(va), C.byref(vb), float(beta), C.byref(vc)))

    def blk_trsm(self, upper, t, b) -> None:
        """upper False / 0: B <- L^-1 B (unit lower); True / 1: B <- U^-1 B; 2 or "right": B <- B U^-1."""
        vt, vb = self._view(t), self._view(b)
        mode = 2 if upper in (2, "right") else (1 if upper else 0)
        self._check(self._lib.rmhip_blk_trsm(self._ctx, mode, C.byref(vt), C.byref(vb)))

    def blk_absmax(self, view) -> float:
        """max |a_ij| over a view; NaN if the view holds one (rmhip_blk_absmax: the multiplier guard of the row-partitioned solve)."""
        out = C.c_double()
        va = self._view(view)
        self._check(self._lib.rmhip_blk_absmax(self._ctx, C.byref(va), C.byref(out)))
        return out.value

    def matmul_row_sharded(self, a_rows: GpuTensorHandle, b: GpuTensorHandle, rows_total: int, gather: bool = False,
                           granule: int = 128) -> GpuTensorHandle:
        """rmhip_matmul_row_sharded: this rank's rows of C = A * B (B replicated); gather=True appends the row-block all-gather."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_matmul_row_sharded(self._ctx, self._id(a_rows), self._id(b), int(rows_total), int(granule),
                                                       1 if gather else 0, C.byref(out)))
        return self._handle(out.value)

    def comm_wait_bounded(self, timeout_s: float = 0.0) -> None:
        """rmhip_comm_wait_bounded: host-side wait for everything queued so far, bounded (the communicator is aborted on expiry)."""
        self._check(self._lib.rmhip_comm_wait_bounded(self._ctx, float(timeout_s)))

    def rp_phase_ms(self) -> dict:
        """rmhip_rp_phase_ms: device milliseconds of the last `mldivide_row_partitioned` call by phase."""
        out = (C.c_double * 4)()
        self._check(self._lib.rmhip_rp_phase_ms(self._ctx, out))
        return {"panel": out[0], "broadcast_wait": out[1], "update": out[2], "exchange": out[3]}

    def mldivide_row_partitioned(self, ab_local: GpuTensorHandle, n: int, nrhs: int, rb: int = 512, tau: float = 8.0) -> GpuTensorHandle:
        """rmhip_mldivide_row_partitioned: the row-partitioned A\\b driver inside the library (depth-1 look-ahead); raises ProviderError
        with code ERR_GROWTH (10) on every rank when the multiplier guard or any rank fails."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_mldivide_row_partitioned(self._ctx, self._id(ab_local), int(n), int(nrhs), int(rb), float(tau),
                                                             C.byref(out)))
        return self._handle(out.value, (int(n), int(nrhs)))

    def blk_lu(self, a) -> Tuple[GpuTensorHandle, int]:
        out, info = C.c_uint64(), C.c_int()
        va = self._view(a)
        self._check(self._lib.rmhip_blk_lu(self._ctx, C.byref(va), C.byref(out), C.byref(info)))
        return self._handle(out.value), int(info.value)

    def blk_swap_rows(self, a, ipiv: GpuTensorHandle) -> None:
        va = self._view(a)
        self._check(self._lib.rmhip_blk_swap_rows(self._ctx, C.byref(va), self._id(ipiv)))

    # -- multi-GPU collectives (include/rmhip.h; no counterpart in the reference) ------------------
    @staticmethod
    def comm_unique_id(transport: str = "rccl") -> bytes:
        """Rank 0 creates the id; the host distributes these 128 bytes to every rank (any transport)."""
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = lib.rmhip_comm_unique_id(_lib.COMM_HOST_SHM if transport in ("shm", "host") else _lib.COMM_RCCL, buf)
        if rc != _lib.OK:
            raise ProviderError(rc, _lib.last_error())
        return bytes(buf.raw)

    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        self._check(self._lib.rmhip_comm_init(self._ctx, buf, int(rank), int(world)))

    def comm_destroy(self) -> None:
        self._check(self._lib.rmhip_comm_destroy(self._ctx))

    def comm_abort(self) -> None:
        """rmhip_comm_abort: give up inside a sequence of collectives without leaving the peers blocked (they fail in their next barrier)."""
        self._check(self._lib.rmhip_comm_abort(self._ctx))

    def comm_rank(self) -> Tuple[int, int]:
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.rmhip_comm_rank(self._ctx, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def comm_barrier(self) -> None:
        self._check(self._lib.rmhip_comm_barrier(self._ctx))

    def comm_bcast(self, block, root: int, async_: bool = False) -> None:
        """In-place broadcast of a handle (whole buffer) or a view tuple (buf, row_off, col_off, rows, cols)."""
        if isinstance(block, GpuTensorHandle):
            rows = block.shape[0] if block.shape else 1
            cols = int(np.prod(block.shape[1:])) if len(block.shape) > 1 else 1
            block = (block, 0, 0, rows, cols)
        v = self._view(block)
        self._check(self._lib.rmhip_comm_bcast(self._ctx, C.byref(v), int(root), 1 if async_ else 0))

    def comm_wait(self) -> None:
        self._check(self._lib.rmhip_comm_wait(self._ctx))

    def comm_allgather_f64(self, local: GpuTensorHandle) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_comm_allgather_f64(self._ctx, self._id(local), C.byref(out)))
        return self._handle(out.value)

    def comm_allgather_rows(self, local: GpuTensorHandle, rows_total: int, granule: int = 128) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_comm_allgather_rows(self._ctx, self._id(local), int(rows_total), int(granule), C.byref(out)))
        return self._handle(out.value)

    # -- RNG ------------------------------------------------------------------------------------
    def set_rng_state(self, state: int) -> None:
        self._check(self._lib.rmhip_set_rng_state(self._ctx, int(state) & 0xFFFFFFFFFFFFFFFF))

    def set_lazy_random(self, enabled: bool, min_numel: int = 0) -> None:
        """rmhip_set_lazy_random: whether `random_normal` returns storage-less handles that a streaming fused elementwise kernel
        generates in registers (default on from 1024 elements on f64 providers); `min_numel` 0 keeps the threshold."""
        self._check(self._lib.rmhip_set_lazy_random(self._ctx, 1 if enabled else 0, int(min_numel)))

    def lazy_random_stats(self) -> dict:
        a, b, c_ = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_lazy_random_stats(self._ctx, C.byref(a), C.byref(b), C.byref(c_)))
        return {"created": int(a.value), "fused": int(b.value), "materialised": int(c_.value)}

    def get_rng_state(self) -> int:
        s = C.c_uint64()
        self._check(self._lib.rmhip_get_rng_state(self._ctx, C.byref(s)))
        return int(s.value)

    def rng_seed(self, seed: int) -> None:
        self._check(self._lib.rmhip_rng_seed(self._ctx, int(seed)))

    def random_uniform(self, shape: Sequence[int]) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_random_uniform(self._ctx, sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def random_normal(self, shape: Sequence[int]) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_random_normal(self._ctx, sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    # -- subscript / grid / slice-write hooks, per-element forms of a real tensor (index_ops.hip) -----
    def ndgrid(self, axes: Sequence[GpuTensorHandle], output_shape: Sequence[int], output_count: int) -> List[GpuTensorHandle]:
        """lib.rs:1567-1569 (`ProviderNdgridRequest{axes, output_shape, output_count}` -> `ProviderNdgridResult{outputs}`)."""
        ids = (C.c_uint64 * max(len(axes), 1))(*[self._id(a) for a in axes])
        sh, rank = _shape_array(output_shape)
        outs = (C.c_uint64 * max(int(output_count), 1))()
        self._check(self._lib.rmhip_ndgrid(self._ctx, ids, len(axes), sh, rank, int(output_count), outs))
        return [self._handle(outs[i]) for i in range(int(output_count))]

    def sub2ind(self, dims: Sequence[int], strides: Sequence[int], inputs: Sequence[GpuTensorHandle], scalar_mask: Sequence[bool], length: int,
                output_shape: Sequence[int]) -> GpuTensorHandle:
        """lib.rs:3084-3094."""
        if not (len(dims) == len(strides) == len(inputs) == len(scalar_mask)):
            raise RmhipError(1, f"sub2ind: expected {len(dims)} subscripts for {len(dims)} dimensions")
        n = len(dims)
        d, st = (C.c_size_t * max(n, 1))(*dims), (C.c_size_t * max(n, 1))(*strides)
        ids = (C.c_uint64 * max(n, 1))(*[self._id(h) for h in inputs])
        mask = (C.c_ubyte * max(n, 1))(*[1 if m else 0 for m in scalar_mask])
        sh, rank = _shape_array(output_shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_sub2ind(self._ctx, d, st, ids, mask, n, int(length), sh, rank, C.byref(out)))
        return self._handle(out.value)

    def supports_ind2sub(self) -> bool:
        """lib.rs:3097-3099."""
        return True

    def ind2sub(self, dims: Sequence[int], strides: Sequence[int], indices: GpuTensorHandle, total: int, length: int,
                output_shape: Sequence[int]) -> List[GpuTensorHandle]:
        """lib.rs:3102-3112: one subscript tensor per dimension."""
        n = len(dims)
        d, st = (C.c_size_t * max(n, 1))(*dims), (C.c_size_t * max(n, 1))(*strides)
        sh, rank = _shape_array(output_shape)
        outs = (C.c_uint64 * max(n, 1))()
        self._check(self._lib.rmhip_ind2sub(self._ctx, d, st, n, self._id(indices), int(total), int(length), sh, rank, outs))
        return [self._handle(outs[i]) for i in range(n)]

    def scatter_column(self, matrix, col_index: int, values) -> GpuTensorHandle:
        """lib.rs:3064-3071: a new handle = the matrix with column `col_index` (zero-based) replaced."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_scatter_line(self._ctx, self._id(matrix), 1, int(col_index), self._id(values), C.byref(out)))
        return self._handle(out.value)

    def scatter_row(self, matrix, row_index: int, values) -> GpuTensorHandle:
        """lib.rs:3075-3082."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_scatter_line(self._ctx, self._id(matrix), 0, int(row_index), self._id(values), C.byref(out)))
        return self._handle(out.value)

    def pow2_scale(self, mantissa, exponent) -> GpuTensorHandle:
        """lib.rs:2325-2331: mantissa .* 2.^exponent, operands of one shape."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_pow2_scale(self._ctx, self._id(mantissa), self._id(exponent), C.byref(out)))
        return self._handle(out.value)

    def round_digits(self, a, digits: int, significant: bool = False) -> GpuTensorHandle:
        """lib.rs:2197-2204 (decimals; `significant` is refused -> the host path)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_round_digits(self._ctx, self._id(a), int(digits), 1 if significant else 0, C.byref(out)))
        return self._handle(out.value)

    def _real_part(self, part: int, a) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_real_part(self._ctx, part, self._id(a), C.byref(out)))
        return self._handle(out.value)

    def unary_real(self, a): return self._real_part(0, a)   # lib.rs:2229
    def unary_imag(self, a): return self._real_part(1, a)   # lib.rs:2223
    def unary_conj(self, a): return self._real_part(2, a)   # lib.rs:2235
    def unary_angle(self, a): return self._real_part(3, a)  # lib.rs:2217

    def logical_isreal(self, a) -> bool:
        """lib.rs:2055-2057."""
        res = C.c_int()
        self._check(self._lib.rmhip_isreal(self._ctx, self._id(a), C.byref(res)))
        return bool(res.value)

    # -- small construction / linear-algebra hooks (misc_ops.hip) -----------------------------------
    def diag_from_vector(self, vector, offset: int = 0) -> GpuTensorHandle:
        """lib.rs:1600-1608: the square matrix of size len + |offset| with the vector on diagonal `offset`."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_diag_from_vector(self._ctx, self._id(vector), int(offset), -1, -1, C.byref(out)))
        return self._handle(out.value)

    def diag_from_vector_sized(self, vector, offset: int, rows: int, cols: int) -> GpuTensorHandle:
        """lib.rs:1613-1623: the same into an explicit rows x cols (elements outside are dropped)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_diag_from_vector(self._ctx, self._id(vector), int(offset), int(rows), int(cols), C.byref(out)))
        return self._handle(out.value)

    def kron(self, a, b) -> GpuTensorHandle:
        """lib.rs:2697-2699; kron.rs:358-485."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_kron(self._ctx, self._id(a), self._id(b), C.byref(out)))
        return self._handle(out.value)

    def cross(self, lhs, rhs, dim: Optional[int] = None) -> GpuTensorHandle:
        """lib.rs:2701-2708: `dim` is ONE-based (`Option<usize>` as the builtin parsed it), None = the first dimension of extent 3."""
        if dim is not None and int(dim) < 1:
            raise RmhipError(1, "cross: dimension must be >= 1")
        out = C.c_uint64()
        self._check(self._lib.rmhip_cross(self._ctx, self._id(lhs), self._id(rhs), 0 if dim is None else int(dim), C.byref(out)))
        return self._handle(out.value)

    def gradient_dim(self, a, dim: int, spacing: float = 1.0) -> GpuTensorHandle:
        """lib.rs:2604-2611: zero-based dim, scalar spacing (gradient.rs:650-720)."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_gradient_dim(self._ctx, self._id(a), int(dim), float(spacing), 0, C.byref(out)))
        return self._handle(out.value)

    def gradient_dim_with_coordinates(self, a, dim: int, coordinates) -> GpuTensorHandle:
        """lib.rs:2612-2620: denominators from a resident coordinate vector of the dimension's extent."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_gradient_dim(self._ctx, self._id(a), int(dim), 1.0, self._id(coordinates), C.byref(out)))
        return self._handle(out.value)

    def _trapz(self, a, dim: int, spacing, cumulative: bool) -> GpuTensorHandle:
        kind, scalar, sid = 0, 0.0, 0
        if spacing is None:
            pass
        elif isinstance(spacing, (int, float)):
            kind, scalar = 1, float(spacing)
        else:
            name, h = spacing
            kind = {"scalar_handle": 2, "vector": 3, "tensor": 4}[name]
            sid = self._id(h)
        out = C.c_uint64()
        self._check(self._lib.rmhip_trapz_dim(self._ctx, self._id(a), int(dim), 1 if cumulative else 0, kind, scalar, sid, C.byref(out)))
        return self._handle(out.value)

    def trapz_dim(self, a, dim: int, spacing=None) -> GpuTensorHandle:
        """lib.rs:2893-2900; `ProviderTrapezoidSpacing` (:1060-1066) as None (Unit), a float (Scalar) or ("scalar_handle" | "vector" |
        "tensor", handle)."""
        return self._trapz(a, dim, spacing, False)

    def cumtrapz_dim(self, a, dim: int, spacing=None) -> GpuTensorHandle:
        """lib.rs:2901-2908."""
        return self._trapz(a, dim, spacing, True)

    def norm(self, tensor, order="two", p: float = 2.0) -> GpuTensorHandle:
        """lib.rs:2451-2457; `ProviderNormOrder` (:745-754) as "one" | "two" | "inf" | "-inf" | "zero" | "fro" | "nuc" | "p" (with `p`) -> [1, 1]."""
        codes = {"one": 1, "two": 2, "inf": 3, "-inf": 4, "zero": 5, "fro": 6, "nuc": 7, "p": 8}
        if order not in codes:
            raise RmhipError(1, f"norm: order {order!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_norm(self._ctx, self._id(tensor), codes[order], float(p), C.byref(out)))
        return self._handle(out.value)

    def issymmetric(self, matrix, kind: str = "symmetric", tolerance: float = 0.0) -> bool:
        """lib.rs:3115-3124 (`ProviderSymmetryKind::{Symmetric, Skew}`): decided on the device, only the bool comes back."""
        if kind not in ("symmetric", "skew"):
            raise RmhipError(1, f"issymmetric: kind {kind!r}")
        res = C.c_int()
        self._check(self._lib.rmhip_issymmetric(self._ctx, self._id(matrix), 1 if kind == "skew" else 0, float(tolerance), C.byref(res)))
        return bool(res.value)

    def unique(self, handle, rows: bool = False, order: str = "sorted", occurrence: str = "first"):
        """lib.rs:2645-2651 -> `UniqueResult { values, ia, ic }` as host arrays ([count, 1], [count, 1], [numel, 1])."""
        if rows:
            raise ProviderError(_lib.ERR_UNSUPPORTED, "unique: the 'rows' form is not served")
        if order not in ("sorted", "stable") or occurrence not in ("first", "last"):
            raise RmhipError(1, f"unique: order {order!r} / occurrence {occurrence!r}")
        n = int(np.prod(handle.shape, dtype=np.int64)) if len(handle.shape) else 1
        values, ia, ic = np.empty(max(n, 1)), np.empty(max(n, 1)), np.empty(max(n, 1))
        count = C.c_size_t()
        ptr = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self._lib.rmhip_unique(self._ctx, self._id(handle), 1 if order == "stable" else 0, 1 if occurrence == "last" else 0, C.byref(count),
                                           ptr(values), ptr(ia), ptr(ic)))
        g = count.value
        return values[:g].reshape(g, 1).copy(), ia[:g].reshape(g, 1).copy(), ic[:n].reshape(n, 1).copy()

    def union(self, a, b, rows: bool = False, order: str = "sorted"):
        """lib.rs:2652-2659 -> `UnionResult { values, ia, ib }` as host arrays ([g, 1] each)."""
        if rows:
            raise ProviderError(_lib.ERR_UNSUPPORTED, "union: the 'rows' form is not served")
        if order not in ("sorted", "stable"):
            raise RmhipError(1, f"union: order {order!r}")
        na, nb = (int(np.prod(h.shape, dtype=np.int64)) if len(h.shape) else 1 for h in (a, b))
        values, ia, ib = np.empty(max(na + nb, 1)), np.empty(max(na, 1)), np.empty(max(nb, 1))
        cnt, ca, cb = C.c_size_t(), C.c_size_t(), C.c_size_t()
        ptr = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self._lib.rmhip_union(self._ctx, self._id(a), self._id(b), 1 if order == "stable" else 0, C.byref(cnt), ptr(values), C.byref(ca), ptr(ia),
                                          C.byref(cb), ptr(ib)))
        return values[:cnt.value].reshape(-1, 1).copy(), ia[:ca.value].reshape(-1, 1).copy(), ib[:cb.value].reshape(-1, 1).copy()

    def setdiff(self, a, b, rows: bool = False, order: str = "sorted"):
        """lib.rs:2660-2667 -> `SetdiffResult { values, ia }` as host arrays."""
        if rows:
            raise ProviderError(_lib.ERR_UNSUPPORTED, "setdiff: the 'rows' form is not served")
        if order not in ("sorted", "stable"):
            raise RmhipError(1, f"setdiff: order {order!r}")
        na = int(np.prod(a.shape, dtype=np.int64)) if len(a.shape) else 1
        values, ia = np.empty(max(na, 1)), np.empty(max(na, 1))
        cnt = C.c_size_t()
        ptr = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self._lib.rmhip_setdiff(self._ctx, self._id(a), self._id(b), 1 if order == "stable" else 0, C.byref(cnt), ptr(values), ptr(ia)))
        return values[:cnt.value].reshape(-1, 1).copy(), ia[:cnt.value].reshape(-1, 1).copy()

    def ismember(self, a, b, rows: bool = False):
        """lib.rs `ismember` -> `IsMemberResult { mask, loc }` as host arrays in a's shape (mask: uint8)."""
        if rows:
            raise ProviderError(_lib.ERR_UNSUPPORTED, "ismember: the 'rows' form is not served")
        n = int(np.prod(a.shape, dtype=np.int64)) if len(a.shape) else 1
        mask, loc = np.zeros(max(n, 1), dtype=np.uint8), np.zeros(max(n, 1))
        self._check(self._lib.rmhip_ismember(self._ctx, self._id(a), self._id(b), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), loc.ctypes.data_as(C.POINTER(C.c_double))))
        return mask[:n].reshape(a.shape, order="F").copy(), loc[:n].reshape(a.shape, order="F").copy()

    def iir_filter(self, b, a, x, dim: int, zi=None, unit_denominator: bool = False):
        """lib.rs:2551-2559 -> `ProviderIirFilterResult { output, final_state }` as a pair of handles."""
        out, fin = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_iir_filter(self._ctx, self._id(b), self._id(a), self._id(x), int(dim), self._id(zi) if zi is not None else 0,
                                               1 if unit_denominator else 0, C.byref(out), C.byref(fin)))
        return self._handle(out.value), self._handle(fin.value)

    def imfilter(self, image, kernel, padding="constant", shape: str = "same", mode: str = "correlation") -> GpuTensorHandle:
        """lib.rs:1809-1817 (`ImfilterOptions`, :1193-1222); padding: a number (constant fill) | "replicate" | "symmetric" | "circular" ("constant": 0)."""
        pads = {"constant": (0, 0.0), "replicate": (1, 0.0), "symmetric": (2, 0.0), "circular": (3, 0.0)}
        pad, fill = pads[padding] if isinstance(padding, str) else (0, float(padding))
        shapes, modes = {"same": 0, "full": 1, "valid": 2}, {"correlation": 0, "convolution": 1}
        if shape not in shapes or mode not in modes:
            raise RmhipError(1, f"imfilter: shape {shape!r} / mode {mode!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_imfilter(self._ctx, self._id(image), self._id(kernel), pad, fill, shapes[shape], modes[mode], C.byref(out)))
        return self._handle(out.value)

    def interp1(self, x, y, xq, sample_len: int, series_count: int, query_len: int, output_shape, method: str = "linear", extrapolation="nan") -> GpuTensorHandle:
        """lib.rs:2458-2463 (`ProviderInterp1Request`, :769-783); extrapolation: "nan" | "extrapolate" | a fill value."""
        if method not in ("linear", "nearest"):
            raise RmhipError(1, f"interp1: method {method!r}")
        mode, fill = (0, 0.0) if extrapolation == "nan" else (1, 0.0) if extrapolation == "extrapolate" else (2, float(extrapolation))
        sh, rank = _shape_array(output_shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_interp1(self._ctx, self._id(x), self._id(y), self._id(xq), int(sample_len), int(series_count), int(query_len), sh, rank,
                                            1 if method == "nearest" else 0, mode, fill, C.byref(out)))
        return self._handle(out.value)

    def polyval(self, coefficients, points, mu: Optional[Tuple[float, float]] = None) -> GpuTensorHandle:
        """lib.rs:1652-1660 (`ProviderPolyvalOptions { mu: Option<{mean, scale}> }`, :705-713)."""
        out = C.c_uint64()
        mean, scale = (float(mu[0]), float(mu[1])) if mu is not None else (0.0, 1.0)
        self._check(self._lib.rmhip_polyval(self._ctx, self._id(coefficients), self._id(points), 1 if mu is not None else 0, mean, scale, C.byref(out)))
        return self._handle(out.value)

    def _polyder(self, p, q, quotient: bool):
        out, den = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rmhip_polyder(self._ctx, self._id(p), self._id(q) if q is not None else 0, 1 if quotient else 0, C.byref(out),
                                            C.byref(den) if quotient else None))
        return (self._handle(out.value), self._handle(den.value)) if quotient else self._handle(out.value)

    def polyder_single(self, polynomial) -> GpuTensorHandle:
        """lib.rs:1674-1679: the derivative's coefficients, trimmed of leading zeros, in the input's orientation."""
        return self._polyder(polynomial, None, False)

    def polyder_product(self, p, q) -> GpuTensorHandle:
        """lib.rs:1682-1688: (p q)' = p' q + p q'."""
        return self._polyder(p, q, False)

    def polyder_quotient(self, u, v) -> Tuple[GpuTensorHandle, GpuTensorHandle]:
        """lib.rs:1691-1701 (`ProviderPolyderQuotient { numerator, denominator }`): u' v - u v' and v v."""
        return self._polyder(u, v, True)

    def polyint(self, polynomial, constant: float = 0.0) -> GpuTensorHandle:
        """lib.rs:1704-1710."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_polyint(self._ctx, self._id(polynomial), float(constant), C.byref(out)))
        return self._handle(out.value)

    def meshgrid(self, axes: Sequence[Sequence[float]]) -> List[GpuTensorHandle]:
        """lib.rs:1561-1564: two or three HOST axes (`MeshgridAxisView`) -> `ProviderMeshgridResult.outputs` (X, Y[, Z])."""
        if len(axes) not in (2, 3):
            raise RmhipError(1, "meshgrid: provider expects two or three axes")
        arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel()) for a in axes]
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        outs = (C.c_uint64 * 3)()
        z = arrs[2] if len(arrs) == 3 else None
        self._check(self._lib.rmhip_meshgrid(self._ctx, ptr(arrs[0]), arrs[0].size, ptr(arrs[1]), arrs[1].size, ptr(z) if z is not None else None,
                                             z.size if z is not None else 0, outs))
        return [self._handle(outs[i]) for i in range(len(arrs))]

    def zeros_with_storage(self, shape: Sequence[int], storage: str = "real") -> GpuTensorHandle:
        """lib.rs:1472-1489: `GpuTensorStorage::{Real, ComplexInterleaved}` as "real" | "complex"."""
        if storage == "real":
            return self.zeros(shape)
        if storage != "complex":
            raise RmhipError(1, f"zeros_with_storage: storage {storage!r}")
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_zeros_complex(self._ctx, sh, rank, C.byref(out)))
        return self._handle(out.value)

    def conv1d(self, signal, kernel, mode: str = "full", orientation: str = "row") -> GpuTensorHandle:
        """lib.rs:2535-2542 (`ProviderConv1dOptions { mode, orientation }`, :1277-1293)."""
        modes = {"full": 0, "same": 1, "valid": 2}
        if mode not in modes or orientation not in ("row", "column"):
            raise RmhipError(1, f"conv1d: mode {mode!r} / orientation {orientation!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_conv1d(self._ctx, self._id(signal), self._id(kernel), modes[mode], 1 if orientation == "column" else 0, C.byref(out)))
        return self._handle(out.value)

    def conv2d(self, signal, kernel, mode: str = "full") -> GpuTensorHandle:
        """lib.rs:2543-2550."""
        modes = {"full": 0, "same": 1, "valid": 2}
        if mode not in modes:
            raise RmhipError(1, f"conv2d: mode {mode!r}")
        out = C.c_uint64()
        self._check(self._lib.rmhip_conv2d(self._ctx, self._id(signal), self._id(kernel), modes[mode], C.byref(out)))
        return self._handle(out.value)

    def moving_window(self, input, output_shape, dim: int, before: int, after: int, op: str, endpoints="shrink", nan_mode: str = "include",
                      normalization: str = "sample") -> GpuTensorHandle:
        """lib.rs:2852-2857 (`ProviderMovingWindowRequest`, :990-1003); endpoints: "shrink" | "discard" | a fill value."""
        ops = {"sum": 0, "mean": 1, "prod": 2, "min": 3, "max": 4, "median": 5, "std": 6, "var": 7}
        if op not in ops or nan_mode not in ("include", "omit") or normalization not in ("sample", "population"):
            raise RmhipError(1, f"moving_window: op {op!r} / nan_mode {nan_mode!r} / normalization {normalization!r}")
        ep, fill = (0, 0.0) if endpoints == "shrink" else (1, 0.0) if endpoints == "discard" else (2, float(endpoints))
        sh, rank = _shape_array(output_shape)
        out = C.c_uint64()
        self._check(self._lib.rmhip_moving_window(self._ctx, self._id(input), int(dim), int(before), int(after), ops[op], ep, fill, 1 if nan_mode == "omit" else 0,
                                                  1 if normalization == "population" else 0, sh, rank, C.byref(out)))
        return self._handle(out.value)

    def _window(self, kind: int, length: int, periodic: bool) -> GpuTensorHandle:
        out = C.c_uint64()
        self._check(self._lib.rmhip_window(self._ctx, kind, int(length), 1 if periodic else 0, C.byref(out)))
        return self._handle(out.value)

    def hann_window(self, length: int, periodic: bool = False): return self._window(0, length, periodic)      # lib.rs:1797
    def hamming_window(self, length: int, periodic: bool = False): return self._window(1, length, periodic)   # lib.rs:1801
    def blackman_window(self, length: int, periodic: bool = False): return self._window(2, length, periodic)  # lib.rs:1805

    def fft_dim(self, handle, length: Optional[int], dim: int) -> GpuTensorHandle:
        """lib.rs:2622-2630: the transform along zero-based `dim`, padded / truncated to `length` (None: the extent) -> complex tensor."""
        return self._fft(handle, length, dim, 0)

    def ifft_dim(self, handle, length: Optional[int], dim: int) -> GpuTensorHandle:
        """lib.rs:2631-2638: the inverse transform (scaled by 1 / length)."""
        return self._fft(handle, length, dim, 1)

    def _fft(self, handle, length, dim, inverse):
        out = C.c_uint64()
        self._check(self._lib.rmhip_fft_dim(self._ctx, self._id(handle), -1 if length is None else int(length), int(dim), inverse, C.byref(out)))
        return self._handle(out.value)

    def signal_hilbert(self, input, length: Optional[int], dim: int) -> GpuTensorHandle:
        """lib.rs:2572-2577 (`ProviderHilbertRequest`): the analytic signal along zero-based `dim` -> complex tensor."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_hilbert(self._ctx, self._id(input), -1 if length is None else int(length), int(dim), C.byref(out)))
        return self._handle(out.value)

    def fft_extract_real(self, handle) -> GpuTensorHandle:
        """lib.rs:2639-2644: the real parts of a complex tensor as a real tensor."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_complex_real(self._ctx, self._id(handle), C.byref(out)))
        return self._handle(out.value)

    def complex_from_real(self, real) -> GpuTensorHandle:
        """lib.rs:1940-1947: complex-interleaved storage with a zero imaginary lane."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_complex(self._ctx, self._id(real), 0, C.byref(out)))
        return self._handle(out.value)

    def complex_from_real_imag(self, real, imag) -> GpuTensorHandle:
        """lib.rs:1949-1959: `complex(real, imag)`, a one-element operand expands."""
        out = C.c_uint64()
        self._check(self._lib.rmhip_complex(self._ctx, self._id(real), self._id(imag), C.byref(out)))
        return self._handle(out.value)

    def ishermitian(self, matrix, kind: str = "hermitian", tolerance: float = 0.0) -> bool:
        """lib.rs:3126-3138 (`ProviderHermitianKind::{Hermitian, Skew}`), real data: issymmetric's test plus "a NaN diagonal fails"."""
        if kind not in ("hermitian", "skew"):
            raise RmhipError(1, f"ishermitian: kind {kind!r}")
        res = C.c_int()
        self._check(self._lib.rmhip_ishermitian(self._ctx, self._id(matrix), 1 if kind == "skew" else 0, float(tolerance), C.byref(res)))
        return bool(res.value)

    def bandwidth(self, matrix) -> Tuple[int, int]:
        """lib.rs:3140-3143 -> `ProviderBandwidth { lower, upper }` as a pair; only the two counts leave the device."""
        lo, up = C.c_uint(), C.c_uint()
        self._check(self._lib.rmhip_bandwidth(self._ctx, self._id(matrix), C.byref(lo), C.byref(up)))
        return int(lo.value), int(up.value)

    def random_uniform_like(self, prototype: GpuTensorHandle) -> GpuTensorHandle:
        """lib.rs:1718-1720: `random_uniform(&prototype.shape)`."""
        return self.random_uniform(prototype.shape)

    def random_normal_like(self, prototype: GpuTensorHandle) -> GpuTensorHandle:
        """lib.rs:1728-1730."""
        return self.random_normal(prototype.shape)

    def _random_dist(self, fn, args, shape) -> GpuTensorHandle:
        sh, rank = _shape_array(shape)
        out = C.c_uint64()
        self._check(fn(self._ctx, *args, sh, rank, C.byref(out)))
        return self._handle(out.value, shape)

    def random_unifrnd(self, a: float, b: float, shape: Sequence[int]) -> GpuTensorHandle:
        """lib.rs:1750-1757: a + (b - a) * u per element (random.rs:514-528)."""
        return self._random_dist(self._lib.rmhip_random_unifrnd, (float(a), float(b)), shape)

    def random_exponential(self, mu: float, shape: Sequence[int]) -> GpuTensorHandle:
        """lib.rs:1733-1737: -mu * ln(max(u, MIN_POSITIVE)) (random.rs:290-300)."""
        return self._random_dist(self._lib.rmhip_random_exponential, (float(mu),), shape)

    def random_normrnd(self, mu: float, sigma: float, shape: Sequence[int]) -> GpuTensorHandle:
        """lib.rs:1740-1747: mu + sigma * z over Box-Muller pairs (random.rs:302-320)."""
        return self._random_dist(self._lib.rmhip_random_normrnd, (float(mu), float(sigma)), shape)

    def random_integer_range(self, lower: int, upper: int, shape: Sequence[int]) -> GpuTensorHandle:
        """lib.rs:1820-1829: uniform integers in [lower, upper] (simple_provider.rs:3683-3725); an empty or > 2^53 range is an error."""
        if not (-2**63 <= int(lower) < 2**63 and -2**63 <= int(upper) < 2**63):
            raise RmhipError(1, "random_integer_range: bounds outside i64")
        return self._random_dist(self._lib.rmhip_random_integer_range, (int(lower), int(upper)), shape)

    def random_integer_like(self, prototype: GpuTensorHandle, lower: int, upper: int) -> GpuTensorHandle:
        """lib.rs:1832-1839."""
        return self.random_integer_range(lower, upper, prototype.shape)

    # -- telemetry / timing ---------------------------------------------------------------------
    def telemetry_snapshot(self) -> dict:
        """`telemetry_snapshot` (lib.rs:3023-3045): the counters of `ProviderTelemetry` (:1337-1357) plus its two lists,
        `solve_fallbacks` [(reason, count)] and `kernel_launches` [{kernel, precision, shape, tuning}] (oldest first)."""
        t = _lib.Telemetry()
        self._check(self._lib.rmhip_telemetry(self._ctx, C.byref(t)))
        snap = {f: int(getattr(t, f)) for f, _ in t._fields_}
        fallbacks, i = [], 0
        buf, cnt = C.create_string_buffer(96), C.c_uint64()
        while self._lib.rmhip_telemetry_solve_fallback(self._ctx, i, buf, 96, C.byref(cnt)) == _lib.OK:
            fallbacks.append((buf.value.decode(), int(cnt.value)))
            i += 1
        launches, i = [], 0
        rec = _lib.KernelLaunch()
        while self._lib.rmhip_telemetry_kernel_launch(self._ctx, i, C.byref(rec)) == _lib.OK:
            launches.append({"kernel": rec.kernel.decode(), "precision": rec.precision.decode(),
                             "shape": {rec.shape[k].key.decode(): int(rec.shape[k].value) for k in range(rec.n_shape)},
                             "tuning": {rec.tuning[k].key.decode(): int(rec.tuning[k].value) for k in range(rec.n_tuning)}})
            i += 1
        snap["solve_fallbacks"] = fallbacks
        snap["kernel_launches_log"] = launches
        return snap

    def lu_stats(self) -> dict:
        """`rmhip_lu_stats`: what a solve did on the device beyond `solve_fallbacks` - solve-path factorisations accepted,
        refactorisations after a multiplier exceeded tau, exchange / substitution time-outs, the last largest multiplier."""
        st = _lib.LuStats()
        self._check(self._lib.rmhip_lu_stats(self._ctx, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def reset_telemetry(self) -> None:
        self._check(self._lib.rmhip_reset_telemetry(self._ctx))

    def timer_begin(self) -> None:
        self._check(self._lib.rmhip_timer_begin(self._ctx))

    def timer_end(self) -> float:
        ms = C.c_double()
        self._check(self._lib.rmhip_timer_end(self._ctx, C.byref(ms)))
        return float(ms.value)


def wgsl_translate(shader: str, kind: str = "elementwise") -> str:
    """Front-end only (no GPU): the HIP source librmhip would compile for `shader`."""
    lib = _lib.load()
    needed = C.c_size_t()
    k = 0 if kind == "elementwise" else 1
    rc = lib.rmhip_wgsl_translate(shader.encode(), k, None, 0, C.byref(needed))
    if rc != _lib.OK:
        raise ProviderError(rc, _lib.last_error())
    buf = C.create_string_buffer(needed.value)
    rc = lib.rmhip_wgsl_translate(shader.encode(), k, buf, needed.value, C.byref(needed))
    if rc != _lib.OK:
        raise ProviderError(rc, _lib.last_error())
    return buf.value.decode()


def wgsl_compile_check(shader: str, kind: str = "elementwise") -> None:
    """Front-end + hipRTC compile for gfx950 (no GPU needed). Raises ProviderError on failure."""
    lib = _lib.load()
    rc = lib.rmhip_wgsl_compile_check(shader.encode(), 0 if kind == "elementwise" else 1)
    if rc != _lib.OK:
        raise ProviderError(rc, _lib.last_error())


def _attach_unary_hooks() -> None:
    for name, op in UNARY_HOOKS.items():
        def hook(self, a, _op=op):
            return self._unary(_op, a)
        hook.__name__ = name
        hook.__doc__ = f"`{name}` through rmhip_unary(RMHIP_{op.upper()})"
        setattr(HipProvider, name, hook)


_attach_unary_hooks()
