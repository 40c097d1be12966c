"""ctypes binding of librmhip.so (include/rmhip.h).

There is deliberately no fallback: if the shared library is missing or fails to load this module
raises, and if no gfx950 device is present `rmhip_init` fails (RMHIP_ERR_NO_DEVICE).  The product
path never imports anything under oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "librmhip.so"

OK = 0
COMM_ID_BYTES = 128
COMM_RCCL, COMM_HOST_SHM = 0, 1
ERR_INVALID, ERR_UNSUPPORTED, ERR_SHAPE, ERR_HIP, ERR_NOT_FOUND = 1, 2, 3, 4, 5
ERR_COMPILE, ERR_SINGULAR, ERR_OOM, ERR_NO_DEVICE = 6, 7, 8, 9


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 32),
        ("device_ordinal", C.c_int),
        ("compute_units", C.c_int),
        ("wavefront_size", C.c_int),
        ("clock_mhz", C.c_int),
        ("total_memory_bytes", C.c_uint64),
        ("precision_bits", C.c_int),
        ("reduction_workgroup_size", C.c_uint32),
        ("two_pass_threshold", C.c_uint32),
        ("vendor", C.c_char * 32),
        ("backend", C.c_char * 32),
        ("xcd_count", C.c_int),
    ]


class Telemetry(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "fused_elementwise_count", "fused_elementwise_ns", "fused_reduction_count", "fused_reduction_ns",
        "matmul_count", "matmul_ns", "mldivide_count", "mldivide_ns", "upload_bytes", "download_bytes",
        "fusion_cache_hits", "fusion_cache_misses", "kernel_launches", "bytes_allocated", "bytes_pooled",
        "linsolve_count", "linsolve_ns", "mrdivide_count", "mrdivide_ns")]


class LuStats(C.Structure):
    """rmhip_lu_stats_t"""
    _fields_ = [("solve_path_factorizations", C.c_uint64), ("pivot_growth_fallbacks", C.c_uint64),
                ("panel_exchange_timeouts", C.c_uint64), ("subst_chain_timeouts", C.c_uint64),
                ("last_max_multiplier", C.c_double), ("tau", C.c_double), ("one_xcd_panels", C.c_int),
                ("conservative_panels", C.c_int), ("svd_solves", C.c_uint64)]


class KernelAttr(C.Structure):
    """rmhip_kernel_attr_t (KernelAttrTelemetry, lib.rs:1366-1370)"""
    _fields_ = [("key", C.c_char * 16), ("value", C.c_uint64)]


class KernelLaunch(C.Structure):
    """rmhip_kernel_launch_t (KernelLaunchTelemetry, lib.rs:1372-1378)"""
    _fields_ = [("kernel", C.c_char * 48), ("precision", C.c_char * 8), ("n_shape", C.c_uint32), ("n_tuning", C.c_uint32),
                ("shape", KernelAttr * 6), ("tuning", KernelAttr * 6)]


class MatmulEpilogue(C.Structure):
    """rmhip_matmul_epilogue_t (MatmulEpilogue, lib.rs:3498-3560)"""
    _fields_ = [("alpha", C.c_double), ("beta", C.c_double), ("row_scale", C.c_uint64), ("col_scale", C.c_uint64),
                ("row_op", C.c_int), ("col_op", C.c_int), ("has_clamp_min", C.c_int), ("has_clamp_max", C.c_int),
                ("has_pow", C.c_int), ("clamp_min", C.c_double), ("clamp_max", C.c_double), ("pow_exponent", C.c_double),
                ("diag_output", C.c_uint64)]


class LinsolveOptions(C.Structure):
    """rmhip_linsolve_options_t (ProviderLinsolveOptions, lib.rs:679-690)"""
    _fields_ = [(n, C.c_int) for n in ("lower", "upper", "rectangular", "transposed", "conjugate", "symmetric", "posdef",
                                       "need_rcond", "has_rcond")] + [("rcond", C.c_double)]


class ImageNormalize(C.Structure):
    """rmhip_image_normalize_t (ImageNormalizeDescriptor, lib.rs:3563-3577)"""
    _fields_ = [("batch", C.c_size_t), ("height", C.c_size_t), ("width", C.c_size_t), ("epsilon", C.c_double),
                ("has_gain", C.c_int), ("has_bias", C.c_int), ("has_gamma", C.c_int), ("clamp_zero", C.c_int),
                ("gain", C.c_double), ("bias", C.c_double), ("gamma", C.c_double)]


class View(C.Structure):
    """rmhip_view_t: rows [row_off, row_off+rows) x cols [col_off, col_off+cols) of a 2-D buffer."""
    _fields_ = [("buf", C.c_uint64), ("row_off", C.c_size_t), ("col_off", C.c_size_t), ("rows", C.c_size_t),
                ("cols", C.c_size_t)]


# Every symbol include/rmhip.h declares: name -> (restype, argtypes). Used both to bind and by the
# CPU-side test that checks the library exports the full ABI.
_P = C.c_void_p
_SZ = C.c_size_t
_SZP = C.POINTER(C.c_size_t)
_DP = C.POINTER(C.c_double)
_BUF = C.c_uint64
_BUFP = C.POINTER(C.c_uint64)
SIGNATURES = {
    "rmhip_version": (C.c_char_p, []),
    "rmhip_last_error": (C.c_char_p, []),
    "rmhip_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rmhip_shutdown": (C.c_int, [_P]),
    "rmhip_device_info": (C.c_int, [_P, C.POINTER(DeviceInfo)]),
    "rmhip_set_precision": (C.c_int, [_P, C.c_int]),
    "rmhip_buffer_bits": (C.c_int, [_P, _BUF, C.POINTER(C.c_int)]),
    "rmhip_set_stream": (C.c_int, [_P, _P]),
    "rmhip_get_stream": (_P, [_P]),
    "rmhip_synchronize": (C.c_int, [_P]),
    "rmhip_upload": (C.c_int, [_P, _DP, _SZP, _SZ, _BUFP]),
    "rmhip_download": (C.c_int, [_P, _BUF, _DP, _SZ]),
    "rmhip_free": (C.c_int, [_P, _BUF]),
    "rmhip_shape": (C.c_int, [_P, _BUF, _SZP, _SZP]),
    "rmhip_numel": (C.c_int, [_P, _BUF, _SZP]),
    "rmhip_fill": (C.c_int, [_P, C.c_double, _SZP, _SZ, _BUFP]),
    "rmhip_reshape": (C.c_int, [_P, _BUF, _SZP, _SZ, _BUFP]),
    "rmhip_wrap_external": (C.c_int, [_P, _P, _SZP, _SZ, _BUFP]),
    "rmhip_device_ptr": (_P, [_P, _BUF]),
    "rmhip_fill_uniform": (C.c_int, [_P, C.c_uint64, C.c_double, C.c_double, _SZP, _SZ, _BUFP]),
    "rmhip_fused_elementwise": (C.c_int, [_P, C.c_char_p, _BUFP, _SZ, _SZP, _SZ, _SZ, _SZ, _BUFP]),
    "rmhip_fused_reduction": (C.c_int, [_P, C.c_char_p, _BUFP, _SZ, _SZP, _SZ, _SZ, _SZ, C.c_uint32, C.c_int,
                                        C.c_double, _BUFP]),
    "rmhip_wgsl_translate": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, _SZ, _SZP]),
    "rmhip_wgsl_compile_check": (C.c_int, [C.c_char_p, C.c_int]),
    "rmhip_binary": (C.c_int, [_P, C.c_int, _BUF, _BUF, _BUFP]),
    "rmhip_unary": (C.c_int, [_P, C.c_int, _BUF, _BUFP]),
    "rmhip_scalar": (C.c_int, [_P, C.c_int, _BUF, C.c_double, _BUFP]),
    "rmhip_reduce": (C.c_int, [_P, C.c_int, _BUF, C.c_int, C.c_int, _BUFP]),
    "rmhip_reduce_minmax_dim": (C.c_int, [_P, C.c_int, _BUF, C.c_int, C.c_int, _BUFP, _BUFP]),
    "rmhip_reduce_std": (C.c_int, [_P, _BUF, C.c_int, C.c_int, C.c_int, _BUFP]),
    "rmhip_reduce_truth": (C.c_int, [_P, C.c_int, _BUF, C.c_int, C.c_int, _BUFP]),
    "rmhip_cumulative": (C.c_int, [_P, C.c_int, _BUF, C.c_int, C.c_int, C.c_int, _BUFP]),
    "rmhip_reduce_nd": (C.c_int, [_P, C.c_int, _BUF, _SZP, _SZ, C.c_int, _BUFP]),
    "rmhip_reduce_moments_nd": (C.c_int, [_P, _BUF, _SZP, _SZ, _BUFP, _BUFP]),
    "rmhip_dot": (C.c_int, [_P, _BUF, _BUF, C.c_int, _BUFP]),
    "rmhip_matmul": (C.c_int, [_P, _BUF, _BUF, _BUFP]),
    "rmhip_matmul_epilogue": (C.c_int, [_P, _BUF, _BUF, C.POINTER(MatmulEpilogue), _BUFP]),
    "rmhip_lu": (C.c_int, [_P, _BUF, _BUFP]),
    "rmhip_mldivide": (C.c_int, [_P, _BUF, _BUF, _BUFP]),
    "rmhip_mrdivide": (C.c_int, [_P, _BUF, _BUF, _BUFP]),
    "rmhip_linsolve": (C.c_int, [_P, _BUF, _BUF, C.POINTER(LinsolveOptions), _BUFP, _DP]),
    "rmhip_transpose": (C.c_int, [_P, _BUF, _BUFP]),
    "rmhip_syrk": (C.c_int, [_P, _BUF, _BUFP]),
    "rmhip_covariance": (C.c_int, [_P, _BUF, C.c_int, _BUFP]),
    "rmhip_diag_extract": (C.c_int, [_P, _BUF, C.c_longlong, _BUFP]),
    "rmhip_matmul_power_step": (C.c_int, [_P, _BUF, _BUF, C.c_double, _BUFP]),
    "rmhip_image_normalize": (C.c_int, [_P, _BUF, C.POINTER(ImageNormalize), _BUFP]),
    "rmhip_blk_copy": (C.c_int, [_P, C.POINTER(View), _BUFP]),
    "rmhip_blk_assign": (C.c_int, [_P, C.POINTER(View), _BUF]),
    "rmhip_blk_gemm": (C.c_int, [_P, C.c_double, C.POINTER(View), C.POINTER(View), C.c_double, C.POINTER(View)]),
    "rmhip_blk_trsm": (C.c_int, [_P, C.c_int, C.POINTER(View), C.POINTER(View)]),
    "rmhip_blk_lu": (C.c_int, [_P, C.POINTER(View), _BUFP, C.POINTER(C.c_int)]),
    "rmhip_blk_swap_rows": (C.c_int, [_P, C.POINTER(View), _BUF]),
    "rmhip_comm_unique_id": (C.c_int, [C.c_int, _P]),
    "rmhip_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rmhip_comm_destroy": (C.c_int, [_P]),
    "rmhip_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rmhip_comm_barrier": (C.c_int, [_P]),
    "rmhip_comm_bcast": (C.c_int, [_P, C.POINTER(View), C.c_int, C.c_int]),
    "rmhip_comm_wait": (C.c_int, [_P]),
    "rmhip_comm_allgather_f64": (C.c_int, [_P, _BUF, _BUFP]),
    "rmhip_comm_allgather_rows": (C.c_int, [_P, _BUF, _SZ, _SZ, _BUFP]),
    "rmhip_set_rng_state": (C.c_int, [_P, C.c_uint64]),
    "rmhip_get_rng_state": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "rmhip_rng_seed": (C.c_int, [_P, C.c_uint64]),
    "rmhip_random_uniform": (C.c_int, [_P, _SZP, _SZ, _BUFP]),
    "rmhip_random_normal": (C.c_int, [_P, _SZP, _SZ, _BUFP]),
    "rmhip_stochastic_evolution": (C.c_int, [_P, _BUF, C.c_double, C.c_double, C.c_uint32, _BUFP]),
    "rmhip_stochastic_evolution_sharded": (C.c_int, [_P, _BUF, C.c_double, C.c_double, C.c_uint32, C.c_uint64, _BUFP]),
    "rmhip_telemetry": (C.c_int, [_P, C.POINTER(Telemetry)]),
    "rmhip_reset_telemetry": (C.c_int, [_P]),
    "rmhip_telemetry_solve_fallback": (C.c_int, [_P, _SZ, C.c_char_p, _SZ, C.POINTER(C.c_uint64)]),
    "rmhip_telemetry_kernel_launch": (C.c_int, [_P, _SZ, C.POINTER(KernelLaunch)]),
    "rmhip_lu_stats": (C.c_int, [_P, C.POINTER(LuStats)]),
    "rmhip_timer_begin": (C.c_int, [_P]),
    "rmhip_timer_end": (C.c_int, [_P, _DP]),
}

_lib = None


def load() -> C.CDLL:
    """Load librmhip.so (once). Raises OSError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("RMHIP_LIBRARY", str(LIB_PATH)))
    if not path.exists():
        raise OSError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C runmat_amd/csrc`. runmat_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().rmhip_last_error().decode("utf-8", "replace")
