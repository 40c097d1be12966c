/*
 * rmhip.h -- C ABI of librmhip.so, the MI355X (gfx950) accelerate backend for RunMat's dense-array
 * hot path.  This is the drop-in boundary: every entry point below replaces one method (or a
 * small family of methods) of the reference's provider trait
 *   `trait AccelProvider` -- crates/runmat-accelerate-api/src/lib.rs:1386-3151
 * and is what a ~300-line `impl AccelProvider for HipProvider` (shim/hip_provider.rs,
 * INTEGRATION.md) binds through `extern "C"`.
 *
 * Conventions
 *   - Plain C types only: pointers, sizes, u64 buffer ids. No torch / C++ types.
 *   - Tensors are f64, column-major, described by (shape[], rank) like `HostTensorView`
 *     (lib.rs:3362-3372).  A buffer id is the `buffer_id` field of `GpuTensorHandle`
 *     (lib.rs:260-264); the provider owns the device memory until rmhip_free.
 *   - Every call returns 0 on success, else an RMHIP_ERR_* code; rmhip_last_error() returns a
 *     thread-local message.  Errors are soft: the reference's callers fall back to the CPU path
 *     on any provider Err (mtimes.rs:212-216, mldivide.rs:223-226, runner.rs:1140-1142), so
 *     unsupported requests return RMHIP_ERR_UNSUPPORTED and never abort.
 *   - Every op returns a NEW buffer (ResidencyPolicy::NewHandle); inputs are never mutated.
 *   - Work is enqueued on the context's HIP stream; rmhip_download / rmhip_synchronize block.
 *   - There is no CPU fallback inside the library: without a gfx950 device rmhip_init fails.
 *   - Every prototype carries a machine-readable `@serves` tag: the `AccelProvider` methods (lib.rs:1386-3151) it
 *     implements, `-` for entry points without a trait counterpart.  scripts/gen_bindings.py derives the ctypes table
 *     (runmat_amd/_abi.py) and the Rust `extern "C"` block (shim/rmhip_sys.rs) from this header, and
 *     tests/test_bindings.py checks that every served method exists in all host mirrors (Python, C++, Rust).
 */
#ifndef RMHIP_H
#define RMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMHIP_API __attribute__((visibility("default")))

typedef struct rmhip_ctx rmhip_ctx;
typedef uint64_t rmhip_buf;

enum rmhip_status {
    RMHIP_OK = 0,
    RMHIP_ERR_INVALID = 1,     /* bad argument / null pointer                                   */
    RMHIP_ERR_UNSUPPORTED = 2, /* request outside the supported subset -> caller falls back     */
    RMHIP_ERR_SHAPE = 3,       /* shape / dimension mismatch ("Inner matrix dimensions ...")    */
    RMHIP_ERR_HIP = 4,         /* HIP runtime error                                             */
    RMHIP_ERR_NOT_FOUND = 5,   /* unknown buffer id (simple_provider.rs:7717 "buffer not found")*/
    RMHIP_ERR_COMPILE = 6,     /* WGSL front-end or hipRTC failure                              */
    RMHIP_ERR_SINGULAR = 7,    /* LU pivot <= 1e-12: caller must use the CPU SVD path           */
    RMHIP_ERR_OOM = 8,
    RMHIP_ERR_NO_DEVICE = 9,
    RMHIP_ERR_GROWTH = 10      /* sharded solve: a multiplier outside the diagonal domains exceeded the bound (or a rank failed): use
                                  the block-column form, whose pivot rule is grid-wide                                            */
};

/* ---- library / context ---------------------------------------------------------------------- */

/* @serves - */
RMHIP_API const char* rmhip_version(void);
/* Thread-local message of the last failing call on this thread. Never NULL. */
/* @serves - */
RMHIP_API const char* rmhip_last_error(void);

/* Replaces provider construction + `register_provider` (lib.rs:3213-3225); one context per GPU
 * (one process per GPU in multi-GPU jobs).  `device_ordinal` is the HIP device index. */
/* @serves - */
RMHIP_API int rmhip_init(int device_ordinal, rmhip_ctx** out_ctx);
/* @serves - */
RMHIP_API int rmhip_shutdown(rmhip_ctx* ctx);

/* `device_info_struct` (lib.rs:1448-1456) + `precision` (:1458; 64 unless rmhip_set_precision chose 32). */
typedef struct rmhip_device_info {
    char name[128];
    char arch[32];          /* "gfx950" */
    int device_ordinal;
    int compute_units;
    int wavefront_size;
    int clock_mhz;
    uint64_t total_memory_bytes;
    int precision_bits;     /* 64 / 32: ProviderPrecision::F64 / F32 (lib.rs:815-818)          */
    uint32_t reduction_workgroup_size; /* default_reduction_workgroup_size (lib.rs:3048)       */
    uint32_t two_pass_threshold;       /* two_pass_threshold (lib.rs:3053)                     */
    char vendor[32];        /* ApiDeviceInfo::vendor (lib.rs:505-511): "AMD"                               */
    char backend[32];       /* ApiDeviceInfo::backend: "hip"                                                */
    int xcd_count;          /* accelerator dies the workgroup dispatcher interleaves over (8 on an MI355X in SPX mode, 1 in CPX),
                               probed at init; the LU's one-XCD placement and XCD-avoiding update kernels need exactly 8    */
    int memory_clock_khz;   /* hipDeviceProp_t::memoryClockRate as the driver reports it (MI355X: 2 000 000; HBM3E moves 4 bits per pin
                               and reported clock = 8 Gb/s per pin)                                                          */
    int memory_bus_width_bits; /* hipDeviceProp_t::memoryBusWidth (MI355X: 8192): bench.py derives the box's HBM and MFMA peaks from
                               these two, compute_units and clock_mhz instead of trusting constants                         */
    int l2_cache_bytes;     /* per-XCD L2 (4 MiB) */
} rmhip_device_info_t;
/* @serves device_info device_info_struct default_reduction_workgroup_size two_pass_threshold */
RMHIP_API int rmhip_device_info(rmhip_ctx* ctx, rmhip_device_info_t* out);

/* `ProviderPrecision` (lib.rs:815-818) is a property of the provider: 64 (default) or 32 bits, chosen before the
 * first buffer exists.  At 32 the host boundary is unchanged (`HostTensorView` is always f64, lib.rs:3362-3372):
 * upload rounds to f32, download widens; tensors live in HBM as f32 (half the traffic of every bandwidth-bound op) and
 * the planner sends its f32 shaders (`scalar_ty`, fusion.rs:1525).  Arithmetic stays f64 in registers and is rounded
 * once on store - what the CPU path does for `single` arrays (f64 storage pre-rounded through f32,
 * runmat-builtins/src/lib.rs:426-436) - so results agree with the CPU to f32 rounding instead of accumulating f32
 * error.  Fused elementwise / fused reduction, the per-op elementwise hooks, reductions, dot, rng, stochastic_evolution
 * and image_normalize read and write f32 storage directly; matmul runs on the f32 matrix cores (f32 accumulation, like
 * the reference's F32 backend; RMHIP_F32_MATMUL=f64 selects the f64-exact path); lu, mldivide/linsolve and the
 * remaining hooks run their f64 kernels on widened copies and narrow the result.  rmhip_buffer_bits reports a buffer's storage width (externally wrapped memory stays f64). */
/* @serves precision */
RMHIP_API int rmhip_set_precision(rmhip_ctx* ctx, int bits);
/* @serves - */
RMHIP_API int rmhip_buffer_bits(rmhip_ctx* ctx, rmhip_buf id, int* bits);

/* Stream plumbing: by default the context owns a non-blocking stream.  A host that already has
 * a stream (e.g. torch's current stream) can make the library enqueue there instead. */
/* @serves - */
RMHIP_API int rmhip_set_stream(rmhip_ctx* ctx, void* hip_stream);
/* @serves - */
RMHIP_API void* rmhip_get_stream(rmhip_ctx* ctx);
/* @serves - */
RMHIP_API int rmhip_synchronize(rmhip_ctx* ctx);

/* ---- memory: upload / download / free  (lib.rs:1387-1389) ----------------------------------- */

/* @serves upload */
RMHIP_API int rmhip_upload(rmhip_ctx* ctx, const double* host, const size_t* shape, size_t rank,
                           rmhip_buf* out);
/* Copies `n` doubles (must equal the buffer's element count) to `out_host`; blocks. */
/* @serves download */
RMHIP_API int rmhip_download(rmhip_ctx* ctx, rmhip_buf id, double* out_host, size_t n);
/* @serves free */
RMHIP_API int rmhip_free(rmhip_ctx* ctx, rmhip_buf id);
/* On entry *rank_inout is the capacity of shape_out; on exit the rank. */
/* @serves - */
RMHIP_API int rmhip_shape(rmhip_ctx* ctx, rmhip_buf id, size_t* rank_inout, size_t* shape_out);
/* @serves - */
RMHIP_API int rmhip_numel(rmhip_ctx* ctx, rmhip_buf id, size_t* out);
/* `zeros` / `ones` / `fill` (lib.rs:1468-1522). */
/* @serves zeros ones fill */
RMHIP_API int rmhip_fill(rmhip_ctx* ctx, double value, const size_t* shape, size_t rank,
                         rmhip_buf* out);
/* `reshape` (lib.rs:2676-2684): same numel, same buffer: the table entry's shape is updated in place and *out
 * receives `id` itself, as the trait default and the wgpu provider (ops/tensor.rs reshape_exec) do - callers
 * consume the source handle and never free it separately.  A transpose view is materialised first. */
/* @serves reshape */
RMHIP_API int rmhip_reshape(rmhip_ctx* ctx, rmhip_buf id, const size_t* shape, size_t rank,
                            rmhip_buf* out);
/* Zero-copy adoption of device memory owned by the host (torch tensor, RCCL receive buffer...).
 * The library never frees it; rmhip_free only drops the table entry. */
/* @serves - */
RMHIP_API int rmhip_wrap_external(rmhip_ctx* ctx, void* device_ptr, const size_t* shape,
                                  size_t rank, rmhip_buf* out);
/* Raw device pointer of a buffer (for RCCL collectives / torch views). NULL if unknown. */
/* @serves - */
RMHIP_API void* rmhip_device_ptr(rmhip_ctx* ctx, rmhip_buf id);
/* Deterministic device-side fill used by bench/tests: element i = lo + (hi-lo)*u53(splitmix64(
 * seed + (i+1)*0x9e3779b97f4a7c15)) -- identical to oracle's orc_fill_uniform. */
/* @serves - */
RMHIP_API int rmhip_fill_uniform(rmhip_ctx* ctx, uint64_t seed, double lo, double hi,
                                 const size_t* shape, size_t rank, rmhip_buf* out);

/* ---- fused kernels  (lib.rs:2946-3008) ------------------------------------------------------ */

/* `fused_elementwise` / `fused_elementwise_multi` (lib.rs:2946-2978).
 * `shader` is the WGSL text the reference planner emits (fusion.rs:1632-1763); the library parses
 * the `let tmpK: f64 = <expr>;` / `output[k].data[g] = <expr>;` body into an expression tape and
 * lowers it to a HIP kernel (hipRTC, cached by tape hash).  Inputs broadcast against `out_shape`
 * with front-padded shapes (elementwise.rs:1680-1697).  `n_out` == 1 writes `output`, > 1 writes
 * `output0..`.  `len` must equal prod(out_shape). */
/* @serves fused_elementwise fused_elementwise_multi */
RMHIP_API int rmhip_fused_elementwise(rmhip_ctx* ctx, const char* shader, const rmhip_buf* inputs,
                                      size_t n_in, const size_t* out_shape, size_t rank,
                                      size_t len, size_t n_out, rmhip_buf* out_ids);

enum rmhip_reduction_flavor { /* ReductionFlavor, lib.rs:865-890 */
    RMHIP_FLAVOR_SUM = 0,
    RMHIP_FLAVOR_MEAN = 1,          /* CPU semantics: sum / reduce_len (mean.rs:1134-1151)     */
    RMHIP_FLAVOR_CUSTOM_SCALE = 2   /* sum * scale                                             */
};
/* `fused_reduction` (lib.rs:2996-3008). `shader` is the text of fusion.rs:1765-2077; the library
 * reads from it: the folded `let val: f64 = <expr>;`, the axis (column-wise `(col * params.nrows)
 * + r` vs row-wise `row + (c * params.ncols)` addressing) and `const OMITNAN`.  Output has
 * `num_slices` elements with shape `out_shape`. `workgroup_size` is advisory (ignored). */
/* @serves fused_reduction */
RMHIP_API int rmhip_fused_reduction(rmhip_ctx* ctx, const char* shader, const rmhip_buf* inputs,
                                    size_t n_in, const size_t* out_shape, size_t rank,
                                    size_t reduce_len, size_t num_slices, uint32_t workgroup_size,
                                    int flavor, double custom_scale, rmhip_buf* out);

/* Front-end only (no GPU needed): translate a reference WGSL shader to the HIP source the library
 * would compile. `kind` 0 = elementwise, 1 = reduction, 0x100 | mask = elementwise whose inputs `mask` (bit k = input k)
 * are lazy random_normal operands (see rmhip_set_lazy_random). Writes a NUL-terminated string of at most
 * `cap` bytes to `out` and the required size to *needed. */
/* @serves - */
RMHIP_API int rmhip_wgsl_translate(const char* shader, int kind, char* out, size_t cap,
                                   size_t* needed);
/* Front-end + hipRTC compile for gfx950 (no GPU needed); 0 if the generated kernel builds. */
/* @serves - */
RMHIP_API int rmhip_wgsl_compile_check(const char* shader, int kind);

/* ---- per-op kernels  (lib.rs:1890-1938, 1979, 2069, 2077-2355) ------------------------------ */

enum rmhip_binary_op { /* elem_add/sub/mul/div/pow/max/min/hypot/atan2 */
    RMHIP_ADD = 0, RMHIP_SUB, RMHIP_MUL, RMHIP_DIV, RMHIP_POW, RMHIP_MAX, RMHIP_MIN, RMHIP_HYPOT,
    RMHIP_ATAN2, RMHIP_MOD, RMHIP_REM,
    /* elem_eq/ne/lt/le/gt/ge and logical_and/or/xor (lib.rs:1939-2068): 1.0 / 0.0 results, IEEE comparisons (NaN
     * compares false, != true), logical operands are "non-zero" tests (simple_provider.rs:4468-4760) */
    RMHIP_EQ, RMHIP_NE, RMHIP_LT, RMHIP_LE, RMHIP_GT, RMHIP_GE, RMHIP_AND, RMHIP_OR, RMHIP_XOR, RMHIP_BINARY_OP_COUNT
};
/* Operands broadcast under MATLAB implicit expansion (broadcast.rs:95-140). */
/* @serves elem_add elem_sub elem_mul elem_div elem_pow elem_max elem_min elem_hypot elem_atan2 elem_eq elem_ne elem_lt elem_le elem_gt elem_ge logical_and logical_or logical_xor */
RMHIP_API int rmhip_binary(rmhip_ctx* ctx, int op, rmhip_buf a, rmhip_buf b, rmhip_buf* out);

enum rmhip_unary_op { /* unary_* ; numbering shared with oracle/oracle.c */
    RMHIP_SIN = 0, RMHIP_COS, RMHIP_TAN, RMHIP_ASIN, RMHIP_ACOS, RMHIP_ATAN, RMHIP_SINH,
    RMHIP_COSH, RMHIP_TANH, RMHIP_ASINH, RMHIP_ACOSH, RMHIP_ATANH, RMHIP_EXP, RMHIP_EXPM1,
    RMHIP_LOG, RMHIP_LOG2, RMHIP_LOG10, RMHIP_LOG1P, RMHIP_SQRT, RMHIP_ABS, RMHIP_SIGN,
    RMHIP_FLOOR, RMHIP_CEIL, RMHIP_ROUND, RMHIP_FIX, RMHIP_NEG, RMHIP_EXP2, RMHIP_HEAVISIDE,
    RMHIP_ISNAN, RMHIP_ISINF, RMHIP_ISFINITE, RMHIP_UPLUS, RMHIP_SINGLE /* round through f32 */, RMHIP_DOUBLE,
    RMHIP_ERF, RMHIP_SINC, RMHIP_NOT /* logical_not: x == 0 */,
    /* special functions (lib.rs:2089-2118, 2319): the CPU builtins' own formulas - gamma / gammaln: Lanczos g = 7, 9 terms
     * (math/elementwise/gamma.rs:289-343, gammaln.rs:254-281); factorial: product table up to 170, NaN for non-integers
     * (factorial.rs:25-34, 272-314); nextpow2: ceil(log2(|x|)), 0 for 0 (nextpow2.rs:157-164); erfcinv: bracket +
     * 110 bisection steps on erfc (erfcinv.rs:261-308) */
    RMHIP_GAMMA, RMHIP_FACTORIAL, RMHIP_NEXTPOW2, RMHIP_GAMMALN, RMHIP_ERFCINV,
    /* `map_nan_to_zero` / `not_nan_mask` (lib.rs:2980-2988; the omitnan forms of sum / mean call them on resident
     * tensors, reduction/sum.rs:795, mean.rs:1077-1078): NaN -> 0, everything else unchanged / 1.0 where the value is not NaN,
     * else 0.0 (backend/wgpu/shaders/nan.rs: `select(v, 0, v != v)`, `select(0, 1, !(v != v))`) */
    RMHIP_NAN_TO_ZERO, RMHIP_NOT_NAN, RMHIP_UNARY_OP_COUNT
};
/* @serves unary_sin unary_cos unary_tan unary_asin unary_acos unary_atan unary_sinh unary_cosh unary_tanh unary_asinh unary_acosh unary_atanh unary_exp unary_expm1 unary_log unary_log2 unary_log10 unary_log1p unary_sqrt unary_abs unary_sign unary_floor unary_ceil unary_round unary_fix unary_pow2 unary_heaviside unary_single unary_double unary_erf unary_sinc unary_gamma unary_factorial unary_nextpow2 unary_gammaln unary_erfcinv logical_not logical_isnan logical_isinf logical_isfinite map_nan_to_zero not_nan_mask */
RMHIP_API int rmhip_unary(rmhip_ctx* ctx, int op, rmhip_buf a, rmhip_buf* out);

enum rmhip_scalar_op { /* scalar_add/sub/mul/div/rsub/rdiv/max/min (lib.rs:2333-2355) */
    RMHIP_SADD = 0, RMHIP_SSUB, RMHIP_SMUL, RMHIP_SDIV, RMHIP_SRSUB, RMHIP_SRDIV, RMHIP_SMAX,
    RMHIP_SMIN, RMHIP_SCALAR_OP_COUNT
};
/* @serves scalar_add scalar_sub scalar_mul scalar_div scalar_rsub scalar_rdiv scalar_max scalar_min */
RMHIP_API int rmhip_scalar(rmhip_ctx* ctx, int op, rmhip_buf a, double s, rmhip_buf* out);

/* ---- reductions  (lib.rs:2709-2721, 2756-2792, 2858-2883) ----------------------------------- */

enum rmhip_reduce_op { RMHIP_RSUM = 0, RMHIP_RMEAN, RMHIP_RMIN, RMHIP_RMAX, RMHIP_RPROD,
                       RMHIP_REDUCE_OP_COUNT };
/* dim < 0: reduce all elements -> shape [1,1] (simple_provider.rs:6728-6748).
 * dim 0 / 1 (zero-based, 2-D): -> [1,cols] / [rows,1] (simple_provider.rs:6750-6806).
 * nan_mode 0 = include (any NaN => NaN, sum.rs:1038-1045), 1 = omit. */
/* @serves reduce_sum reduce_sum_dim reduce_mean reduce_mean_dim reduce_min reduce_max reduce_prod reduce_prod_dim */
RMHIP_API int rmhip_reduce(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int nan_mode,
                           rmhip_buf* out);

/* `reduce_min_dim` / `reduce_max_dim` (lib.rs:2864-2883) -> `ReduceDimResult{values, indices}` (:513-517): minimum / maximum
 * along `dim` (zero-based) of an N-d tensor and WHERE it is - both outputs have the input's shape with extent 1 at `dim`; indices
 * are 1-based positions along `dim`, as f64.  op = RMHIP_RMIN / RMHIP_RMAX.  Semantics are the CPU builtin's
 * (runtime/builtins/math/reduction/min.rs:1443-1531, max.rs:1715-1727), which is what the runtime expects back (it calls the hook
 * in "includenan" mode only, min.rs:795-800): the FIRST occurrence wins ties, -0 is below +0, and with nan_mode 0 (include) the
 * first NaN of a slice fixes the result (value NaN, index of that NaN); nan_mode 1 (omit) skips NaNs, a slice of NaNs gives
 * (NaN, NaN).  Integer work: values and indices are bit-exact with the CPU's. */
/* @serves reduce_min_dim reduce_max_dim */
RMHIP_API int rmhip_reduce_minmax_dim(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int nan_mode, rmhip_buf* values,
                                      rmhip_buf* indices);
/* `reduce_std` / `reduce_std_dim` (lib.rs:2786-2802): standard deviation along `dim` (zero-based; dim < 0: all elements ->
 * [1,1]).  normalization 0 = sample (n - 1; 0 for a single value), 1 = population (`ProviderStdNormalization`, :957-960);
 * nan_mode as above (include: any NaN => NaN; omit: NaNs skipped, none left => NaN).  std.rs:858-935: Welford's update,
 * merged over chunks with Chan's formula. */
/* @serves reduce_std reduce_std_dim */
RMHIP_API int rmhip_reduce_std(rmhip_ctx* ctx, rmhip_buf a, int dim, int normalization, int nan_mode, rmhip_buf* out);
/* `reduce_nnz(_dim)` (lib.rs:2730-2742), `reduce_any(_dim)` / `reduce_all(_dim)` (:2803-2850): counts / truth values as f64.
 * nnz counts NaNs as non-zero (nnz.rs:358).  any: include => a NaN is true, omit_nan => NaNs are skipped (any.rs:722-733);
 * all: NaNs are skipped in both modes and a slice with nothing left is true (all.rs:671-703).  dim < 0: all elements. */
enum rmhip_truth_op { RMHIP_TNNZ = 0, RMHIP_TANY, RMHIP_TALL, RMHIP_TRUTH_OP_COUNT };
/* @serves reduce_nnz reduce_nnz_dim reduce_any reduce_any_dim reduce_all reduce_all_dim */
RMHIP_API int rmhip_reduce_truth(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int omit_nan, rmhip_buf* out);
/* `cumsum_scan` / `cumprod_scan` (lib.rs:2884-2891, 2908-2915): running sum (op 0) / product (op 1) along `dim` (zero-based),
 * forward or reverse (`ProviderScanDirection`, :1053-1056), NaN modes of cumsum.rs:586-650 (include: NaN from the first NaN
 * on; omit: NaNs leave the running value unchanged).  Same shape as the input.  Short lines and many strided lines are the CPU's own
 * left-to-right sequence (bit-identical).  LONG lines - contiguous ones, and strided ones of >= 4096 elements when there are too few
 * lines to fill the chip - are scanned in chunks whose totals are carried forward: equal to the CPU up to rounding (exact for
 * integer-valued data), and a chunk TOTAL can overflow where the running prefix would not (cumprod over magnitudes like 1e-300 then
 * 1e+310 inside later chunks: Inf from that chunk on) - the reference's sequential loop is the authority for such data. */
/* @serves cumsum_scan cumprod_scan */
RMHIP_API int rmhip_cumulative(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int reverse, int nan_mode, rmhip_buf* out);

/* `dot` (lib.rs:2722-2728): sum(a .* b) along `dim` (zero-based) of two same-shape tensors; dim < 0
 * = first non-singleton dimension (vectors -> scalar [1,1]). One fused pass, no temporary. */
/* `reduce_mean_nd` (lib.rs:2763-2769; shape rules backend/wgpu/provider/ops/reduction/nd.rs:59-72): reduce several
 * zero-based dimensions, reduced extents become 1.  Dimensions are taken in ascending order one after the other,
 * which is how the CPU computes `mean(x, vecdim)` (mean of means, mean.rs:1107-1116); works for every reduce op. */
/* @serves reduce_mean_nd */
RMHIP_API int rmhip_reduce_nd(rmhip_ctx* ctx, int op, rmhip_buf a, const size_t* dims_zero_based, size_t ndims,
                              int nan_mode, rmhip_buf* out);
/* `reduce_moments_nd` (lib.rs:2770-2778, `ProviderMoments2` :1317-1320): E[x] and E[x^2] over several zero-based
 * dims in one call (same dim handling as rmhip_reduce_nd; NaNs propagate).  Both outputs keep the reduced extents as 1. */
/* @serves reduce_moments_nd */
RMHIP_API int rmhip_reduce_moments_nd(rmhip_ctx* ctx, rmhip_buf a, const size_t* dims_zero_based, size_t ndims,
                                      rmhip_buf* mean_out, rmhip_buf* ex2_out);
/* @serves dot */
RMHIP_API int rmhip_dot(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int dim, rmhip_buf* out);

/* ---- linear algebra  (lib.rs:2375-2405, 2477-2500) ------------------------------------------ */

/* ---- order statistics along a dimension (runmat_amd/csrc/order_ops.hip) ----
 * `cummin_scan` / `cummax_scan` (lib.rs:2918-2935; CPU semantics cummin.rs:719-876, cummax.rs): the running minimum (is_max 0) or maximum
 * along zero-based `dim` (< rank, as the caller guarantees: cummin.rs:676-688), forward or from the end (reverse != 0), and the 1-based
 * position along the dimension of the FIRST occurrence of that extreme in scan order.  nan_mode 0 (include): from the first NaN of a line on
 * the value is NaN and the index is that NaN's position; 1 (omit): NaNs are skipped, and until a number has been seen value AND index are
 * NaN.  Both outputs have the operand's shape.  Bit-exact (copies of input elements and positions). */
/* @serves cummin_scan cummax_scan */
RMHIP_API int rmhip_cumextreme(rmhip_ctx* ctx, int is_max, rmhip_buf a, int dim, int reverse, int nan_mode, rmhip_buf* values,
                               rmhip_buf* indices);
/* `diff_dim(handle, order, dim)` (lib.rs:2596-2603; diff.rs:439-507 through simple_provider.rs:6474-6496): `order` first differences
 * x[k+1] - x[k] along zero-based `dim` (dimensions beyond the rank are extents of one), each pass rounding once; order 0 = a copy; a pass
 * that leaves nothing ends the loop.  column_major 0 reproduces the reference's OUTPUT ORDER: diff_tensor_once pushes the results with
 * k fastest inside every (before, after) line whatever the dimension (diff.rs:493-503; the wgpu shader shaders/diff.rs:26-45 writes the
 * same order), which is the column-major array of the differences only when the dimensions before `dim` have extent one - and the next
 * pass reads that buffer as if it were.  column_major != 0 writes the column-major array (the two agree for dim 0 and for vectors). */
/* @serves diff_dim */
RMHIP_API int rmhip_diff_dim(rmhip_ctx* ctx, rmhip_buf a, size_t order, int dim, int column_major, rmhip_buf* out);
/* `sort_dim(a, dim, order, comparison)` (lib.rs:2358-2366; sort.rs:413-468, 538-574): every line along zero-based `dim` sorted STABLY,
 * ascending or descending (descend != 0); by_abs != 0 orders by |x| first and by x among equal magnitudes (ComparisonMethod::Abs; Auto
 * and Real are by_abs 0 for real data).  NaNs go last ascending and first descending, -0 and +0 compare equal, equal elements keep
 * their input order.  `indices` holds the 1-based original positions.  A dimension beyond the rank or of extent <= 1 returns the values
 * and ones.  The trait's SortResult carries HOST tensors: the shim downloads both buffers (shim/hip_provider.rs). */
/* @serves sort_dim */
RMHIP_API int rmhip_sort_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, int descend, int by_abs, rmhip_buf* sorted, rmhip_buf* indices);
/* `reduce_median_dim(a, dim)` (lib.rs:2839-2845; median.rs:644-741, called in include-NaN mode only: median.rs:425-431): per line along
 * zero-based `dim` NaN if the line holds a NaN, else the middle element of the stably sorted line or 0.5 * (lower + upper); an extent of
 * zero gives NaN, of one the operand.  Output shape = the operand's with extent 1 at `dim` (any rank; the in-process provider takes 2-D
 * only).  dim < 0 is `reduce_median(a)`: ALL elements as one line -> [1, 1] - what both of the reference's providers compute
 * (simple_provider.rs:7167-7193); the host path of median(x, 'all') instead takes medians dimension after dimension
 * (median.rs:531-541), which is a different number on general inputs: callers wanting that chain the per-dimension form. */
/* @serves reduce_median reduce_median_dim */
RMHIP_API int rmhip_reduce_median(rmhip_ctx* ctx, rmhip_buf a, int dim, rmhip_buf* out);

/* `find(a, limit, direction)` (lib.rs:2937-2944 -> ProviderFindResult { linear, rows, cols, values } :623-628; find.rs:593-633,
 * simple_provider.rs:7500-7575): the elements != 0 (a NaN counts) in ascending linear order (last == 0) or in DESCENDING order from the
 * end (last != 0), at most `limit` of them - limit < 0 is `None`: everything for first, ONE for last.  Four [count, 1] outputs: 1-based
 * linear indices, rows = idx % extent0 + 1, cols = idx / extent0 + 1, and the values.  Index work: bit-exact.  Synchronises the stream
 * once (the count sizes the outputs).  In a precision-32 context the index outputs are stored as f32 like every other buffer: exact up to
 * 2^24 elements. */
/* @serves find */
RMHIP_API int rmhip_find(rmhip_ctx* ctx, rmhip_buf a, long long limit_or_neg, int last, rmhip_buf* linear, rmhip_buf* rows, rmhip_buf* cols,
                         rmhip_buf* values);
/* `sort_rows(a, columns, comparison)` (lib.rs:2367-2374; `SortRowsColumnSpec { index, order }`, :1091-1094; sortrows_host.rs:11-140): the rows
 * of a 2-D tensor in lexicographic order of the listed columns (zero-based; column_descend[k] != 0: that key descending; indices beyond the
 * column count are skipped), NaN last ascending / first descending, by_abs as `SortComparison::Abs`; a stable order (equal rows keep
 * theirs).  sorted: a's shape; indices: [rows, 1], 1-based source rows.  One stable sort pass per key, last key first: bit-exact. */
/* @serves sort_rows */
RMHIP_API int rmhip_sort_rows(rmhip_ctx* ctx, rmhip_buf a, const size_t* column_index, const int* column_descend, size_t n_columns, int by_abs, rmhip_buf* sorted,
                              rmhip_buf* indices);
/* `unique(handle, options)` for elements (lib.rs:2645-2651; `UniqueOptions` :1110-1116 with rows == false; unique.rs:473-556): the distinct
 * values - every NaN one value, both zeros one value, each keeping the bits of its FIRST occurrence -, sorted ascending with NaN last or
 * (stable != 0) in order of first occurrence; ia: the 1-based position of each value's first (or last_occurrence != 0: last) occurrence;
 * ic: for every element the 1-based rank of its value.  The results are HOST tensors as in `UniqueResult` (:1118-1124): values_host and
 * ia_host need room for numel doubles (*count are written: shape [count, 1]), ic_host for numel (shape [numel, 1]).  Integer work on
 * sorted (key, position) pairs: bit-exact. */
/* @serves unique */
RMHIP_API int rmhip_unique(rmhip_ctx* ctx, rmhip_buf a, int stable, int last_occurrence, size_t* count, double* values_host, double* ia_host, double* ic_host);
/* `union(a, b, options)` / `setdiff(a, b, options)` for elements (lib.rs:2652-2667; `UnionOptions` / `SetdiffOptions` with rows == false,
 * :1126-1146, :1237-1254; union.rs:491-544, 1238-1279; setdiff.rs:463-496): union - the distinct values of a's elements followed by b's
 * (first occurrences; sorted, or stable != 0: in order of appearance), ia the 1-based positions in a of the values first seen in a, ib
 * those in b of the rest, both in output order; values_host needs numel(a) + numel(b) doubles, ia_host numel(a), ib_host numel(b).
 * setdiff - a's distinct values that do not occur in b, with their first positions in a; numel(a) doubles each.  Host results, as
 * `UnionResult` / `SetdiffResult`.  Bit-exact. */
/* @serves union */
RMHIP_API int rmhip_union(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int stable, size_t* count, double* values_host, size_t* ia_count, double* ia_host,
                          size_t* ib_count, double* ib_host);
/* @serves setdiff */
RMHIP_API int rmhip_setdiff(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int stable, size_t* count, double* values_host, double* ia_host);
/* `ismember(a, b, options)` for elements (lib.rs:2668-2675; `IsMemberOptions { rows: false }`, :1256-1274; ismember.rs:413-438):
 * mask_host[i] = 1 when a's element i occurs in b (NaN matches NaN, -0 matches +0), loc_host[i] = the 1-based lowest position in b or 0;
 * both in a's shape, numel(a) entries (`HostLogicalOwned` / `HostTensorOwned`). */
/* @serves ismember */
RMHIP_API int rmhip_ismember(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, unsigned char* mask_host, double* loc_host);
/* ---- subscript / grid / slice-write hooks and the per-element forms of a real tensor (runmat_amd/csrc/index_ops.hip): bit-exact ----
 * `ndgrid(request)` (lib.rs:1567-1569, ProviderNdgridRequest :3395-3399; simple_provider.rs:2784-2855): for the first `output_count` axes
 * (resident vectors whose lengths equal the leading output extents) the grid out_d[i] = axis_d[(i / stride_d) % extent_d] of shape
 * `output_shape`; output_count == 0 or > n_axes, an empty shape or a length mismatch are errors as there. */
/* @serves ndgrid */
RMHIP_API int rmhip_ndgrid(rmhip_ctx* ctx, const rmhip_buf* axes, size_t n_axes, const size_t* output_shape, size_t rank, size_t output_count,
                           rmhip_buf* outputs);
/* `meshgrid(axes)` (lib.rs:1561-1564; host axes `MeshgridAxisView`, :3376-3378; ops/constructors.rs:230-308): X(iy, ix, iz) = x[ix],
 * Y = y[iy] (and Z = z[iz] when z_or_null != 0) on [ny, nx] - [ny, nx, nz] when nz > 1.  outputs: 2 ids, or 3 with a Z axis. */
/* @serves meshgrid */
RMHIP_API int rmhip_meshgrid(rmhip_ctx* ctx, const double* x, size_t nx, const double* y, size_t ny, const double* z_or_null, size_t nz, rmhip_buf* outputs);
/* `sub2ind(dims, strides, inputs, scalar_mask, len, output_shape)` (lib.rs:3084-3094; simple_provider.rs:8340-8420, 2268-2291): 1-based
 * linear indices 1 + sum (s_d - 1) * stride_d from `rank` resident subscript tensors (a masked one is a scalar); every subscript must be
 * finite, integral (|round(v) - v| <= eps) and within 1..dims[d]: the FIRST offender in the CPU's (element, dimension) order gives
 * RMHIP_ERR_INVALID with the CPU's message ("sub2ind: subscript in dimension 2 must be an integer", ...).  One stream synchronisation. */
/* @serves sub2ind */
RMHIP_API int rmhip_sub2ind(rmhip_ctx* ctx, const size_t* dims, const size_t* strides, const rmhip_buf* inputs, const unsigned char* scalar_mask,
                            size_t rank, size_t len, const size_t* output_shape, size_t out_rank, rmhip_buf* out);
/* `ind2sub(dims, strides, indices, total, len, output_shape)` (lib.rs:3102-3112, `supports_ind2sub` :3097; CPU ind2sub.rs:289-353): `rank`
 * subscript tensors ((idx - 1) / stride_d) % dims[d] + 1 of shape `output_shape`; a non-finite, non-integral or < 1 index is
 * "Linear indices must be positive integers.", one above `total` "Index exceeds number of array elements. Index must not exceed N." -
 * the builtin hands a provider error on to the user (ind2sub.rs:284), so the wording is the CPU's. */
/* @serves ind2sub supports_ind2sub */
RMHIP_API int rmhip_ind2sub(rmhip_ctx* ctx, const size_t* dims, const size_t* strides, size_t rank, rmhip_buf indices, size_t total, size_t len,
                            const size_t* output_shape, size_t out_rank, rmhip_buf* outputs);
/* `scatter_column(matrix, col_index, values)` / `scatter_row(matrix, row_index, values)` (lib.rs:3064-3082; the slice-assignment fast path of
 * runmat-vm/src/indexing/write_slice.rs:680-709): a NEW handle holding the matrix with one whole column (is_column != 0) or row replaced by
 * the `rows` (`cols`) resident values; zero-based index; the operand is untouched. */
/* @serves scatter_column scatter_row */
RMHIP_API int rmhip_scatter_line(rmhip_ctx* ctx, rmhip_buf matrix, int is_column, size_t index, rmhip_buf values, rmhip_buf* out);
/* `pow2_scale(mantissa, exponent)` (lib.rs:2325-2331; simple_provider.rs:5822-5850): m .* 2.^e for operands of one shape (else
 * "shape mismatch"); an integral exponent is the exact power of two (bit-exact product), a fractional one goes through exp2 (2 ulp). */
/* @serves pow2_scale */
RMHIP_API int rmhip_pow2_scale(rmhip_ctx* ctx, rmhip_buf mantissa, rmhip_buf exponent, rmhip_buf* out);
/* `round_digits(a, digits, significant)` (lib.rs:2197-2204; simple_provider.rs:5359-5420): round(x * 10^digits) / 10^digits with the factor
 * formed as Rust's powi does (square-and-multiply, reciprocal last), half away from zero, non-finite values and overflowing factors passed
 * through: bit-exact.  significant != 0 (digits counted from floor(log10|x|) of the host's libm) is RMHIP_ERR_UNSUPPORTED. */
/* @serves round_digits */
RMHIP_API int rmhip_round_digits(rmhip_ctx* ctx, rmhip_buf a, int digits, int significant, rmhip_buf* out);
/* `unary_real / unary_imag / unary_conj / unary_angle` on REAL storage (lib.rs:2217-2240; simple_provider.rs:5482-5640): part 0 real = the
 * operand, 1 imag = zeros, 2 conj = the operand, 3 angle = atan2(+0, x): +0 for x > 0 and +0, pi for x < 0 and -0, NaN for NaN.
 * `logical_isreal` (lib.rs:2055; simple_provider.rs:4786): always 1 here - this library has no complex-interleaved storage. */
/* @serves unary_real unary_imag unary_conj unary_angle */
RMHIP_API int rmhip_real_part(rmhip_ctx* ctx, int part, rmhip_buf a, rmhip_buf* out);
/* @serves logical_isreal */
RMHIP_API int rmhip_isreal(rmhip_ctx* ctx, rmhip_buf a, int* result);

/* ---- small construction / linear-algebra hooks (runmat_amd/csrc/misc_ops.hip): one or two rounded operations per element, bit-exact ----
 * `diag_from_vector(vector, offset)` / `diag_from_vector_sized(vector, offset, rows, cols)` (lib.rs:1600-1623; simple_provider.rs:3222-3281):
 * element idx of a vector-like operand on (idx, idx + offset) or (idx - offset, idx), zeros elsewhere; rows / cols < 0 = the square of
 * size len + |offset|; elements that fall outside an explicit size are dropped.  A matrix operand is RMHIP_ERR_UNSUPPORTED. */
/* @serves diag_from_vector diag_from_vector_sized */
RMHIP_API int rmhip_diag_from_vector(rmhip_ctx* ctx, rmhip_buf vector, long long offset, long long rows_or_neg, long long cols_or_neg,
                                     rmhip_buf* out);
/* `kron(a, b)` (lib.rs:2697-2699; kron.rs:358-485): shapes padded with ones to a common rank (<= 8), out[a_c * extent_b + b_c] = a * b. */
/* @serves kron */
RMHIP_API int rmhip_kron(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out);
/* `cross(lhs, rhs, dim)` (lib.rs:2701-2708; cross.rs:332-364, 443-467): operands of one shape; `dim` ONE-based as the trait passes it,
 * 0 = None = the first dimension of extent 3; a dimension beyond the rank or not of extent 3 is RMHIP_ERR_INVALID.  Each component
 * is two products and a difference, unfused. */
/* @serves cross */
RMHIP_API int rmhip_cross(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int dim_one_based_or_0, rmhip_buf* out);
/* `gradient_dim(handle, dim, spacing)` / `gradient_dim_with_coordinates(handle, dim, coordinates)` (lib.rs:2604-2620;
 * gradient.rs:650-720, 814-833): along zero-based `dim`, (x[1] - x[0]) / h and (x[n-1] - x[n-2]) / h at the ends, (x[k+1] - x[k-1]) /
 * (2 h) inside - or the matching coordinate differences when `coordinates` (a resident vector of the dimension's extent) is given;
 * an extent of one gives zeros.  A zero coordinate difference is the caller's to refuse (gradient.rs:1044-1051): here it divides. */
/* @serves gradient_dim gradient_dim_with_coordinates */
RMHIP_API int rmhip_gradient_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, double spacing, rmhip_buf coordinates_or_0, rmhip_buf* out);
/* `trapz_dim` / `cumtrapz_dim(input, dim, spacing)` (lib.rs:2893-2908, ProviderTrapezoidSpacing :1060-1066; simple_provider.rs:2421-2598):
 * along zero-based `dim`, terms 0.5 * w_k * (x[k] + x[k+1]) formed exactly as on the CPU and summed by the library's reduction (trapz:
 * extent 1 at `dim`) or cumulative scan (cumtrapz: the operand's shape, 0 at k = 0).  spacing_kind 0 Unit, 1 Scalar(`scalar`),
 * 2 ScalarHandle (first element of `spacing`), 3 Vector (coordinates of the dimension), 4 Tensor (coordinates of the operand's shape).
 * The CPU sums the terms strictly in order: results agree to the summation-order tolerance of `cumsum` / `sum` (n eps sum|t|). */
/* @serves trapz_dim cumtrapz_dim */
RMHIP_API int rmhip_trapz_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, int cumulative, int spacing_kind, double scalar, rmhip_buf spacing_or_0,
                              rmhip_buf* out);
/* `norm(tensor, order)` (lib.rs:2451-2457, ProviderNormOrder :745-754; CPU norm.rs:269-529) -> a [1, 1] tensor.  order 1 One, 2 Two,
 * 3 Inf, 4 NegInf, 5 Zero, 6 Fro, 7 Nuc, 8 P(p).  Vectors (a dimension <= 1): sum |x|, root of the sum of squares (scaled by a power
 * of two when the squares would overflow or underflow), max / min |x| (an empty or all-infinite minimum is 0), the count of nonzeros,
 * (sum |x|^p)^(1/p) for finite p >= 1.  Matrices: One = largest column sum of |a|, Inf = largest row sum, Fro as for vectors.  Any NaN
 * gives NaN.  The matrix 2-norm and the nuclear norm are the largest / the sum of the singular values of the one-sided Jacobi
 * decomposition (min(rows, cols) <= 4096 and finite data, else RMHIP_ERR_UNSUPPORTED); orders the builtin refuses (matrix
 * -Inf / 0 / p, a vector's nuclear norm, p < 1) are RMHIP_ERR_INVALID.  Sums in a different order than the CPU's loops: n eps relative. */
/* @serves norm */
RMHIP_API int rmhip_norm(rmhip_ctx* ctx, rmhip_buf a, int order, double p, rmhip_buf* out);
/* `issymmetric(matrix, kind, tolerance)` (lib.rs:3115-3124; issymmetric.rs:461-487, 517-526): 1 when a(i,j) equals a(j,i) (skew != 0:
 * -a(j,i), and a zero diagonal), pairs compared as `v == r || (both finite && |v - r| <= tolerance)`; a non-square operand is 0, an
 * operand with trailing extents > 1 RMHIP_ERR_INVALID.  Synchronises the stream (a host bool comes back). */
/* @serves issymmetric */
RMHIP_API int rmhip_issymmetric(rmhip_ctx* ctx, rmhip_buf a, int skew, double tolerance, int* result);
/* ---- convolutions and windows (signal_ops.hip) ---------------------------------------------------------------------------------------
 * `conv1d(signal, kernel, options)` (lib.rs:2535-2542; conv.rs:481-517): the 1-D convolution of the two tensors' elements (any shapes,
 * taken in storage order), direct sums in the CPU's order - bit-exact.  mode: 0 full, 1 same, 2 valid (`ProviderConvMode`, lib.rs:1277-1281);
 * column != 0: the result is [len, 1], otherwise [1, len] (`ProviderConvOrientation`); an empty operand or a valid-mode kernel longer than
 * the signal gives the empty result of that orientation (simple_provider.rs:1780-1787). */
/* @serves conv1d */
RMHIP_API int rmhip_conv1d(rmhip_ctx* ctx, rmhip_buf signal, rmhip_buf kernel, int mode, int column, rmhip_buf* out);
/* `conv2d(signal, kernel, mode)` (lib.rs:2543-2550; conv2.rs:595-640, the reference's kernel indexing included): 2-D operands (trailing
 * singleton dimensions allowed, else RMHIP_ERR_INVALID "input must be 2-D"); bit-exact; empty operands give [0, 0] (same: the signal's shape). */
/* @serves conv2d */
RMHIP_API int rmhip_conv2d(rmhip_ctx* ctx, rmhip_buf signal, rmhip_buf kernel, int mode, rmhip_buf* out);
/* `moving_window(request)` (lib.rs:2852-2857, `ProviderMovingWindowRequest` :990-1003; moving.rs:737-825): movsum / movmean / movprod /
 * movmin / movmax / movmedian / movstd / movvar over the count window [center - before, center + after] along zero-based `dim`.
 * op: 0 sum 1 mean 2 prod 3 min 4 max 5 median 6 std 7 var (`ProviderMovingWindowOp`); endpoints: 0 shrink, 1 discard, 2 fill(`fill`);
 * nan_omit / population: `ProviderNanMode::Omit` / `ProviderStdNormalization::Population`.  out_shape: the request's output shape (checked).
 * Bit-exact (the CPU's accumulation order).  RMHIP_ERR_UNSUPPORTED: a median window of more than 64 points; a product padded with a
 * value other than 0, 1 or NaN (the CPU multiplies by `powf(fill, count)`). */
/* @serves moving_window */
RMHIP_API int rmhip_moving_window(rmhip_ctx* ctx, rmhip_buf a, int dim, size_t before, size_t after, int op, int endpoints, double fill, int nan_omit,
                                  int population, const size_t* out_shape, size_t out_rank, rmhip_buf* out);
/* `iir_filter(b, a, x, options)` (lib.rs:2551-2559; `ProviderIirFilterOptions { dim, zi, unit_denominator }`, :1295-1306; filter.rs:1119-1222):
 * `filter(b, a, x, zi, dim)` in direct form II transposed along zero-based `dim`, every channel an independent recurrence (one thread per
 * channel); coefficients normalised by a(1) as the CPU's complex division rounds them; `zi_or_0`: initial states in the signal's shape with
 * `dim` of extent max(nb, na) - 1.  output: x's shape; final_state: that state shape (`ProviderIirFilterResult`).  Bit-exact.
 * RMHIP_ERR_UNSUPPORTED: order > 64; fewer than 256 channels on more than 4096 samples (a recurrence: the host's one core is faster). */
/* @serves iir_filter */
RMHIP_API int rmhip_iir_filter(rmhip_ctx* ctx, rmhip_buf b, rmhip_buf a, rmhip_buf x, int dim, rmhip_buf zi_or_0, int unit_denominator, rmhip_buf* output,
                               rmhip_buf* final_state);
/* `imfilter(image, kernel, options)` (lib.rs:1809-1817; `ImfilterOptions`, :1193-1222; imfilter.rs:476-545, 620-783): N-D correlation (or,
 * convolution != 0, convolution: the kernel read back to front) of up to four dimensions.  padding: 0 constant (`constant_value`), 1 replicate,
 * 2 symmetric, 3 circular; shape_mode: 0 same, 1 full, 2 valid; the kernel's origin is floor(extent / 2) per dimension.  Sums over the kernel's
 * points in storage order, products rounded before the sum: bit-exact. */
/* @serves imfilter */
RMHIP_API int rmhip_imfilter(rmhip_ctx* ctx, rmhip_buf image, rmhip_buf kernel, int padding, double constant_value, int shape_mode, int convolution,
                             rmhip_buf* out);
/* `interp1(request)` (lib.rs:2458-2463; `ProviderInterp1Request`, :769-783; simple_provider.rs:1396-1472, 8135-8204): every series of y
 * (sample_len values each, back to back) interpolated at the query_len points of xq over the strictly increasing coordinates x - the
 * result holds series after series, in `output_shape`.  nearest: `ProviderInterp1Method::Nearest` (ties to the left sample), else Linear
 * (y0 + ((xq - x0) / (x1 - x0)) * (y1 - y0), as the CPU rounds it); extrapolation: 0 NaN, 1 extrapolate, 2 `extrapolation_value`; a
 * non-finite query yields NaN.  Bit-exact. */
/* @serves interp1 */
RMHIP_API int rmhip_interp1(rmhip_ctx* ctx, rmhip_buf x, rmhip_buf y, rmhip_buf xq, size_t sample_len, size_t series_count, size_t query_len,
                            const size_t* output_shape, size_t out_rank, int nearest, int extrapolation, double extrapolation_value, rmhip_buf* out);
/* `polyval(coefficients, points, options)` (lib.rs:1652-1660; polyval.rs:886-905): Horner's rule over the coefficients (highest power
 * first) at every point, the result in the points' shape; has_mu: the point is centred and scaled first, ((x - mean) * scale) / (scale *
 * scale) as the CPU's complex division rounds it.  Bit-exact while every intermediate is finite; otherwise RMHIP_ERR_UNSUPPORTED (the
 * CPU's complex recurrence yields NaN + NaN i there: a complex result the caller forms on the host).  Synchronises the stream once. */
/* @serves polyval */
RMHIP_API int rmhip_polyval(rmhip_ctx* ctx, rmhip_buf coefficients, rmhip_buf points, int has_mu, double mean, double scale, rmhip_buf* out);
/* `polyder_single(p)`, `polyder_product(p, q)`, `polyder_quotient(u, v)` (lib.rs:1674-1701; simple_provider.rs:3137-3188): the derivative
 * of a polynomial (coefficient vector, highest power first), of a product (p'q + pq') or of a quotient (numerator u'v - uv' in `out`,
 * denominator v*v in `denominator_or_null`); q_or_0 = 0 selects the single form.  Results are trimmed of leading coefficients with
 * |c| <= 1e-12 ([0] when nothing is left) and take the first operand's orientation (the denominator: v's); an input with more than one
 * extent above 1 is RMHIP_ERR_INVALID.  Bit-exact: the CPU's sums in the CPU's order.  Synchronises the stream (the length is data). */
/* @serves polyder_single polyder_product polyder_quotient */
RMHIP_API int rmhip_polyder(rmhip_ctx* ctx, rmhip_buf p, rmhip_buf q_or_0, int quotient, rmhip_buf* out, rmhip_buf* denominator_or_null);
/* `polyint(polynomial, constant)` (lib.rs:1704-1710; simple_provider.rs:374-390, 3190-3215): coefficient i divided by its new power, the
 * constant appended; real polynomials (a complex-interleaved one is RMHIP_ERR_UNSUPPORTED).  Bit-exact. */
/* @serves polyint */
RMHIP_API int rmhip_polyint(rmhip_ctx* ctx, rmhip_buf p, double constant, rmhip_buf* out);
/* `hann_window / hamming_window / blackman_window(len, periodic)` (lib.rs:1797-1807; simple_provider.rs:95-120) -> [len, 1]; kind 0 / 1 / 2.
 * One cosine (two for Blackman) per point: within 2 ulp of the cosine of the CPU's libm (tests state the bound). */
/* @serves hann_window hamming_window blackman_window */
RMHIP_API int rmhip_window(rmhip_ctx* ctx, int kind, size_t len, int periodic, rmhip_buf* out);

/* ---- discrete Fourier transforms and complex-interleaved storage (fft.hip) --------------------------------------------------------
 * A transform's result is a COMPLEX-INTERLEAVED tensor (`GpuTensorStorage::ComplexInterleaved`, lib.rs:247-251): its shape is the
 * logical one, its storage 2 * numel doubles (re, im, re, im, ...).  `rmhip_download` hands such a tensor back as 2 * numel doubles
 * (what `HostTensorOwned { data, shape, storage }` carries, lib.rs:3362-3366); `rmhip_storage` tells which kind an id is.  Only the
 * entry points of this section accept complex tensors; every other one returns RMHIP_ERR_UNSUPPORTED for them (the caller gathers,
 * as for any `Err`).  A precision-32 context rounds the values of a complex result through f32 (storage stays 2 x f64). */
/* @serves - */
RMHIP_API int rmhip_storage(rmhip_ctx* ctx, rmhip_buf id, int* complex_interleaved);
/* `fft_dim(handle, len, dim)` / `ifft_dim` (lib.rs:2622-2638; semantics of the wgpu provider's host form, ops/fft/fallback.rs:4-150):
 * the DFT of every line along zero-based `dim` (a dimension beyond the rank has extent 1 and extends the shape), zero-padded or
 * truncated to `len_or_neg` points (< 0: the extent); forward unnormalised (exp(-2 pi i jk / n)), inverse scaled by 1 / n; real or
 * complex input; any length up to 2^24 (2^27 for a power of two along dimension 0; powers of two by LDS-resident radix-8 passes, others by Bluestein's chirp convolution).
 * Accuracy: error <= a small multiple of eps * log2(n) * ||line||_2 per point (tests/test_gpu_fft.py); the reference transforms
 * with rustfft 6.4.1, so parity is by that tolerance, not by bits. */
/* @serves fft_dim ifft_dim */
RMHIP_API int rmhip_fft_dim(rmhip_ctx* ctx, rmhip_buf a, long long len_or_neg, int dim, int inverse, rmhip_buf* out);
/* `signal_hilbert(request)` (lib.rs:2572-2577; `ProviderHilbertRequest { input, length, dim }`, :331-338; hilbert.rs:349-412): the analytic signal of a
 * real tensor along zero-based `dim` (< rank), padded / truncated to `len_or_neg` points (< 0: the extent; 0 is invalid): forward transform, the
 * one-sided mask (1, 2 ... 2, [1], 0 ... 0), inverse transform - a complex-interleaved result whose real part is the input.  Tolerance as the
 * transforms'. */
/* @serves signal_hilbert */
RMHIP_API int rmhip_hilbert(rmhip_ctx* ctx, rmhip_buf a, long long len_or_neg, int dim, rmhip_buf* out);
/* `complex_from_real(real)` (imag_or_0 == 0) / `complex_from_real_imag(real, imag)` (lib.rs:1940-1959): equal shapes, or either
 * operand a one-element tensor that expands. */
/* @serves complex_from_real complex_from_real_imag */
RMHIP_API int rmhip_complex(rmhip_ctx* ctx, rmhip_buf real, rmhip_buf imag_or_0, rmhip_buf* out);
/* `zeros_with_storage(shape, ComplexInterleaved)` (lib.rs:1472-1489): a zero complex tensor (the Real kind is `rmhip_fill`). */
/* @serves zeros_with_storage */
RMHIP_API int rmhip_zeros_complex(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out);
/* `fft_extract_real(handle)` (lib.rs:2639-2644; ifft(..., 'symmetric'), ifft.rs:362-372): the real parts as a new real tensor of the
 * same shape (a real input is copied: the caller frees the handle it passed). */
/* @serves fft_extract_real */
RMHIP_API int rmhip_complex_real(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out);

/* `ishermitian(matrix, kind, tolerance)` for this backend's real data (lib.rs:3126-3138; ishermitian.rs:455-482, 522-530): the test of
 * `rmhip_issymmetric` with one more rule - the Hermitian kind fails on a NaN diagonal entry.  Same shape rules and errors. */
/* @serves ishermitian */
RMHIP_API int rmhip_ishermitian(rmhip_ctx* ctx, rmhip_buf a, int skew, double tolerance, int* result);
/* `bandwidth(matrix)` (lib.rs:3140-3143; bandwidth.rs:303-318, 341-365): lower = max(row - col), upper = max(col - row) over the entries
 * that are non-zero or NaN; (0, 0) for an empty matrix; a rank-1 shape is a row; trailing dimensions > 1 -> RMHIP_ERR_INVALID. */
/* @serves bandwidth */
RMHIP_API int rmhip_bandwidth(rmhip_ctx* ctx, rmhip_buf a, unsigned* lower, unsigned* upper);

/* `matmul`: C = A*B, 2-D, column-major; inner dims must agree else RMHIP_ERR_SHAPE
 * (simple_provider.rs:7698-7741). fp64 MFMA kernel. */
/* @serves matmul */
RMHIP_API int rmhip_matmul(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out);
/* `matmul_epilogue` (lib.rs:2394-2405, descriptor `MatmulEpilogue` lib.rs:3498-3560): the epilogue is
 * folded into the dgemm store.  Order (simple_provider.rs:7800-7836): v = acc*alpha + beta; row scale;
 * column scale; clamp_min (max); clamp_max (min); pow; then the diagonal copy.  Buffer ids of 0 mean
 * "absent".  `diag_output` (length >= min(m,n)) is written in place -- the one documented exception to
 * "inputs are never mutated". */
typedef struct rmhip_matmul_epilogue {
    double alpha, beta;
    rmhip_buf row_scale, col_scale;  /* 0 = none; lengths m / n */
    int row_op, col_op;              /* ScaleOp: 0 = Multiply, 1 = Divide */
    int has_clamp_min, has_clamp_max, has_pow;
    double clamp_min, clamp_max, pow_exponent;
    rmhip_buf diag_output;           /* 0 = none */
} rmhip_matmul_epilogue_t;
/* @serves matmul_epilogue */
RMHIP_API int rmhip_matmul_epilogue(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b,
                                    const rmhip_matmul_epilogue_t* ep, rmhip_buf* out);
/* `matmul_power_step` + PowerStepEpilogue (lib.rs:2414-2421, 3553-3561; CPU semantics
 * crates/runmat-accelerate/src/simple_provider.rs:7852-7891): P = lhs * rhs, then every column of P is divided by
 * sqrt(sum(P(:,c).^2) + epsilon)  (the power-iteration step of the PCA benchmark). */
/* @serves matmul_power_step */
RMHIP_API int rmhip_matmul_power_step(rmhip_ctx* ctx, rmhip_buf lhs, rmhip_buf rhs, double epsilon, rmhip_buf* out);
/* `image_normalize` + ImageNormalizeDescriptor (lib.rs:2407-2413, 3563-3577; CPU semantics simple_provider.rs:7893-7993):
 * input is [batch, height, width]; per batch element mean / two-pass variance over the plane, then
 * y = (x - mean) / sqrt(var + epsilon) [* gain] [+ bias] [max 0] [^ gamma].  Any batch extent (up to 256 planes: one block layout with fixed planes per
 * thread; more: the statistics by the strided moments reduction and flat apply threads). */
typedef struct rmhip_image_normalize {
    size_t batch, height, width;
    double epsilon;
    int has_gain, has_bias, has_gamma, clamp_zero;
    double gain, bias, gamma;
} rmhip_image_normalize_t;
/* @serves image_normalize */
RMHIP_API int rmhip_image_normalize(rmhip_ctx* ctx, rmhip_buf input, const rmhip_image_normalize_t* desc, rmhip_buf* out);
/* `covariance` (lib.rs:1857-1865, CovarianceOptions :937-953) for the dense unweighted case the CenteredGram fusion
 * pattern issues (fusion_exec.rs:630-672: second = None, weights = None, rows = All): column means, centring,
 * (Xc' * Xc) / denom with denom = rows - 1 (biased = 0) or rows (biased = 1), cov.rs:916-953, 1080-1100, 1218-1227.
 * rows - 1 <= 0 gives the all-NaN matrix the CPU returns.  The centred product runs as A'*A on the MFMA path. */
/* @serves covariance */
RMHIP_API int rmhip_covariance(rmhip_ctx* ctx, rmhip_buf matrix, int biased, rmhip_buf* out);
/* `rank(matrix, tolerance)` (lib.rs:2464-2470; rank.rs:280-295), `cond(matrix, norm)` for the 2-norm (norm == 0; lib.rs:2444-2450; cond.rs:276-330,
 * 448-467; the other norms: RMHIP_ERR_UNSUPPORTED) and `pinv(matrix, options)` (lib.rs:2437-2443; pinv.rs:242-285) from the one-sided Jacobi SVD of
 * svdsolve.hip (min(rows, cols) <= 4096, finite data; else RMHIP_ERR_UNSUPPORTED): rank = singular values above the tolerance (default max(m, n) *
 * eps(s_max), common/linalg.rs:209-228), cond = s_max / s_min (inf when s_min == 0; 0 for an empty matrix), pinv = V diag(1 / s_i, s_i > tol) U' as
 * [cols, rows].  rank and cond return [1, 1] tensors.  The CPU decomposes with nalgebra: parity by tolerance (singular values to relative accuracy). */
/* @serves rank */
RMHIP_API int rmhip_rank(rmhip_ctx* ctx, rmhip_buf matrix, int has_tolerance, double tolerance, rmhip_buf* out);
/* @serves cond */
RMHIP_API int rmhip_cond(rmhip_ctx* ctx, rmhip_buf matrix, int norm, rmhip_buf* out);
/* `rcond(matrix)` (lib.rs:2471-2476; rcond.rs:304-320): s_min / s_max of a square matrix from the same decomposition (0 when s_max == 0, inf for the
 * empty matrix; a non-square input is RMHIP_ERR_INVALID). */
/* @serves rcond */
RMHIP_API int rmhip_rcond(rmhip_ctx* ctx, rmhip_buf matrix, rmhip_buf* out);
/* @serves pinv */
RMHIP_API int rmhip_pinv(rmhip_ctx* ctx, rmhip_buf matrix, int has_tolerance, double tolerance, rmhip_buf* out);
/* `covariance_to_correlation(matrix)` (lib.rs:1876-1884; simple_provider.rs:885-975): a covariance matrix validated as the CPU validates it
 * (finite-or-NaN entries, non-negative diagonal, symmetric to 1e-10 relative, |cov| within the variance bound - the CPU's messages, its order of
 * checks) and scaled: correlation = cov / (sd_i sd_j) (NaN where that product is zero), sigma = sqrt(diag) as [n, 1].  Bit-exact; one
 * stream synchronisation for the verdict. */
/* @serves covariance_to_correlation */
RMHIP_API int rmhip_covariance_to_correlation(rmhip_ctx* ctx, rmhip_buf matrix, rmhip_buf* correlation, rmhip_buf* sigma);
/* `peaks(n)` (x_or_0 == y_or_0 == 0; lib.rs:1781-1785) / `peaks_xy(x, y)` (lib.rs:1787-1795; peaks.rs:511-550): the `peaks` test surface on the
 * n x n grid over [-3, 3]^2 (n == 1: the point (3, 3)), or at same-shape coordinate tensors.  Products and sums in the CPU's order; three
 * exponentials per point: within 2e-14 absolute of the oracle (terms of magnitude up to ~8). */
/* @serves peaks peaks_xy */
RMHIP_API int rmhip_peaks(rmhip_ctx* ctx, size_t n, rmhip_buf x_or_0, rmhip_buf y_or_0, rmhip_buf* out);
/* `corrcoef(matrix, options)` (lib.rs:1867-1874; `CorrcoefOptions { normalization, rows }`, :906-911; corrcoef.rs:720-787, 895-926) for
 * rows == All (rows_mode 0; Complete / Pairwise: RMHIP_ERR_UNSUPPORTED): the covariance path above, then r = cov / (sd_i sd_j) with the CPU's
 * NaN rules (a variance that is not finite and positive), its 1e-12 clamp onto [-1, 1] and an exact unit diagonal.  Sums of products:
 * parity by tolerance, as for `covariance` (the reference's own test allows 1e-10). */
/* @serves corrcoef */
RMHIP_API int rmhip_corrcoef(rmhip_ctx* ctx, rmhip_buf matrix, int biased, int rows_mode, rmhip_buf* out);
/* `diag_extract` (lib.rs:1625-1632; simple_provider.rs:3281-3312): the offset-th diagonal of a matrix as a column
 * vector [len, 1]; vectors are rejected ("matrix input required"). */
/* @serves diag_extract */
RMHIP_API int rmhip_diag_extract(rmhip_ctx* ctx, rmhip_buf matrix, long long offset, rmhip_buf* out);
/* `lu` -> ProviderLuResult {combined, lower, upper, perm_matrix, perm_vector} (lib.rs:649-698);
 * pivot rule and singular cut-off of host_lu.rs:37-59.  out5 order: combined, L, U, P, pivots. */
/* @serves lu */
RMHIP_API int rmhip_lu(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf out5[5]);
/* `mldivide`: x = A\b.  Square, numerically non-singular A: blocked LU with partial pivoting.  Rectangular FULL-RANK A with a
 * Gram pivot ratio >= 1e-11 (cond(A) up to ~3e5): the least-squares (rows > cols) / minimum-norm (rows < cols) solution the
 * reference's SVD solve returns (mldivide.rs:380-404), through the LU of A'A or AA' and one refinement step.  Anything
 * else (a pivot <= 1e-12, rank-deficient or ill-conditioned rectangular systems) returns SINGULAR / UNSUPPORTED so the
 * caller uses its CPU SVD path (mldivide.rs:223-229 `.ok()`).  Scalar A => b * (1/A) (mldivide.rs:321-325). */
/* @serves mldivide */
RMHIP_API int rmhip_mldivide(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out);
/* `mrdivide` (lib.rs:2484-2490): X = B / A, i.e. X * A = B.  CPU semantics crates/runmat-runtime/src/builtins/math/linalg/ops/
 * mrdivide.rs:317-341 (scalar A: B * (1/A); column counts must agree) and :379-388 (the solve is A' \ B' transposed
 * back); here the same transposition around the LU solve, with the same soft failures as rmhip_mldivide. */
/* @serves mrdivide */
RMHIP_API int rmhip_mrdivide(rmhip_ctx* ctx, rmhip_buf b, rmhip_buf a, rmhip_buf* out);
/* `chol(a, lower)` (lib.rs:2502-2508 -> ProviderCholResult { factor, info } :658-662; host algorithm chol.rs:374-433: the reference's own
 * Cholesky-Crout with a symmetry check of every pair to 1e-12 relative): the upper factor R (A = R'R) or, lower != 0, L = R'.  The SUCCESS
 * path only - a recursive blocked factorisation (deep MFMA products, 64 x 64 leaves in LDS), *info = 0, forward error ~ cond * eps against
 * the host's in-order sums.  A matrix that fails the symmetry check or has a pivot that is not positive and finite is
 * RMHIP_ERR_UNSUPPORTED: the builtin falls back (chol.rs:331-342) and its host code produces `info` and the partial factor. */
/* @serves chol */
RMHIP_API int rmhip_chol(rmhip_ctx* ctx, rmhip_buf a, int lower, rmhip_buf* factor, unsigned* info);
/* `inv(matrix, options)` (lib.rs:2430-2436, ProviderInvOptions {} :716; CPU inv.rs:209-230, 258-280: nalgebra 0.32.6 `try_inverse`, an LU
 * with partial pivoting and substitutions on the identity - absent from /root/reference, parity by residual as for mldivide): X = A \ I on
 * the LU path.  Scalars, [n, n] and [n, n, 1, ...] operands (the shape is kept); a non-square or higher-rank operand is
 * RMHIP_ERR_INVALID with the reference's wording; a pivot below the solver's cut-off is RMHIP_ERR_SINGULAR (the CPU path then raises
 * "matrix is singular to working precision" or returns nalgebra's answer for a pivot in (0, 1e-12]).  [0, 0] gives [0, 0]. */
/* @serves inv */
RMHIP_API int rmhip_inv(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out);
/* `linsolve` + ProviderLinsolveOptions / ProviderLinsolveResult (lib.rs:2422-2429, 679-697); CPU
 * semantics crates/runmat-runtime/src/builtins/math/linalg/solve/linsolve.rs:691-726 (option order:
 * TRANSA transposes A and swaps LT<->UT), 769-833 (substitution; a zero diagonal entry is the
 * "singular to working precision" error; rcond = min|d_ii| / max|d_ii|), 1000-1009 (RCOND threshold).
 *   - lower / upper : triangular solve on the device, *rcond as the reference computes it.
 *   - general square: LU solve as rmhip_mldivide; *rcond = NaN.  With need_rcond or has_rcond the
 *     reference reports sigma_min/sigma_max of an SVD (linsolve.rs:933-944): UNSUPPORTED here so the
 *     caller takes its CPU path (linsolve.rs:414-417 `.ok()`).
 *   - general rectangular (no rcond requested): full-rank least squares / minimum norm as rmhip_mldivide; *rcond = NaN.
 *   - conjugate / symmetric / posdef are accepted and, as in the reference's real path, have no effect. */
typedef struct rmhip_linsolve_options {
    int lower, upper, rectangular, transposed, conjugate, symmetric, posdef, need_rcond;
    int has_rcond;   /* Option<f64> rcond */
    double rcond;
} rmhip_linsolve_options_t;
/* @serves linsolve */
RMHIP_API int rmhip_linsolve(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, const rmhip_linsolve_options_t* opts,
                             rmhip_buf* out, double* reciprocal_condition);
/* `transpose` (lib.rs:2532): out[j,i] = a[i,j] for a 2-D tensor, shape [cols, rows].  Like the reference's wgpu
 * provider (ops/tensor.rs:828-846, `record_handle_transpose` lib.rs:218-245) the result is a VIEW that aliases the
 * operand's storage: rmhip_matmul / rmhip_syrk read it in place through transposed-operand MFMA kernels (`A'*B`,
 * `A*B'`), every other entry point sees a materialised copy on first use (one 64x64-tile LDS transpose pass). */
/* @serves transpose */
RMHIP_API int rmhip_transpose(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out);
/* `syrk` (lib.rs:2383): A' * A (cols x cols), reference loop accelerate/tests/syrk.rs:14-31. */
/* @serves syrk */
RMHIP_API int rmhip_syrk(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out);

/* ---- shape / indexing hooks the hot-path builtins call around the kernels above --------------------------- */

/* `repmat` (lib.rs:2689-2695; tiling rule crates/runmat-accelerate/src/simple_provider.rs:2174-2240, 6681-6697).  `reps` has
 * n_reps >= 1 factors; ONE factor r means r along every dimension of max(rank, 2) dimensions, otherwise the rank is
 * max(rank, n_reps) with missing factors 1; the shape is base .* factors.
 * The result is a VIEW that aliases the operand's storage (like rmhip_transpose): the reference's plus / minus / times /
 * rdivide / power builtins expand an operand with `repmat` only to hand it to `elem_*` and free it again
 * (math/elementwise/times.rs:501-543 - the error of a provider without `repmat` goes to the user there, `?`), so
 * rmhip_binary and rmhip_fused_elementwise read the view in place with stride-0 indexing and the tiled tensor never exists
 * in HBM; any other entry point (download included) materialises it once, on first use, under the same id.  The base may
 * be freed while the view lives.  A tiled extent of 0 gives an empty tensor. */
/* @serves repmat */
RMHIP_API int rmhip_repmat(rmhip_ctx* ctx, rmhip_buf a, const size_t* reps, size_t n_reps, rmhip_buf* out);
/* `permute` (lib.rs:2579-2585; simple_provider.rs:1645-1740, 6402-6417): `order` is zero-based, a permutation of
 * 0..n_order-1 with n_order >= rank (the shape is padded with 1s); out dimension d is source dimension order[d]. */
/* @serves permute */
RMHIP_API int rmhip_permute(rmhip_ctx* ctx, rmhip_buf a, const size_t* order, size_t n_order, rmhip_buf* out);
/* `zeros_like` / `ones_like` / `fill_like` (lib.rs:1497, 1547, 1524-1545): a constant tensor of the prototype's shape
 * (power.rs:577,617, reduction/max.rs:2047-2105, any.rs:397 ask for these next to a resident operand). */
/* @serves zeros_like ones_like fill_like */
RMHIP_API int rmhip_fill_like(rmhip_ctx* ctx, rmhip_buf prototype, double value, rmhip_buf* out);
/* `read_scalar` (lib.rs:1463; simple_provider.rs:3415-3429): element `linear_index` (zero-based, column-major) of a
 * resident tensor as f64 - every `y(i)` on a device result (common/indexing.rs:533).  Blocks on the context's stream;
 * views are indexed in place, nothing is materialised.  Out of range: RMHIP_ERR_INVALID. */
/* @serves read_scalar */
RMHIP_API int rmhip_read_scalar(rmhip_ctx* ctx, rmhip_buf a, size_t linear_index, double* out);
/* `gather_linear` (lib.rs:1423-1430; simple_provider.rs:2609-2654): out[k] = source[indices[k]] (zero-based linear
 * indices) with shape `out_shape` (prod == n_indices).  An index >= numel(source) is RMHIP_ERR_INVALID. */
/* @serves gather_linear */
RMHIP_API int rmhip_gather_linear(rmhip_ctx* ctx, rmhip_buf source, const uint32_t* indices, size_t n_indices,
                                  const size_t* out_shape, size_t rank, rmhip_buf* out);
/* `scatter_linear` (lib.rs:1438-1445; simple_provider.rs:2656-2720): target[indices[k]] = values[k], in place (the
 * trait's one mutating hook); later duplicates win, as in the reference's sequential loop.  numel(values) must be
 * n_indices. */
/* @serves scatter_linear */
RMHIP_API int rmhip_scatter_linear(rmhip_ctx* ctx, rmhip_buf target, const uint32_t* indices, size_t n_indices, rmhip_buf values);
/* `linspace` (lib.rs:1887; simple_provider.rs:3488-3513): [1, count]; element i = start + i * ((stop - start) / (count - 1)),
 * the last one is `stop` exactly; count == 1 gives [stop]. */
/* @serves linspace */
RMHIP_API int rmhip_linspace(rmhip_ctx* ctx, double start, double stop, size_t count, rmhip_buf* out);

/* ---- block-level building blocks for the multi-GPU solver ------------------------------------ *
 * A distributed (block-column cyclic) A\b has no counterpart in the reference (it has no multi-device
 * code at all); its host driver (runmat_amd/sharding.py) needs in-place operations on
 * sub-blocks of provider buffers.  A view addresses rows [row_off, row_off+rows) x columns
 * [col_off, col_off+cols) of a 2-D buffer.  These calls MUTATE the viewed buffer (documented exception
 * to "every op returns a new buffer"; they are not part of the AccelProvider surface). */
typedef struct rmhip_view {
    rmhip_buf buf;
    size_t row_off, col_off, rows, cols;
} rmhip_view_t;
/* Copy a sub-block into a new contiguous rows x cols buffer / write a contiguous buffer into a sub-block. */
/* @serves - */
RMHIP_API int rmhip_blk_copy(rmhip_ctx* ctx, const rmhip_view_t* src, rmhip_buf* out);
/* @serves - */
RMHIP_API int rmhip_blk_assign(rmhip_ctx* ctx, const rmhip_view_t* dst, rmhip_buf src);
/* C = alpha*A*B + beta*C on views (fp64 MFMA dgemm). */
/* @serves - */
RMHIP_API int rmhip_blk_gemm(rmhip_ctx* ctx, double alpha, const rmhip_view_t* a, const rmhip_view_t* b,
                             double beta, const rmhip_view_t* c);
/* upper 0: B <- L^-1 B with L the unit-diagonal lower triangle of the square view `t`; 1: B <- U^-1 B, U its upper triangle with the
 * stored diagonal; 2: B <- B U^-1 (right-hand side: `b` is rows x w against a w x w triangle - the multipliers of a row block against a
 * factored diagonal tile, used by the row-partitioned multi-GPU solve). */
/* @serves - */
RMHIP_API int rmhip_blk_trsm(rmhip_ctx* ctx, int upper, const rmhip_view_t* t, const rmhip_view_t* b);
/* In-place LU (host_lu.rs pivot rule) of a tall view; `ipiv_out` receives a [min(rows,cols),1] tensor
 * of LAPACK-style interchange targets (row k swapped with row ipiv[k], zero-based, relative to the
 * view).  *info = number of pivots <= 1e-12. */
/* @serves - */
RMHIP_API int rmhip_blk_lu(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf* ipiv_out, int* info);
/* The same factorisation for a driver that has a guard of its own (rmhip_mldivide_row_partitioned): solve-path panel kernels
 * (pivoting inside the top blocks, multipliers recorded), NO host round trip - the interchange vector is written on the device and the
 * status is folded into `guard`, a 1 x 1 f64 tensor: NaN once a pivot hit the singular cut-off, else max(guard, largest multiplier).
 * The block is not saved: a value above the caller's bound means the caller's whole factorisation is refused.  Falls back to the
 * ordinary checks (and may then return RMHIP_ERR_GROWTH) when the solve-path kernels do not apply (RMHIP_LU_FAST=0). */
/* @serves - */
RMHIP_API int rmhip_blk_lu_deferred(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf guard, rmhip_buf* ipiv_out);
/* Apply those interchanges (in order) to every column of a view whose row 0 is the panel's row 0. */
/* @serves - */
RMHIP_API int rmhip_blk_swap_rows(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf ipiv);

/* `eye` / `eye_like` (lib.rs:1552-1559; simple_provider.rs:2293-2336, 3471-3486): ones on the leading diagonal of the first two
 * dimensions of every page, zeros elsewhere; shape [] is [1,1], [n] is [n,n] (normalize_shape, simple_provider.rs:779-788). */
/* @serves eye eye_like */
RMHIP_API int rmhip_eye(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out);
/* `flip` (lib.rs:2586; simple_provider.rs:1739-1776): reverse the zero-based `axes` (an axis named twice flips back; axes beyond
 * the rank are extent-1 dimensions).  Same shape. */
/* @serves flip */
RMHIP_API int rmhip_flip(rmhip_ctx* ctx, rmhip_buf a, const size_t* axes, size_t n_axes, rmhip_buf* out);
/* `circshift` (lib.rs:2589-2595; simple_provider.rs:2083-2143, 6431-6455): out(c) = in((c - shift) mod extent) per dimension;
 * `shifts` may be shorter than the rank (missing = 0) or longer (the shape is padded with 1s). */
/* @serves circshift */
RMHIP_API int rmhip_circshift(rmhip_ctx* ctx, rmhip_buf a, const long long* shifts, size_t n_shifts, rmhip_buf* out);
/* `tril` / `triu` (lib.rs:1635-1651; simple_provider.rs:1974-2081): zero the entries above (upper == 0: row - col < -offset) or below
 * (upper != 0: col - row < offset) the offset-th diagonal of the first two dimensions of every page. */
/* @serves tril triu */
RMHIP_API int rmhip_tri(rmhip_ctx* ctx, rmhip_buf a, int upper, long long offset, rmhip_buf* out);
/* `cat` (lib.rs:2686; shape rules backend/wgpu/provider/ops/tensor.rs:457-540): concatenate >= 2 tensors along the ONE-based `dim`;
 * every other extent must agree; the result drops trailing 1s down to max(dim, 2) dimensions (normalize_concat_shape). */
/* @serves cat */
RMHIP_API int rmhip_cat(rmhip_ctx* ctx, size_t dim_one_based, const rmhip_buf* inputs, size_t n_inputs, rmhip_buf* out);

/* max |a_ij| over a view (NaN if the view holds one): the multiplier guard of the row-partitioned solve. */
/* @serves - */
RMHIP_API int rmhip_blk_absmax(rmhip_ctx* ctx, const rmhip_view_t* a, double* out);

/* ---- multi-GPU collectives ------------------------------------------------------------------------- *
 * One process per GPU, one context per process.  The reference has no multi-device code; these entry
 * points are what the sharded forms of the hot path need from a host that is not Python: a row-block
 * all-gather for a replicated C = A*B, an ordered small-vector exchange for reductions / Monte-Carlo partial sums, and
 * the panel broadcast of the block-column-cyclic A\b.  Rendezvous is the host's business: rank 0 creates an id, the
 * host distributes its RMHIP_COMM_ID_BYTES bytes by any means (MPI, a file, torch.distributed), every rank calls
 * rmhip_comm_init with it.  Collectives must be issued in the same order on every rank; they are enqueued on the
 * context's stream like every other call (results are ordered behind them), except the asynchronous broadcast. */
#define RMHIP_COMM_ID_BYTES 128
enum rmhip_comm_transport {
    RMHIP_COMM_RCCL = 0,     /* one rank per GPU; RCCL over xGMI / PCIe (librccl is loaded on first use)            */
    RMHIP_COMM_HOST_SHM = 1  /* ranks of ONE node staged through POSIX shared memory: several ranks may share a GPU */
};
/* @serves - */
RMHIP_API int rmhip_comm_unique_id(int transport, void* id_out /* RMHIP_COMM_ID_BYTES */);
/* @serves - */
RMHIP_API int rmhip_comm_init(rmhip_ctx* ctx, const void* unique_id, int rank, int world);
/* @serves - */
RMHIP_API int rmhip_comm_destroy(rmhip_ctx* ctx);
/* A rank that cannot go on inside a sequence of collectives (a local allocation or device failure) calls this instead of leaving
 * its peers blocked: on the host shared-memory transport every rank's next (or current) barrier fails at once with RMHIP_ERR_HIP; on
 * RCCL only the LOCAL communicator is aborted (ncclCommAbort): RCCL does not release the peers by itself - a peer already inside a
 * collective keeps polling on the device.  What bounds their wait is rmhip_comm_wait_bounded below, which the library's own sequences
 * (rmhip_mldivide_row_partitioned's guard) call before the host blocks on a result of a collective.
 * Afterwards every collective on this context fails until rmhip_comm_destroy + rmhip_comm_init.  No communicator: no-op. */
/* @serves - */
RMHIP_API int rmhip_comm_abort(rmhip_ctx* ctx);
/* Host-side bounded wait for everything queued on the context's stream so far (collectives included): polls an event for at most
 * `timeout_s` seconds (<= 0: RMHIP_COMM_TIMEOUT_S, default 300) and the communicator's asynchronous error state; on expiry or error the
 * local communicator is aborted - which makes its device-side kernels exit - the stream is drained and RMHIP_ERR_HIP returned
 * ("comm: timed out ...").  With no communicator it is a plain stream synchronisation. */
/* @serves - */
RMHIP_API int rmhip_comm_wait_bounded(rmhip_ctx* ctx, double timeout_s);
/* rank 0 / world 1 when the context has no communicator */
/* @serves - */
RMHIP_API int rmhip_comm_rank(rmhip_ctx* ctx, int* rank, int* world);
/* @serves - */
RMHIP_API int rmhip_comm_barrier(rmhip_ctx* ctx);
/* In-place broadcast of a sub-block (or, with the full extent, of a whole buffer) from `root`.  async != 0: the
 * broadcast runs on the context's communication stream behind everything enqueued so far, and later calls do NOT wait
 * for it - the look-ahead of the block-cyclic solver posts the next panel's broadcast and keeps updating; call
 * rmhip_comm_wait before anything reads (or frees) the block.  Only dense blocks (whole columns) stay asynchronous;
 * a strided sub-block is packed through a staging buffer and the call stream waits for it. */
/* @serves - */
RMHIP_API int rmhip_comm_bcast(rmhip_ctx* ctx, const rmhip_view_t* block, int root, int async);
/* @serves - */
RMHIP_API int rmhip_comm_wait(rmhip_ctx* ctx);
/* Every rank contributes its k-element f64 vector `local`; *out is a new [k, world] buffer, column r = rank r's values,
 * identical on all ranks: the caller adds the columns in rank order, so sums do not depend on a reduction tree. */
/* @serves - */
RMHIP_API int rmhip_comm_allgather_f64(rmhip_ctx* ctx, rmhip_buf local, rmhip_buf* out);
/* Row-block all-gather of a column-major matrix: rank r holds rows [start_r, stop_r) x n of a rows_total x n matrix,
 * the balanced contiguous split of rows_total in units of `granule` (first ranks take the extra units, the last one the
 * ragged tail); *out is the replicated rows_total x n matrix. */
/* @serves - */
RMHIP_API int rmhip_comm_allgather_rows(rmhip_ctx* ctx, rmhip_buf local, size_t rows_total, size_t granule, rmhip_buf* out);

/* ---- sharded forms of the hot path behind the C ABI ---------------------------------------------- *
 * For hosts that are not Python (runmat_amd/sharding.py issues the same sequences from Python; tests/test_gpu_multirank.py
 * runs both side by side).  Every rank calls the same entry point with its local block, after rmhip_comm_init.
 *
 * rmhip_matmul_row_sharded: C[rows_g, :] = A[rows_g, :] * B with B replicated - the embarrassingly parallel form, no exchange.
 * gather != 0 appends the row-block all-gather (rmhip_comm_allgather_rows: balanced split of rows_total in units of `granule`,
 * 0 = 128) and returns the replicated rows_total x n product; gather == 0 returns this rank's rows. */
/* @serves - */
RMHIP_API int rmhip_matmul_row_sharded(rmhip_ctx* ctx, rmhip_buf a_rows, rmhip_buf b, size_t rows_total, size_t granule, int gather,
                                       rmhip_buf* out);
/* rmhip_mldivide_row_partitioned: x = A \ b with [A | b] (n x (n + nrhs)) distributed BY ROWS - row block q of height rb lives on rank
 * q % world, blocks in ownership order in `ab_local`, which is OVERWRITTEN with this rank's rows of the factors.  *out is the
 * replicated n x nrhs solution, identical on every rank.  Pivots never leave a solve, so - as on one GPU (lu.hip, solve path) -
 * pivoting is restricted to a diagonal domain and verified: panel p is factored by its owner with partial pivoting among the owner's
 * OWN rows from the diagonal tile down, one broadcast per panel carries the owner's tile row [L11\U11 | U12 | y], every rank forms
 * its rows' multipliers and trailing update without further exchange, the last `world` row blocks are gathered and finished by every
 * rank with the single-GPU solve, the back substitution is redundant.  Depth-1 look-ahead: the owner of panel p + 1 updates that
 * panel first, factors it and posts its broadcast on the communication stream under the rest of update p.  Guard: the largest
 * multiplier outside the owners' domains, one exchange at the end; beyond `tau` (8 is the single-GPU default) - or when ANY rank
 * failed (a singular pivot inside its domain, an allocation): the failing rank keeps taking part in every collective with NaN-poisoned
 * data, so no rank is left blocked - every rank returns RMHIP_ERR_GROWTH and the host falls back to the block-column form.  Needs a
 * precision-64 provider. */
/* @serves - */
RMHIP_API int rmhip_mldivide_row_partitioned(rmhip_ctx* ctx, rmhip_buf ab_local, size_t n, size_t nrhs, size_t rb, double tau,
                                             rmhip_buf* out);
/* Device time of the last rmhip_mldivide_row_partitioned call on this context by phase, in milliseconds (timed events on the streams the
 * phases ran on): out4 = { panel (the owner's factorisation, interchanges, U12, tile copy), wait (this rank's stream idling for a tile
 * broadcast), update (multipliers + trailing updates; with the overlap on they run beside the panels, so the sum may exceed the wall
 * clock), exchange (the multiplier guard's gather, the gathered tail and its replicated solve, the back substitution) }.
 * RMHIP_RP_TIMERS=0 turns the events off (zeros). */
/* @serves - */
RMHIP_API int rmhip_rp_phase_ms(rmhip_ctx* ctx, double* out4);

/* ---- RNG  (lib.rs:1713-1728, 1772) ---------------------------------------------------------- */

/* `set_rng_state`: raw 64-bit LCG state (random.rs:7-13). rmhip_rng_seed applies mix_seed
 * (random.rs:128-141) like `rng(seed)`. */
/* @serves set_rng_state */
RMHIP_API int rmhip_set_rng_state(rmhip_ctx* ctx, uint64_t state);
/* @serves - */
RMHIP_API int rmhip_get_rng_state(rmhip_ctx* ctx, uint64_t* state);
/* @serves - */
RMHIP_API int rmhip_rng_seed(rmhip_ctx* ctx, uint64_t seed);
/* `random_uniform` / `random_normal`: CPU-parity stream (64-bit LCG + Box-Muller pairs,
 * random.rs:271-288,530-543); the state advances exactly as the CPU generator's does. */
/* @serves random_uniform random_uniform_like */
RMHIP_API int rmhip_random_uniform(rmhip_ctx* ctx, const size_t* shape, size_t rank,
                                   rmhip_buf* out);
/* @serves random_normal random_normal_like */
RMHIP_API int rmhip_random_normal(rmhip_ctx* ctx, const size_t* shape, size_t rank,
                                  rmhip_buf* out);
/* Lazy `random_normal` (f64 contexts, on by default from 1024 elements; RMHIP_LAZY_RANDN=0 / RMHIP_LAZY_RANDN_MIN=<n> in the
 * environment): rmhip_random_normal returns a handle with NO storage - the stream advances as usual - and a streaming
 * rmhip_fused_elementwise kernel that reads it generates the normals in registers, bit for bit the values an eager call writes
 * (8 B per sample neither written nor read back: the Monte-Carlo step `S .* exp(drift + scale .* randn(M, 1))` moves 16 B per sample
 * instead of 32).  Every other consumer (download, per-op calls, reductions, views, device pointers) first materialises the tensor
 * under the same id from the recorded stream position; the two forms are indistinguishable through the API except by
 * rmhip_lazy_random_stats.  `min_numel` == 0 keeps the current threshold. */
/* @serves - */
RMHIP_API int rmhip_set_lazy_random(rmhip_ctx* ctx, int enabled, size_t min_numel);
/* counts since rmhip_init: lazy handles created, consumed in registers by a fused kernel (per use), materialised */
/* @serves - */
RMHIP_API int rmhip_lazy_random_stats(rmhip_ctx* ctx, uint64_t* created, uint64_t* fused, uint64_t* materialised);
/* Scaled / transformed draws of the same stream (lib.rs:1732-1757, 1820-1839; CPU forms random.rs:290-320, 514-528; the in-process
 * provider simple_provider.rs:3560-3626, 3683-3725), one draw per element in column-major order:
 *   unifrnd:        a + (b - a) * u, the difference rounded once, then one multiply and one add (bit-exact)
 *   exponential:    -mu * ln(max(u, f64::MIN_POSITIVE))   (the logarithm within 2 ulp of the exact value; against the CPU's libm chain 8e-16 relative)
 *   normrnd:        mu + sigma * z over whole Box-Muller pairs, the stream of rmhip_random_normal (an odd count drops the last z1)
 *   integer_range:  lower + min(floor(u * span), span - 1), span = upper - lower + 1 (bit-exact); lower > upper or span > 2^53 is
 *                   RMHIP_ERR_INVALID; span == 1 fills `lower` and consumes no draws.
 * The state advances by the draws consumed, exactly as the CPU generator's. */
/* @serves random_unifrnd */
RMHIP_API int rmhip_random_unifrnd(rmhip_ctx* ctx, double a, double b, const size_t* shape, size_t rank, rmhip_buf* out);
/* @serves random_exponential */
RMHIP_API int rmhip_random_exponential(rmhip_ctx* ctx, double mu, const size_t* shape, size_t rank, rmhip_buf* out);
/* @serves random_normrnd */
RMHIP_API int rmhip_random_normrnd(rmhip_ctx* ctx, double mu, double sigma, const size_t* shape, size_t rank, rmhip_buf* out);
/* @serves random_integer_range random_integer_like */
RMHIP_API int rmhip_random_integer_range(rmhip_ctx* ctx, long long lower, long long upper, const size_t* shape, size_t rank,
                                         rmhip_buf* out);
/* `stochastic_evolution` (lib.rs:1759-1769; CPU loop builtins/stats/random/stochastic_evolution.rs:10-30):
 * `steps` times { z = randn(size(state)) from the shared stream; state .*= exp(drift + scale .* z) }, as one
 * kernel that keeps the state in registers.  Advances the RNG state exactly as the CPU loop does
 * (steps * 2 * ceil(numel / 2) draws).  steps == 0 returns a copy. */
/* @serves stochastic_evolution */
RMHIP_API int rmhip_stochastic_evolution(rmhip_ctx* ctx, rmhip_buf state, double drift, double scale, uint32_t steps,
                                         rmhip_buf* out);
/* Multi-GPU form (no counterpart in the reference, which has no multi-device code): `state` is the
 * shard [offset, offset + numel) of a global vector whose every step draws `draws_per_step` values
 * (2 * ceil(global_numel / 2)); the caller positions the RNG at global_state + offset beforehand
 * (rmhip_set_rng_state) and the shard then consumes exactly the normals the single-device run would give
 * those elements.  The RNG state advances by steps * draws_per_step.  draws_per_step == 0 is the plain call. */
/* @serves - */
RMHIP_API int rmhip_stochastic_evolution_sharded(rmhip_ctx* ctx, rmhip_buf state, double drift, double scale,
                                                 uint32_t steps, uint64_t draws_per_step, rmhip_buf* out);

/* ---- telemetry  (lib.rs:1337-1376, 3023-3045) ----------------------------------------------- */

typedef struct rmhip_telemetry {
    uint64_t fused_elementwise_count, fused_elementwise_ns;
    uint64_t fused_reduction_count, fused_reduction_ns;
    uint64_t matmul_count, matmul_ns;
    uint64_t mldivide_count, mldivide_ns;
    uint64_t upload_bytes, download_bytes;
    uint64_t fusion_cache_hits, fusion_cache_misses;
    uint64_t kernel_launches;
    uint64_t bytes_allocated, bytes_pooled;
    uint64_t linsolve_count, linsolve_ns;   /* ProviderTelemetry::linsolve / mrdivide (lib.rs:1342-1344) */
    uint64_t mrdivide_count, mrdivide_ns;
} rmhip_telemetry_t;
/* @serves telemetry_snapshot fused_cache_counters */
RMHIP_API int rmhip_telemetry(rmhip_ctx* ctx, rmhip_telemetry_t* out);
/* @serves reset_telemetry */
RMHIP_API int rmhip_reset_telemetry(rmhip_ctx* ctx);
/* `ProviderTelemetry::solve_fallbacks` (lib.rs:1347, `ProviderFallbackStat` :1331-1335): one (reason, count) pair per
 * distinct reason a solve was handed back to the caller's CPU path ("mldivide:unsupported", "mldivide:singular",
 * "linsolve:unsupported", ...).  index >= the number of reasons returns RMHIP_ERR_NOT_FOUND. */
/* @serves telemetry_snapshot */
RMHIP_API int rmhip_telemetry_solve_fallback(rmhip_ctx* ctx, size_t index, char* reason, size_t cap, uint64_t* count);
/* `ProviderTelemetry::kernel_launches` (lib.rs:1355-1356, `KernelLaunchTelemetry` :1372-1378): bounded log of recent
 * dispatches, oldest first (index 0); same kernel names and attribute keys as the reference's wgpu provider records
 * (ops/telemetry.rs:26-34,140-146, helpers.rs:36-50): "fused_elementwise" {len, inputs, rank}, "fused_elementwise_multi"
 * {.., num_outputs}, "fused_reduction" {reduce_len, slices, rank} / {wg, flavor}, "matmul" {m, n, k}. */
typedef struct rmhip_kernel_attr {
    char key[16];
    uint64_t value;
} rmhip_kernel_attr_t;
typedef struct rmhip_kernel_launch {
    char kernel[48];
    char precision[8];  /* "f64" / "f32" */
    uint32_t n_shape, n_tuning;
    rmhip_kernel_attr_t shape[6];
    rmhip_kernel_attr_t tuning[6];
} rmhip_kernel_launch_t;
/* @serves telemetry_snapshot */
RMHIP_API int rmhip_telemetry_kernel_launch(rmhip_ctx* ctx, size_t index, rmhip_kernel_launch_t* out);

/* Counters of the LU / solve machinery (no counterpart in the reference, whose GPU solve is a host round trip,
 * backend/wgpu/provider/ops/solve.rs:144-168).  They are what telemetry.solve_fallbacks cannot say: a solve that was
 * answered on the device but not on its first attempt. */
typedef struct rmhip_lu_stats {
    uint64_t solve_path_factorizations; /* mldivide / linsolve / mrdivide factorisations accepted with pivoting restricted to the panels' top blocks */
    uint64_t pivot_growth_fallbacks;    /* ... refactored with the grid-wide pivot rule: a multiplier exceeded tau (also in solve_fallbacks as "lu:pivot_growth") */
    uint64_t panel_exchange_timeouts;   /* persistent panel: a bounded spin expired (workgroups not co-resident), matrix refactored on the next, more conservative path */
    uint64_t subst_chain_timeouts;      /* one-launch substitution timed out, repeated as one launch per block */
    double last_max_multiplier;         /* largest |l| below a top block in the last solve-path factorisation */
    double tau;                         /* the bound it is checked against (RMHIP_LU_TAU, default 8) */
    int one_xcd_panels;                 /* 1 while panels of <= 32 workgroups are placed on one XCD */
    int conservative_panels;            /* 1 once the context fell back to one launch per column */
    uint64_t svd_solves;                /* systems the LU / Gram paths refused (singular, rank deficient, ill conditioned) that were answered by the
                                           Jacobi-SVD path with the reference's tolerance rule (min(rows, cols) <= 4096) */
} rmhip_lu_stats_t;
/* @serves - */
RMHIP_API int rmhip_lu_stats(rmhip_ctx* ctx, rmhip_lu_stats_t* out);

/* HIP-event timing on the context stream (for bench.py's roofline leg): begin records an event,
 * end records another, synchronizes and returns the elapsed milliseconds between them. */
/* @serves - */
RMHIP_API int rmhip_timer_begin(rmhip_ctx* ctx);
/* @serves - */
RMHIP_API int rmhip_timer_end(rmhip_ctx* ctx, double* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* RMHIP_H */
