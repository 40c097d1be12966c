// common.h -- internal declarations shared by the librmhip.so translation units.
// Context = one GPU: HIP stream, buffer table (buffer_id -> device allocation + shape, the
// provider-side half of `GpuTensorHandle`, crates/runmat-accelerate-api/src/lib.rs:260-264),
// a size-bucketed device-memory pool, the fused-kernel cache and telemetry counters.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rmhip.h"

namespace rmhip {

// ---- errors ------------------------------------------------------------------------------------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define RMHIP_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::rmhip::fail(RMHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

// RMHIP_TRACE=1: step-by-step host trace on stderr (debugging aid)
#define RMHIP_TRACEF(...)                                  \
    do {                                                   \
        static const bool _on = std::getenv("RMHIP_TRACE") != nullptr; \
        if (_on) {                                         \
            std::fprintf(stderr, "[rmhip] " __VA_ARGS__);  \
            std::fputc('\n', stderr);                      \
        }                                                  \
    } while (0)

#define RMHIP_TRY(expr)              \
    do {                             \
        int _rc = (expr);            \
        if (_rc != RMHIP_OK) return _rc; \
    } while (0)

// ---- device allocations (reference counted so `reshape` can alias storage) -----------------------
struct Context;
struct Allocation {
    Context* ctx = nullptr;
    double* ptr = nullptr;
    size_t bytes = 0;     // bucket size actually reserved
    bool external = false;  // adopted via rmhip_wrap_external: never freed by us
    ~Allocation();
};

enum : uint8_t { DT_F64 = 0, DT_F32 = 1 };

struct Buffer {
    std::shared_ptr<Allocation> alloc;
    std::vector<size_t> shape;
    size_t numel = 0;
    // Lazy transpose view (`transpose`, lib.rs:2532; the reference's wgpu provider does the same with
    // `record_handle_transpose`, ops/tensor.rs:828-846): `shape` is the LOGICAL [R, C] but the storage still holds
    // the base matrix C x R (column-major, leading dimension C).  matmul / syrk consume views in place through the
    // transposed-operand dgemm variants; every other consumer sees a materialised copy (Context::get).
    bool tview = false;
    // Storage type.  A context created with 32-bit precision (`ProviderPrecision::F32`, lib.rs:815-818) keeps its
    // tensors as f32 in HBM; arithmetic stays f64 in registers (the CPU path computes `single` arrays in f64 and
    // rounds the result, runmat-builtins lib.rs:426-436), so only loads and stores differ.
    uint8_t dtype = DT_F64;
    // Lazy repmat view (`repmat`, lib.rs:2689-2695; tiling rule simple_provider.rs:2174-2240): `shape` is the LOGICAL tiled
    // shape, the storage still holds the base tensor whose extents (padded with 1s to shape.size()) are `rep_base`;
    // element (c0, c1, ...) of the view is base element (c0 % rep_base[0], c1 % rep_base[1], ...).  rmhip_binary and
    // rmhip_fused_elementwise read views in place with stride-0 indexing - the reference's plus/minus/times/rdivide/power
    // callers expand an operand with `repmat` only to hand it to `elem_*` and free it (times.rs:501-543) - every other
    // consumer sees a materialised copy on first use (Context::get / get_view).  Never combined with `tview`.
    std::vector<size_t> rep_base;
    // Complex storage (`GpuTensorStorage::ComplexInterleaved`, lib.rs:247-251): `shape` / `numel` are the LOGICAL complex extents, the
    // storage holds 2 * numel doubles (re, im, re, im, ...) and is always f64 (a precision-32 context rounds the VALUES through f32).
    // Only the transforms and the complex constructors (fft.hip) and upload-free plumbing (download, shape, free) accept such a
    // buffer; every real-valued entry point refuses it (Context::get_raw), so the caller gathers - as it does for any `Err`.
    bool cplx = false;
    // Lazy `random_normal` (f64 contexts): no storage yet - element 2g / 2g + 1 is the Box-Muller pair drawn from the stream `rng_state`
    // advanced by 2g + 1 / 2g + 2 steps (rng.hip k_rng_normal).  rmhip_fused_elementwise's streaming kernel generates the values in
    // registers (codegen.cpp, skel_rng.h: bit for bit what k_rng_normal writes); every other consumer - anything that goes through
    // Context::get_raw - sees the tensor materialised under the same id first.  Not part of lazy(): it shares no storage with anyone.
    bool rng_lazy = false;
    uint64_t rng_state = 0;
    bool lazy() const { return tview || !rep_base.empty(); }
    size_t stored_numel() const {  // elements the storage holds (the base of a repmat view)
        if (rep_base.empty()) return numel;
        size_t n = 1;
        for (size_t e : rep_base) n *= e;
        return n;
    }
    double* data() const { return alloc ? alloc->ptr : nullptr; }
    float* data_f32() const { return alloc ? reinterpret_cast<float*>(alloc->ptr) : nullptr; }
};

struct Telemetry {
    std::atomic<uint64_t> fused_elementwise_count{0}, fused_elementwise_ns{0};
    std::atomic<uint64_t> fused_reduction_count{0}, fused_reduction_ns{0};
    std::atomic<uint64_t> matmul_count{0}, matmul_ns{0};
    std::atomic<uint64_t> mldivide_count{0}, mldivide_ns{0};
    std::atomic<uint64_t> upload_bytes{0}, download_bytes{0};
    std::atomic<uint64_t> cache_hits{0}, cache_misses{0};
    std::atomic<uint64_t> kernel_launches{0};
    std::atomic<uint64_t> bytes_allocated{0}, bytes_pooled{0};
    std::atomic<uint64_t> linsolve_count{0}, linsolve_ns{0};
    std::atomic<uint64_t> mrdivide_count{0}, mrdivide_ns{0};
};

// `KernelLaunchTelemetry` (lib.rs:1372-1378): names and keys are string literals, recording is a handful of stores
struct LaunchRecord {
    const char* kernel = nullptr;
    int bits = 64;
    int n_shape = 0, n_tuning = 0;
    const char* shape_key[6];
    uint64_t shape_val[6];
    const char* tuning_key[6];
    uint64_t tuning_val[6];
};
static constexpr int kLaunchLog = 64;  // bounded log, newest overwrites oldest

struct FusedKernel;  // codegen.h
struct Comm;         // comm.cpp

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipDeviceProp_t props{};
    int num_cus = 256;

    std::mutex mu;  // guards table, pool, kernel cache
    // One C-ABI call at a time per context: the provider trait is `Send + Sync` (lib.rs:1386) and calls may arrive from
    // several host threads, but scratch, rng_state, narrow_pending and the stream retargeting of the look-ahead LU are
    // per-context state.  Taken (recursively: entry points call entry points) by CTX_OR_FAIL; uncontended cost ~20 ns.
    std::recursive_mutex call_mu;
    // kernels whose dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) was set on THIS context's device
    std::vector<const void*> lds_opt_in;
    std::unordered_map<uint64_t, Buffer> table;
    size_t n_lazy = 0;  // records in `table` with lazy() set (guarded by `mu`): lets detach_views_of return at once when there are none
    uint64_t next_id = 1;

    // pool: bucket bytes -> free device pointers
    std::multimap<size_t, double*> pool;
    size_t pooled_bytes = 0;
    size_t pool_limit_bytes = 0;  // set at init (fraction of HBM)

    std::unordered_map<uint64_t, std::shared_ptr<FusedKernel>> kernel_cache;
    std::unordered_map<uint64_t, std::shared_ptr<Allocation>> fft_tables;  // twiddle / chirp tables by (kind, length) (fft.hip)

    uint64_t rng_state = 0x9e3779b97f4a7c15ULL;  // DEFAULT_RNG_SEED, random.rs:7
    // random_normal returns lazy records (Buffer::rng_lazy) from `lazy_randn_min` elements on (f64 contexts); RMHIP_LAZY_RANDN=0 or
    // rmhip_set_lazy_random(ctx, 0, 0) turn it off
    bool lazy_randn = true;
    size_t lazy_randn_min = 1024;
    uint64_t lazy_randn_created = 0, lazy_randn_fused = 0, lazy_randn_materialised = 0;  // rmhip_lazy_random_stats
    size_t n_rng_lazy = 0;  // lazy random_normal records alive (guarded by `mu`): rmhip_fused_elementwise skips its look for them when there are none

    // scratch for reductions / LU (grown on demand, reused)
    double* scratch = nullptr;
    size_t scratch_bytes = 0;

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    Telemetry tel;
    LaunchRecord launch_log[kLaunchLog];
    uint64_t launch_seq = 0;                                       // records written so far
    std::vector<std::pair<const char*, uint64_t>> solve_fallbacks;  // reason (literal) -> count
    void record_launch(const char* kernel, std::initializer_list<std::pair<const char*, uint64_t>> shape,
                       std::initializer_list<std::pair<const char*, uint64_t>> tuning);
    void record_solve_fallback(const char* reason);

    // LU look-ahead (lu.hip getrf_blocked): extra dynamic LDS requested by launch_dgemm so that only ONE
    // dgemm block fits per CU and latency-bound kernels of the other stream find room beside it
    size_t gemm_lds_pad = 0;
    // look-ahead LU, late phase: the update stream's dgemm runs as a persistent kernel that stays off the panels' XCD
    unsigned* gemm_tile_counters = nullptr;  // device, zeroed by the driver; one per launch
    size_t gemm_counter_next = 0, gemm_counter_cap = 0;
    const int* gemm_avoid_xcc = nullptr;     // device word written by the panel kernel (-1: none)
    // rmhip_blk_lu factors with the solve path's panel kernels (pivoting inside each base panel's top block, multiplier bound checked;
    // a violation restores the block and factors it with the grid-wide rule): set by the row-partitioned multi-GPU solve around its
    // panel factorisations, whose pivots never leave the provider
    bool blk_lu_solve_path = false;
    // solve-path LU: inverses of the 16 x 16 unit-lower diagonal blocks of L (k_rp_top leaves them: block q = columns 16 q .. 16 q + 15,
    // 256 doubles, [k][i] = inv(L_qq)[i][k]) for the matrix-core triangular solve; lu_work / lu_work_ld locate a T operand's diagonal
    const double* lu_linv = nullptr;
    std::vector<unsigned char>* lu_linv_ok = nullptr;  // host: block q has its inverse queued (a base panel of another width leaves none)
    const double* lu_work = nullptr;
    size_t lu_work_ld = 0;
    bool gemm_chain_prio = false;  // the look-ahead LU's main-stream dgemm launches raise their wave priority (RMHIP_LU_GEMM_PRIO=0 disables)
    unsigned* gemm_announce_tab = nullptr;  // the table itself while a two-level factorisation runs (k_trsm_lower_mfma on the main stream)
    bool lu_yield_trsm = false;
    unsigned* gemm_announce = nullptr;  // two-level LU: the yield table for the main stream's dgemm blocks to count themselves into (RMHIP_LU_YIELD_ALL)
    unsigned* ext_yield_tab = nullptr;  // a yield table (kYieldSlots counters) owned by a driver outside lu.hip for the panels it factors (sharded.cpp), or nullptr
    const unsigned* gemm_yield_word = nullptr;  // two-level LU: the CU (key) whose update blocks pause while k_rp_top runs there (device word; 0: none)
    double* gemm_split_ws = nullptr;  // caller-owned workspace for split-K partial products (per stream; see launch_dgemm)
    size_t gemm_split_ws_elems = 0;
    size_t gemm_split_min_k = 0;  // != 0: launch_dgemm splits the inner dimension of few-tile products from this k on (the LU's products with inverted L11 blocks)
    double lu_last_minv = 0.0;    // largest |entry| of the inverted L11 blocks of the last solve-path factorisation (0: none were formed)
    bool in_lookahead = false;  // inside the LU's look-ahead driver: main-stream dgemm blocks must fit beside the update stream's
    // set after a persistent-panel factorisation found its workgroups not co-resident (device shared with
    // another context): from then on LU uses the one-launch-per-column panels on a single stream
    double rp_phase_ms[4] = {0, 0, 0, 0};  // rmhip_rp_phase_ms: panel / broadcast wait / update / exchange device time of the last row-partitioned solve
    hipStream_t lu_side_stream = nullptr;  // update stream of the look-ahead LU (low priority), created on first use
    hipStream_t lu_aux_stream = nullptr;   // solve path: the full-height kernels' rows below the band of the panel in flight (lu.hip, LuState::aux)
    hipStream_t lu_prep_stream = nullptr;  // interchanges + triangular solves of one half of the trailing columns under the other half's dgemm
    hipStream_t lu_far_stream = nullptr;   // two-level driver (solve path): the deep rank-W updates of the columns beyond the next super-panel
    hipStream_t lu_mid_stream = nullptr;   // two-level driver: the updates inside the super-panel in flight (normal priority)
    std::vector<hipEvent_t> lu_events;     // its event pool
    bool lu_conservative = false;
    bool subst_chain_failed = false;  // the one-launch substitution timed out once: keep the launch-per-block form
    bool one_xcd_ok = true;  // LU panels may place their blocks on one XCD (cleared when such a panel timed out once)
    bool lu_used_one_xcd = false;
    bool lu_last_fast = false;  // the last lu_factor_device call restricted pivoting to the panels' top blocks (mode 1 taken, not merely asked for)
    bool solve_strict = false;  // a refused square solve stays refused (inv): no SVD answer for a singular matrix
    double lu_last_growth = 0.0;       // largest multiplier the last solve-path factorisation saw below its top blocks
    uint64_t lu_fast_count = 0, lu_growth_fallbacks = 0;  // solve-path factorisations accepted / refactored with the grid-wide rule
    uint64_t lu_exchange_timeouts = 0, lu_subst_timeouts = 0;
    uint64_t svd_solves = 0;  // systems answered by the Jacobi-SVD path (svdsolve.hip)
    double lu_tau = 8.0;
    int num_xcc = 8;  // accelerator dies the dispatcher interleaves workgroups over (probed at init; 1 on a CPX partition)
    // 64 or 32 (rmhip_set_precision).  At 32 every op output is stored as f32: kernels with a native f32-storage variant
    // (fused elementwise / reduction, per-op elementwise, reductions, dot) read and write f32 directly, every other op
    // runs its f64 kernel on widened temporaries and the entry point narrows what it created on return (NarrowScope).
    int precision = 64;
    std::vector<uint64_t> narrow_pending;  // buffers created by new_buffer since the enclosing entry point began
    int trsm_base = 128;  // base width of the triangular-solve recursion (64 on the main stream under LU look-ahead)
    Comm* comm = nullptr;  // communicator of the multi-GPU entry points (rmhip_comm_init), owned by the context

    // ---- helpers (rmhip_core.cpp) ----
    int alloc_device(size_t numel, std::shared_ptr<Allocation>* out);
    void release_device(double* ptr, size_t bytes);
    int new_buffer(const size_t* shape, size_t rank, uint64_t* id, Buffer* out);
    int register_buffer(Buffer&& b, uint64_t* id);
    int new_buffer_complex(const size_t* shape, size_t rank, uint64_t* id, Buffer* out = nullptr);
    int lookup(uint64_t id, Buffer* out);   // the table entry as it is (plumbing: shape, storage, download)
    int get_any(uint64_t id, Buffer* out);  // like get(), but a complex buffer is handed over as it is
    int new_buffer_f32(const size_t* shape, size_t rank, uint64_t* id, Buffer* out);  // f32 storage, never narrowed
    int get(uint64_t id, Buffer* out);       // f64 data, plain layout: widens f32 storage into a temporary, materialises a transpose view
    int get_view(uint64_t id, Buffer* out);  // f64 data, `tview` may be set (matmul / syrk read views in place); repmat views are materialised
    // the record as stored: dtype may be DT_F32, `tview` / `rep_base` may be set.  A lazy random_normal record is materialised first
    // unless `keep_rng` (only the fused elementwise entry point consumes one as it is)
    int get_raw(uint64_t id, Buffer* out, bool keep_rng = false);
    int settle_rng(uint64_t id);             // materialise a lazy random_normal record under its id (k_rng_normal on its recorded state)
    int settle_view(uint64_t id);            // materialise a transpose / repmat view in its own storage type and keep it under this id
    // Before an IN-PLACE write to buffer `id` (scatter_linear, the block views' assign / gemm / trsm / lu / swap_rows, the epilogue's
    // diagonal output, a raw device pointer handed out): every OTHER handle that is a lazy view of the same storage is materialised
    // first, so that it keeps the values it was created from (the reference's repmat / transpose results are buffers of their own).
    int detach_views_of(uint64_t id);
    int narrow(uint64_t id);                 // replace an f64 buffer's storage by its f32 rounding
    void finish_outputs(size_t mark);        // narrow everything new_buffer created since `mark` (precision 32 only)
    int ensure_scratch(size_t bytes);
    void ensure_max_lds(const void* kernel, size_t bytes);  // once per kernel per context (the attribute is per device)
};

inline size_t shape_numel(const size_t* shape, size_t rank) {
    size_t n = 1;
    for (size_t i = 0; i < rank; ++i) n *= shape[i];
    return n;
}

// RAII guard selecting the context's device for the calling thread.
struct DeviceGuard {
    explicit DeviceGuard(const Context* c) { (void)hipSetDevice(c->device); }
};

// Declared first in every C-ABI entry point (CTX_OR_FAIL): on return, outputs the entry point created through
// new_buffer are rounded to f32 storage when the context runs at 32-bit precision.  Nested entry points narrow their
// own outputs, which also reproduces the CPU path's rounding after every builtin.
struct NarrowScope {
    Context* c;
    size_t mark;
    explicit NarrowScope(Context* ctx) : c(ctx), mark(ctx->narrow_pending.size()) {}
    ~NarrowScope() {
        if (c->precision == 32) c->finish_outputs(mark);
        else c->narrow_pending.resize(mark);
    }
};

struct ScopedTimer {
    std::atomic<uint64_t>* count;
    std::atomic<uint64_t>* ns;
    uint64_t t0;
    ScopedTimer(std::atomic<uint64_t>* c, std::atomic<uint64_t>* n);
    ~ScopedTimer();
};

// ---- kernel launchers implemented in the .hip translation units -------------------------------
// elementwise (ew_kernels.hip)
int launch_widen(Context* c, const float* src, double* dst, size_t n);
int launch_narrow(Context* c, const double* src, float* dst, size_t n);
// f32-storage variants: same arithmetic (f64 in registers), f32 loads and stores
int launch_unary_f32(Context* c, int op, const float* a, float* out, size_t n);
int launch_scalar_f32(Context* c, int op, const float* a, double s, float* out, size_t n);
int launch_binary_same_f32(Context* c, int op, const float* a, const float* b, float* out, size_t n);
int launch_fill(Context* c, double* dst, size_t n, double value);
int launch_fill_uniform(Context* c, double* dst, size_t n, uint64_t seed, double lo, double hi);
int launch_unary(Context* c, int op, const double* a, double* out, size_t n);
int launch_scalar(Context* c, int op, const double* a, double s, double* out, size_t n);
struct BroadcastDesc {  // collapsed, front-padded; dim 0 fastest. rank <= 8.
    int rank;
    uint64_t out_shape[8];
    uint64_t stride_a[8];
    uint64_t stride_b[8];
};
int launch_binary_same(Context* c, int op, const double* a, const double* b, double* out, size_t n);
int launch_binary_bcast(Context* c, int op, const double* a, const double* b, double* out,
                        size_t n, const BroadcastDesc& d);
int launch_binary_bcast_f32(Context* c, int op, const float* a, const float* b, float* out, size_t n,
                            const BroadcastDesc& d);

// tensor_ops.hip: out[idx] = src[sum_d map_d(c_d) * stride[d]] with (c_0, c_1, ...) the column-major coordinates of idx in
// `shape`; map_d(c) = (off[d] + c) % mod[d], or off[d] - c when rev[d] (flip).  One kernel family behind repmat
// (mod = base extent), permute (permuted strides), circshift (off) and flip.  rank <= 8.
struct IndexMap {
    int rank = 0;
    uint64_t shape[8], stride[8], off[8], mod[8];
    uint8_t rev[8];
};
int launch_index_copy(Context* c, const double* src, double* dst, size_t n, const IndexMap& m);
int launch_index_copy_f32(Context* c, const float* src, float* dst, size_t n, const IndexMap& m);
int materialize_repmat(Context* c, const Buffer& view, void* dst);  // tile a repmat view's base into dst (the view's storage type)

// reductions (reduce_kernels.hip)
int launch_reduce_all(Context* c, int op, int nan_mode, const double* x, size_t n, double* out);
// x viewed as [pre, red, post] column-major; reduces the middle extent. out has pre*post elements.
int launch_reduce_mid(Context* c, int op, int nan_mode, const double* x, size_t pre, size_t red,
                      size_t post, double* out);

// f32 storage in, f64 accumulation and f64 result (the caller narrows it)
int launch_reduce_mid_f32(Context* c, int op, int nan_mode, const float* x, size_t pre, size_t red, size_t post, double* out);
int launch_reduce_dot_f32(Context* c, const float* a, const float* b, size_t pre, size_t red, size_t post, double* out);

// reduce2.hip: arg-min / arg-max with indices, std, truth counts, cumulative scans over the [pre, red, post] view
int launch_argreduce(Context* c, int op, int nan_mode, const double* x, size_t pre, size_t red, size_t post, double* values, double* indices);
int launch_reduce_std(Context* c, int population, int nan_mode, const double* x, size_t pre, size_t red, size_t post, double* out);
int launch_reduce_std_f32(Context* c, int population, int nan_mode, const float* x, size_t pre, size_t red, size_t post, double* out);
int launch_argreduce_f32(Context* c, int op, int nan_mode, const float* x, size_t pre, size_t red, size_t post, double* values, double* indices);
int launch_reduce_truth_f32(Context* c, int op, int omit_nan, const float* x, size_t pre, size_t red, size_t post, double* out);
int launch_reduce_moments(Context* c, const double* x, size_t pre, size_t red, size_t post, double* mean, double* ex2);
int launch_plane_stats(Context* c, const double* x, size_t batch, size_t plane, double eps, double* stats);  // image_normalize, batch > 256
int launch_reduce_truth(Context* c, int op, int omit_nan, const double* x, size_t pre, size_t red, size_t post, double* out);
int launch_cumulative(Context* c, int prod, int reverse, int omit, const double* x, size_t pre, size_t len, size_t post, double* y);

// sum(a .* b) over the middle extent of [pre, red, post]
int launch_reduce_dot(Context* c, const double* a, const double* b, size_t pre, size_t red, size_t post, double* out);

// dgemm (dgemm.hip): C[m x n] = alpha * A[m x k] * B[k x n] + beta * C, column-major with leading
// dimensions. beta == 0 ignores C's previous contents.
int launch_dgemm(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                 const double* B, size_t ldb, double beta, double* C, size_t ldc);
// C = epilogue(A*B): MatmulEpilogue folded into the store (lib.rs:3498-3560).
struct GemmEpilogue {
    int flags;  // EP_* bits
    double alpha, beta_add;
    const double* row_scale;
    const double* col_scale;
    double clamp_min, clamp_max, pow_exp;
    double* diag;
};
enum { EP_ACTIVE = 1, EP_ROW = 2, EP_ROW_DIV = 4, EP_COL = 8, EP_COL_DIV = 16, EP_CLAMP_MIN = 32, EP_CLAMP_MAX = 64,
       EP_POW = 128, EP_DIAG = 256 };
int launch_dgemm_trans(Context* c, bool ta, bool tb, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                       const double* B, size_t ldb, double beta, double* C, size_t ldc);
// f32 GEMM for precision-32 providers (sgemm.hip): C = op(A) * op(B), f32 accumulation in the matrix cores
int launch_sgemm_trans(Context* c, bool ta, bool tb, size_t m, size_t n, size_t k, const float* A, size_t lda, const float* B,
                       size_t ldb, float* C, size_t ldc);
int launch_dgemm_epilogue(Context* c, size_t m, size_t n, size_t k, const double* A, size_t lda, const double* B,
                          size_t ldb, double* C, size_t ldc, const GemmEpilogue& ep);

// rng (rng.hip)
int launch_rng_uniform(Context* c, uint64_t state, double* out, size_t n);
void lcg_jump_host(unsigned long long delta, unsigned long long* mult, unsigned long long* plus);  // s -> mult * s + plus advances `delta` steps
int launch_rng_normal(Context* c, uint64_t state, double* out, size_t n);
// scaled / transformed draws of the same stream (rng.hip): exactly one of out64 / out32 is set
int launch_rng_unifrnd(Context* c, uint64_t state, double a, double b, double* out64, float* out32, size_t n);
int launch_rng_exponential(Context* c, uint64_t state, double mu, double* out64, float* out32, size_t n);
int launch_rng_normrnd(Context* c, uint64_t state, double mu, double sigma, double* out64, float* out32, size_t n);
int launch_rng_integer_range(Context* c, uint64_t state, long long lower, unsigned long long span, double* out64, float* out32, size_t n);
int launch_rng_uniform_f32(Context* c, uint64_t state, float* out, size_t n);
int launch_rng_normal_f32(Context* c, uint64_t state, float* out, size_t n);
int launch_stochastic_evolution_f32(Context* c, uint64_t state, const float* in, float* out, size_t n, double drift,
                                    double scale, unsigned steps, uint64_t draws_per_step);
int image_normalize_device_f32(Context* c, const float* x, float* y, size_t batch, size_t height, size_t width, double epsilon,
                               int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma);
uint64_t lcg_advance(uint64_t state, uint64_t delta);
int image_normalize_device(Context* c, const double* x, double* y, size_t batch, size_t height, size_t width, double epsilon,
                           int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma);
int diag_extract_device(Context* c, const double* a, size_t rows, long long offset, size_t len, double* out);
int cov_sanitize_diag_device(Context* c, double* cm, size_t n);
// tall-skinny Gram matrix (X - 1 mu)' (X - 1 mu) on the VALU, cols <= 40 (special.hip); mu may be null
bool gram_skinny_applies(size_t rows, size_t cols);
// x2 / cols2: further columns from a second matrix of the same height (g is then (cols + cols2)^2)
int gram_skinny_device(Context* c, const double* x, size_t rows, size_t cols, const double* mu, double denom, bool sanitize, double* g,
                       const double* x2 = nullptr, size_t cols2 = 0);
// f32 storage read in place, f64 products and sums (a precision-32 provider's covariance / syrk of such shapes)
int gram_skinny_device_f32(Context* c, const float* x, size_t rows, size_t cols, const double* mu, double denom, bool sanitize, double* g);
int launch_stochastic_evolution(Context* c, uint64_t state, const double* in, double* out, size_t n, double drift,
                                double scale, unsigned steps, uint64_t draws_per_step);

// LU / solve (lu.hip)
// internal status of lu_factor_device: the matrix is clobbered, refactor a fresh copy (c->lu_conservative is now set)
static constexpr int RMHIP_LU_RETRY = -77;
static constexpr int RMHIP_LU_GROWTH = -79;  // internal status of lu_factor_device (mode 1): a multiplier exceeded the bound, refactor a fresh copy in mode 0
static constexpr int RMHIP_SUBST_RETRY = -78;  // internal status of substitute_few_rhs: the chain kernel timed out, gather the right-hand side again
int lu_factor_device(Context* c, double* A, size_t rows, size_t cols, size_t lda, int* perm_dev,
                     int* info_host, std::vector<int>* ipiv_host = nullptr, int mode = 0, double* ipiv_dev_f64 = nullptr,
                     double* deferred_guard = nullptr);  // deferred_guard (mode 1): no host read at all, status folded into *deferred_guard (device)
// interchanges from a device vector of doubles (rmhip_blk_lu's result), composed and applied on the device; RMHIP_ERR_UNSUPPORTED (no error
// string) when the view is too tall for the LDS map - the caller then composes on the host
int lu_swap_rows_from_device(Context* c, double* A, size_t lda, size_t nrows, size_t ncols, const double* ipiv_dev, size_t npiv);
int lu_swap_rows_device(Context* c, double* A, size_t lda, size_t ncols, const std::vector<int>& ipiv);
int trsm_lower_unit_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc);
int trsm_upper_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc);
int trsm_lower_nonunit_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc);
int transpose_device(Context* c, const double* src, size_t lds_, size_t rows, size_t cols, double* dst, size_t ldd);
int transpose_device_f32(Context* c, const float* src, size_t lds_, size_t rows, size_t cols, float* dst, size_t ldd);
int diag_stats_device(Context* c, const double* A, size_t lda, size_t n, double* min_abs, double* max_abs, size_t* zeros);
int lu_pad_identity_device(Context* c, double* W, size_t ldw, size_t n, size_t np);
int lu_solve_device(Context* c, const double* LU, size_t n, size_t lda, const int* perm_dev,
                    const double* B, size_t nrhs, size_t ldb, double* X, size_t ldx);
int lu_extract_device(Context* c, const double* LU, size_t rows, size_t cols, const int* perm_dev,
                      double* L, double* U, double* P, double* piv);

// svdsolve.hip: minimum-norm least squares by a one-sided Jacobi SVD with the reference's tolerance rule (mldivide.rs:380-404) - what the
// LU / Gram paths refuse (rank deficient, ill conditioned, singular), for min(rows, cols) <= svd_max_cols() (RMHIP_SVD_MAX_COLS)
static constexpr int kSvdMaxColsDefault = 4096;  // 0.15 s at 512, 0.45 s at 1024, 1.7 s at 2048, 9.6 s at 4096 (scripts/svd_sizes.py)
int svd_max_cols();
static constexpr int kSvdProxyMaxCols = 1024;  // up to here an LU with a tiny pivot RATIO (no pivot below the cut-off) is re-answered by the SVD
int svd_solve_device(Context* c, const double* A, size_t m, size_t n, const double* B, size_t nrhs, double* X, int* rank_out);
// the same decomposition behind rank / cond / pinv (rank.rs, cond.rs, pinv.rs: nalgebra's SVD on the CPU)
int svd_values_host(Context* c, const char* who, const double* A, size_t m, size_t n, std::vector<double>* values);
double svd_default_tolerance(const std::vector<double>& values, size_t m, size_t n);
int svd_pinv_device(Context* c, const double* A, size_t m, size_t n, double tol, double* X);

// small_solve.hip: x = A \ B for small n (policy: <= 64), nrhs <= 16 in one launch (the augmented matrix in the LDS of one CU) + the pivot statistics
bool small_solve_applies(size_t n, size_t nrhs);
int small_solve_device(Context* c, const double* A, const double* B, size_t n, size_t nrhs, double* X, double* min_abs, double* max_abs, size_t* bad);

// opaque handle -> Context (rmhip_core.cpp)
Context* context_of(rmhip_ctx* h);
void comm_destroy(Context* c);  // comm.cpp


// ---- cooperative yield table (two-level LU) -----------------------------------------------------------------------------------------
// One counter per CU (index: XCC id << 8 | the cu / sh / se byte of HW_ID; kYieldSlots entries).  A workgroup of the LU's critical chain
// (k_rp_top, the main stream's dgemm / trsm / rows-below kernels) counts itself in while it runs; the update streams' eight-wave dgemm
// blocks read their CU's counter once per k tile and sleep while it is non-zero (dgemm.hip w8_tile<YIELD>): the fp64 VALU and the matrix
// pipe are one datapath per SIMD, and every instruction of a chain kernel otherwise queues behind the update block's MFMAs.
static constexpr unsigned kYieldSlots = 4096;
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned cu_slot() {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
}
struct CuAnnounce {
    unsigned* p;
    __device__ __forceinline__ explicit CuAnnounce(unsigned* tab) : p(tab ? tab + cu_slot() : nullptr) {
        if (p && threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every thread of the workgroup calls this (it synchronises the workgroup first: no wave is still computing when the CU is released)
    __device__ __forceinline__ void done() const {
        if (!p) return;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_sub(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
#endif

}  // namespace rmhip
