// reduce_kernels.hip -- ahead-of-time reductions: reduce_sum / reduce_sum_dim / reduce_mean(_dim) /
// reduce_min / reduce_max / reduce_prod (crates/runmat-accelerate-api/src/lib.rs:2709-2721,
// 2743-2792, 2858-2883; reference semantics crates/runmat-accelerate/src/simple_provider.rs:
// 6728-6806 and the CPU sum_tensor, runtime/.../reduction/sum.rs:996-1079).
// HBM-bound: coalesced loads, per-lane f64 accumulators, wave64 __shfl_down tree, LDS across the
// four waves of a block, deterministic two-stage combine (skel_reduce.h).
#include "common.h"
#include "reduce_plan.h"
#include "skel_common.h"
#include "skel_reduce.h"

namespace rmhip {

// T = storage type (double, or float for precision-32 contexts); accumulation is f64 either way
template <class T>
struct IdentityVal {
    const T* __restrict__ x;
    __device__ __forceinline__ double operator()(rm_u64 idx) const { return (double)x[idx]; }
};

template <int OP, class T>
__global__ void __launch_bounds__(RM_ABLOCK) k_reduce_contig(const T* x, rm_u64 red, rm_u64 nslices,
                                                             rm_u64 nsplit, double* pv, double* pn) {
    IdentityVal<T> f{x};
    rm_reduce_contig<OP>(f, red, nslices, nsplit, pv, pn);
}
// Same reduction over 16-byte vectors (plain tensors, even slice length, 16-byte aligned base): 1 KiB per wave
// instruction instead of 512 B, non-temporal.  The pairing changes only the (deterministic) summation grouping.
typedef float rm_rv2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rm_rv2 rm_load2(const double* x, rm_u64 i2) { return __builtin_nontemporal_load((const rm_rv2*)x + i2); }
__device__ __forceinline__ rm_rv2 rm_load2(const float* x, rm_u64 i2) {
    const rm_rv2f v = __builtin_nontemporal_load((const rm_rv2f*)x + i2);
    rm_rv2 r = {(double)v.x, (double)v.y};
    return r;
}
typedef rm_rv2 rm_rv2u __attribute__((aligned(8)));
typedef rm_rv2f rm_rv2fu __attribute__((aligned(4)));
__device__ __forceinline__ rm_rv2 rm_load2_at(const double* p, bool single) {
    if (single) return rm_rv2{__builtin_nontemporal_load(p), 0.0};
    return (rm_rv2)__builtin_nontemporal_load((const rm_rv2u*)p);
}
__device__ __forceinline__ rm_rv2 rm_load2_at(const float* p, bool single) {
    if (single) return rm_rv2{(double)__builtin_nontemporal_load(p), 0.0};
    const rm_rv2f v = (rm_rv2f)__builtin_nontemporal_load((const rm_rv2fu*)p);
    return rm_rv2{(double)v.x, (double)v.y};
}
template <class T>
struct IdentityVal2 {
    const T* __restrict__ x;
    __device__ __forceinline__ rm_rv2 operator()(rm_u64 i2) const { return rm_load2(x, i2); }
    // odd slice length: pairs at element offsets that are only element-aligned in every other slice, and one leftover element
    __device__ __forceinline__ rm_rv2 pair_at(rm_u64 e) const { return rm_load2_at(x + e, false); }
    __device__ __forceinline__ double one_at(rm_u64 e) const { return (double)__builtin_nontemporal_load(x + e); }
};
template <int OP, class T, bool ODD = false>
__global__ void __launch_bounds__(RM_ABLOCK) k_reduce_contig_v2(const T* x, rm_u64 red, rm_u64 nslices,
                                                                rm_u64 nsplit, double* pv, double* pn) {
    IdentityVal2<T> f2{x};
    rm_reduce_contig_v2<OP, ODD>(f2, red, nslices, nsplit, pv, pn);
}

// Many SHORT contiguous slices (red < 256: sum(x,1) of a 3 x N or 32 x N matrix).  Kernel A gives every slice a block of its own -
// 256 threads for a few elements, half a million blocks for a 32 x 524288 matrix: 731 us where the bytes take 25.  Here a block
// takes S = min(256, 4096 / red) consecutive slices - one contiguous tile of S * red elements -, stages it in LDS with coalesced
// loads (one pad per 32 elements: the per-thread walks then fall on distinct banks) and thread t folds slice t in ascending order,
// the CPU's own sequence.  One partial per slice; the flat finalize applies the NaN / mean policy.
static constexpr int SHORT_TILE = 4096;
__device__ __forceinline__ int short_pad(int i) { return i + (i >> 5); }
template <int OP, class T>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_short(const T* __restrict__ x, rm_u64 red, rm_u64 nslices, unsigned per_block, double* pv,
                                                            double* pn) {
    __shared__ double tile[SHORT_TILE + SHORT_TILE / 32 + 1];
    const rm_u64 s0 = (rm_u64)blockIdx.x * per_block;
    const rm_u64 ns = nslices - s0 < per_block ? nslices - s0 : per_block;
    const rm_u64 count = ns * red;
    const T* src = x + s0 * red;
    for (rm_u64 i = threadIdx.x; i < count; i += RM_RBLOCK) tile[short_pad((int)i)] = (double)__builtin_nontemporal_load(src + i);
    __syncthreads();
    if (threadIdx.x >= ns) return;
    RmAcc a = rm_acc_init<OP>();
    const int b = (int)(threadIdx.x * red);
    for (int r = 0; r < (int)red; ++r) rm_acc_add<OP>(a, tile[short_pad(b + r)]);
    pv[s0 + threadIdx.x] = a.v;
    pn[s0 + threadIdx.x] = a.nan;
}

template <int OP, class T>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_strided(const T* x, rm_u64 pre, rm_u64 red, rm_u64 nsplit,
                                                              int tx, double* pv, double* pn) {
    IdentityVal<T> f{x};
    rm_reduce_strided<OP>(f, pre, red, nsplit, tx, pv, pn);
}

// Kernel B over 16-byte vectors (plain tensors, even `pre`, 16-byte aligned base): a thread owns TWO adjacent output
// slices and walks its chunk of the reduced extent in ascending order with U non-temporal loads in flight - the same
// per-slice summation order as rm_reduce_strided with ty == 1, at 1 KiB per wave instruction.  grid = (ceil(pre/512),
// nsplit, post).
// ODD: `pre` is odd (or the base only element-aligned).  Pairs start at even element offsets of every line, which are then only
// 8-byte (f64) / 4-byte (f32) aligned in every other line: the same 16-/8-byte loads on unaligned addresses (the hardware splits the
// ones that straddle), and the last row - a pair of one - is loaded as a scalar.  8191 x 8192: 110 us on the generic 8-byte kernel.
template <int OP, class T, int U, bool ODD = false>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_strided_v2(const T* x, rm_u64 pre, rm_u64 red, rm_u64 nsplit, unsigned win,
                                                                 double* pv, double* pn) {
    // a block owns `win` <= 256 pairs: the windows are balanced (host), so a row count just above a multiple of 512 does not
    // leave a column of nearly empty blocks behind (8256 rows: 16 full windows + one of 32 pairs ran 109 us against 86)
    const rm_u64 i2 = (rm_u64)blockIdx.x * win + threadIdx.x;  // pair index along `pre`
    const rm_u64 pre2 = ODD ? (pre + 1) >> 1 : pre >> 1;
    if (threadIdx.x >= win || i2 >= pre2) return;
    const bool single = ODD && 2 * i2 + 1 >= pre;  // the last row of an odd `pre`
    const rm_u64 split = blockIdx.y, j = blockIdx.z;
    const rm_u64 chunk = (red + nsplit - 1) / nsplit;
    const rm_u64 begin = split * chunk;
    rm_u64 end = begin + chunk;
    if (end > red) end = red;
    RmAcc a0 = rm_acc_init<OP>(), a1 = rm_acc_init<OP>();
    const rm_u64 base2 = i2 + pre2 * red * j;          // even form: in pairs
    const T* const xo = x + 2 * i2 + pre * red * j;     // odd form: in elements
    auto ld = [&](rm_u64 rr) -> rm_rv2 { return ODD ? rm_load2_at(xo + pre * rr, single) : rm_load2(x, base2 + pre2 * rr); };
    rm_u64 r = begin;
    for (; r + U <= end; r += U) {
        rm_rv2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld(r + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rm_acc_add<OP>(a0, v[u].x);
            rm_acc_add<OP>(a1, v[u].y);
        }
    }
    if (r < end) {  // the last, partial group: its loads go out together too (one at a time they cost a memory round trip each)
        rm_rv2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u < end) v[u] = ld(r + u);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u < end) {
                rm_acc_add<OP>(a0, v[u].x);
                rm_acc_add<OP>(a1, v[u].y);
            }
    }
    const rm_u64 slice = 2 * i2 + pre * j;
    pv[slice * nsplit + split] = a0.v;
    pn[slice * nsplit + split] = a0.nan;
    if (!single) {
        pv[(slice + 1) * nsplit + split] = a1.v;
        pn[(slice + 1) * nsplit + split] = a1.nan;
    }
}
template <int OP>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_final(const double* pv, const double* pn, rm_u64 nslices,
                                                            rm_u64 nsplit, rm_u64 red, int mean, int omitnan,
                                                            double scale, double* out) {
    rm_reduce_finalize<OP>(pv, pn, nslices, nsplit, red, mean, omitnan, scale, out);
}
template <int OP>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_final_flat(const double* pv, const double* pn, rm_u64 nslices, rm_u64 nsplit, rm_u64 red,
                                                                 int mean, int omitnan, double scale, double* out) {
    rm_reduce_finalize_flat<OP>(pv, pn, nslices, nsplit, red, mean, omitnan, scale, out);
}

template <int OP, class T>
static int run_reduce(Context* c, int mean, int nan_mode, const T* x, size_t pre, size_t red, size_t post,
                      double* out) {
    if (pre == 0 || post == 0) return RMHIP_OK;  // no output slices
    const ReducePlan p = plan_reduction(pre, red, post, c->num_cus, (unsigned)sizeof(T));
    if (!p.valid) return fail(RMHIP_ERR_UNSUPPORTED, "reduce: geometry [%zu,%zu,%zu] exceeds launch limits", pre, red, post);
    // Kernel B in its 16-byte form (two adjacent slices per thread, 256-thread blocks, non-temporal loads, the partial last
    // group of a chunk loaded together like the full ones) with THREE blocks per CU.  The block count matters more than
    // anything inside the kernel, and not monotonically (scripts/red_bpc_ab.sh, sum(x,2) in us for 8192^2 / 16384x4096 /
    // 4096x16384 / 7936x8192): 1 per CU 104 / 95 / 104 / 105, 2: 100 / 103 / 104 / 99, 3: 86 / 88 / 86 / 86, 4: 101 /
    // 100 / 109 / 92, 5: 93 / 96 / 89 / 93, 6: 96 / 106 / 93 / 91, 8: 113, 16: 130 - every stream is a strided column walk
    // and the streams run in lockstep; 1024 blocks (exactly four per CU) with 128-column chunks is the worst point
    // (112 us), 512 or 768 the best.  The generic kernel B at its 8 blocks per CU: 113.8 us.  A 1024-thread version with
    // 16 KiB of every column per block: 137 us.
    uint64_t nsplit = p.nsplit;
    // dev knobs (A/B only): RMHIP_RED_B_MODE 0 = generic kernel B, 1/2 = the 16-byte form with 8 / 4 loads in flight;
    // RMHIP_RED_B_BPC = its target blocks per CU
    static const int b_mode = getenv("RMHIP_RED_B_MODE") ? atoi(getenv("RMHIP_RED_B_MODE")) : 1;
    static const int b_bpc = getenv("RMHIP_RED_B_BPC") ? atoi(getenv("RMHIP_RED_B_BPC")) : 3;
    const bool wide_b = !p.contiguous && b_mode > 0 && pre >= 512 && post <= 65535;
    const bool wide_odd = wide_b && ((pre & 1) != 0 || (((uintptr_t)x) & 15) != 0);  // unaligned pairs + a scalar last row
    unsigned wide_bx = 0, wide_win = RM_RBLOCK, wide_threads = RM_RBLOCK;
    if (wide_b) {
        // The number of windows along `pre` is a multiple of the XCD count.  Workgroups go to XCDs round robin in launch order
        // (x fastest), so with gridDim.x % 8 == 0 a window - the same 4 KiB of every column - is always walked by the same XCD,
        // whatever the chunk; otherwise the windows rotate over the XCDs from chunk to chunk.  Measured (sum(x,2), us, windows
        // before -> after): 8200 x 8192 17 -> 24: 109 -> 90; 8256 x 8192 17 -> 24: 108 -> 95; 16400 x 4096 33 -> 40: 110 -> 97;
        // 5000 x 13000 10 -> 16: 104 -> 84; 12000 x 6000: 93; shapes whose count already was a multiple of eight (8192, 7936,
        // 8190, 16384 rows: 16 / 32 windows) are where round 2's 84-89 us came from - "rows a multiple of 512" was a proxy.
        // The windows are balanced (128-byte granules) and a block has as many waves as its window needs.  RMHIP_RED_B_X8=0
        // restores the old geometry.  (Also measured: two or four pairs per thread, 8-16 KiB of every column per block: 103-152 us.)
        static const int b_x8 = getenv("RMHIP_RED_B_X8") ? atoi(getenv("RMHIP_RED_B_X8")) : 1;
        const StridedWidePlan w = plan_strided_wide(pre, red, post, c->num_cus, c->num_xcc, (unsigned)sizeof(T), b_bpc, b_x8 != 0);  // reduce_plan.h
        wide_bx = w.bx;
        wide_win = w.win;
        wide_threads = w.threads;
        nsplit = w.nsplit;
        static const long dev_chunk = getenv("RMHIP_RED_B_CHUNK") ? atol(getenv("RMHIP_RED_B_CHUNK")) : 0;  // dev knob: columns per chunk
        if (dev_chunk > 0) nsplit = ceil_div_u64(red, (uint64_t)dev_chunk);
        if (nsplit > 65535) nsplit = 65535;
    }
    const bool short_a = p.contiguous && red >= 1 && red < 256 && p.nslices >= 1024;  // many short contiguous slices: a tile of slices per block
    if (short_a) nsplit = 1;
    const size_t nparts = (size_t)(p.nslices * nsplit);
    RMHIP_TRY(c->ensure_scratch(2 * nparts * sizeof(double)));
    double* pv = c->scratch;
    double* pn = c->scratch + nparts;
    if (short_a) {
        unsigned per_block = (unsigned)(SHORT_TILE / red);
        if (per_block > RM_RBLOCK) per_block = RM_RBLOCK;
        hipLaunchKernelGGL((k_reduce_short<OP, T>), dim3((unsigned)ceil_div_u64(p.nslices, per_block)), dim3(RM_RBLOCK), 0, c->stream, x, (rm_u64)red,
                           (rm_u64)p.nslices, per_block, pv, pn);
    } else if (p.contiguous && (red & 1) == 0 && red >= 2048 && (((uintptr_t)x) & 15) == 0)
        hipLaunchKernelGGL((k_reduce_contig_v2<OP, T>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, x, (rm_u64)red,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else if (p.contiguous && red >= 2048)  // odd slice length or an element-aligned base: the same kernel on unaligned pairs
        hipLaunchKernelGGL((k_reduce_contig_v2<OP, T, true>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, x, (rm_u64)red,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else if (p.contiguous)
        hipLaunchKernelGGL((k_reduce_contig<OP, T>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, x,
                           (rm_u64)red, (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else if (wide_b && b_mode == 2 && !wide_odd)
        hipLaunchKernelGGL((k_reduce_strided_v2<OP, T, 4>), dim3(wide_bx, (unsigned)nsplit, (unsigned)post), dim3(wide_threads), 0, c->stream,
                           x, (rm_u64)pre, (rm_u64)red, (rm_u64)nsplit, wide_win, pv, pn);
    else if (wide_odd)
        hipLaunchKernelGGL((k_reduce_strided_v2<OP, T, 8, true>), dim3(wide_bx, (unsigned)nsplit, (unsigned)post), dim3(wide_threads), 0, c->stream,
                           x, (rm_u64)pre, (rm_u64)red, (rm_u64)nsplit, wide_win, pv, pn);
    else if (wide_b)
        hipLaunchKernelGGL((k_reduce_strided_v2<OP, T, 8>), dim3(wide_bx, (unsigned)nsplit, (unsigned)post), dim3(wide_threads), 0, c->stream,
                           x, (rm_u64)pre, (rm_u64)red, (rm_u64)nsplit, wide_win, pv, pn);
    else
        hipLaunchKernelGGL((k_reduce_strided<OP, T>), dim3(p.gx, p.gy, p.gz), dim3(RM_RBLOCK), 0, c->stream, x,
                           (rm_u64)pre, (rm_u64)red, (rm_u64)p.nsplit, p.tx, pv, pn);
    RMHIP_HIP_CHECK(hipGetLastError());
    if ((nsplit <= 8 && p.nslices >= 1024) || (nsplit <= 32 && p.nslices >= 16384)) {  // many slices, a handful of partials each: one thread per slice
        hipLaunchKernelGGL((k_reduce_final_flat<OP>), dim3((unsigned)ceil_div_u64(p.nslices, RM_RBLOCK)), dim3(RM_RBLOCK), 0, c->stream, pv, pn,
                           (rm_u64)p.nslices, (rm_u64)nsplit, (rm_u64)red, mean, nan_mode, 1.0, out);
    } else {
        const unsigned fb = (unsigned)ceil_div_u64(p.nslices, RM_RBLOCK / 64);
        hipLaunchKernelGGL((k_reduce_final<OP>), dim3(fb), dim3(RM_RBLOCK), 0, c->stream, pv, pn, (rm_u64)p.nslices,
                           (rm_u64)nsplit, (rm_u64)red, mean, nan_mode, 1.0, out);
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}

template <class T>
static int reduce_mid_any(Context* c, int op, int nan_mode, const T* x, size_t pre, size_t red, size_t post, double* out) {
    switch (op) {
        case RMHIP_RSUM: return run_reduce<RM_RSUM>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RMEAN: return run_reduce<RM_RSUM>(c, 1, nan_mode, x, pre, red, post, out);
        case RMHIP_RMIN: return run_reduce<RM_RMIN>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RMAX: return run_reduce<RM_RMAX>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RPROD: return run_reduce<RM_RPROD>(c, 0, nan_mode, x, pre, red, post, out);
        default: return fail(RMHIP_ERR_UNSUPPORTED, "reduce op %d not supported by provider", op);
    }
}
int launch_reduce_mid(Context* c, int op, int nan_mode, const double* x, size_t pre, size_t red, size_t post,
                      double* out) {
    return reduce_mid_any(c, op, nan_mode, x, pre, red, post, out);
}
int launch_reduce_mid_f32(Context* c, int op, int nan_mode, const float* x, size_t pre, size_t red, size_t post,
                          double* out) {
    return reduce_mid_any(c, op, nan_mode, x, pre, red, post, out);
}

// ---- dot: the producer a.*b folded into the same skeleton (no temporary array) -------------------
template <class T>
struct ProductVal {
    const T* __restrict__ a;
    const T* __restrict__ b;
    __device__ __forceinline__ double operator()(rm_u64 idx) const { return (double)a[idx] * (double)b[idx]; }
};
template <class T>
__global__ void __launch_bounds__(RM_ABLOCK) k_dot_contig(const T* a, const T* b, rm_u64 red, rm_u64 nslices,
                                                          rm_u64 nsplit, double* pv, double* pn) {
    ProductVal<T> f{a, b};
    rm_reduce_contig<RM_RSUM>(f, red, nslices, nsplit, pv, pn);
}
template <class T>
struct ProductVal2 {
    const T* __restrict__ a;
    const T* __restrict__ b;
    __device__ __forceinline__ rm_rv2 operator()(rm_u64 i2) const {
        const rm_rv2 va = rm_load2(a, i2), vb = rm_load2(b, i2);
        return va * vb;
    }
};
template <class T>
__global__ void __launch_bounds__(RM_ABLOCK) k_dot_contig_v2(const T* a, const T* b, rm_u64 red, rm_u64 nslices,
                                                             rm_u64 nsplit, double* pv, double* pn) {
    ProductVal2<T> f2{a, b};
    rm_reduce_contig_v2<RM_RSUM>(f2, red, nslices, nsplit, pv, pn);
}
template <class T>
__global__ void __launch_bounds__(RM_RBLOCK) k_dot_strided(const T* a, const T* b, rm_u64 pre, rm_u64 red,
                                                           rm_u64 nsplit, int tx, double* pv, double* pn) {
    ProductVal<T> f{a, b};
    rm_reduce_strided<RM_RSUM>(f, pre, red, nsplit, tx, pv, pn);
}

// many short contiguous slices (dot along the 32 rows of a 32 x N pair): the products of a tile of whole slices go through LDS, one
// thread then sums a slice in index order - k_reduce_short with the producer folded in (the generic kernel ran one block per slice:
// 580 us at 32 x 524288)
template <class T>
__global__ void __launch_bounds__(RM_RBLOCK) k_dot_short(const T* __restrict__ a, const T* __restrict__ b, rm_u64 red, rm_u64 nslices, unsigned per_block,
                                                         double* pv, double* pn) {
    __shared__ double tile[SHORT_TILE + SHORT_TILE / 32 + 1];
    const rm_u64 s0 = (rm_u64)blockIdx.x * per_block;
    const rm_u64 ns = nslices - s0 < per_block ? nslices - s0 : per_block;
    const rm_u64 count = ns * red;
    const T* sa = a + s0 * red;
    const T* sb = b + s0 * red;
    for (rm_u64 i = threadIdx.x; i < count; i += RM_RBLOCK)
        tile[short_pad((int)i)] = (double)__builtin_nontemporal_load(sa + i) * (double)__builtin_nontemporal_load(sb + i);
    __syncthreads();
    if (threadIdx.x >= ns) return;
    RmAcc acc = rm_acc_init<RM_RSUM>();
    const int base = (int)(threadIdx.x * red);
    for (int r = 0; r < (int)red; ++r) rm_acc_add<RM_RSUM>(acc, tile[short_pad(base + r)]);
    pv[s0 + threadIdx.x] = acc.v;
    pn[s0 + threadIdx.x] = acc.nan;
}

template <class T>
static int reduce_dot_any(Context* c, const T* a, const T* b, size_t pre, size_t red, size_t post, double* out) {
    if (pre == 0 || post == 0) return RMHIP_OK;  // no output slices
    ReducePlan p = plan_reduction(pre, red, post, c->num_cus, (unsigned)sizeof(T));
    if (!p.valid) return fail(RMHIP_ERR_UNSUPPORTED, "dot: geometry [%zu,%zu,%zu] exceeds launch limits", pre, red, post);
    const bool short_a = p.contiguous && red >= 1 && red < 256 && p.nslices >= 1024;
    if (short_a) p.nsplit = 1;
    const size_t nparts = (size_t)(p.nslices * p.nsplit);
    RMHIP_TRY(c->ensure_scratch(2 * nparts * sizeof(double)));
    double* pv = c->scratch;
    double* pn = c->scratch + nparts;
    if (short_a) {
        unsigned per_block = (unsigned)(SHORT_TILE / red);
        if (per_block > RM_RBLOCK) per_block = RM_RBLOCK;
        hipLaunchKernelGGL((k_dot_short<T>), dim3((unsigned)ceil_div_u64(p.nslices, per_block)), dim3(RM_RBLOCK), 0, c->stream, a, b, (rm_u64)red,
                           (rm_u64)p.nslices, per_block, pv, pn);
    } else if (p.contiguous && (red & 1) == 0 && red >= 2048 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0)
        hipLaunchKernelGGL((k_dot_contig_v2<T>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, a, b, (rm_u64)red,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else if (p.contiguous)
        hipLaunchKernelGGL((k_dot_contig<T>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, a, b, (rm_u64)red,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else
        hipLaunchKernelGGL((k_dot_strided<T>), dim3(p.gx, p.gy, p.gz), dim3(RM_RBLOCK), 0, c->stream, a, b, (rm_u64)pre,
                           (rm_u64)red, (rm_u64)p.nsplit, p.tx, pv, pn);
    RMHIP_HIP_CHECK(hipGetLastError());
    if (p.nsplit <= 8 && p.nslices >= 1024) {
        hipLaunchKernelGGL((k_reduce_final_flat<RM_RSUM>), dim3((unsigned)ceil_div_u64(p.nslices, RM_RBLOCK)), dim3(RM_RBLOCK), 0, c->stream, pv, pn,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, (rm_u64)red, 0, 0, 1.0, out);
    } else {
        const unsigned fb = (unsigned)ceil_div_u64(p.nslices, RM_RBLOCK / 64);
        hipLaunchKernelGGL((k_reduce_final<RM_RSUM>), dim3(fb), dim3(RM_RBLOCK), 0, c->stream, pv, pn, (rm_u64)p.nslices,
                           (rm_u64)p.nsplit, (rm_u64)red, 0, 0, 1.0, out);
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}
int launch_reduce_dot(Context* c, const double* a, const double* b, size_t pre, size_t red, size_t post, double* out) {
    return reduce_dot_any(c, a, b, pre, red, post, out);
}
int launch_reduce_dot_f32(Context* c, const float* a, const float* b, size_t pre, size_t red, size_t post, double* out) {
    return reduce_dot_any(c, a, b, pre, red, post, out);
}

int launch_reduce_all(Context* c, int op, int nan_mode, const double* x, size_t n, double* out) {
    return launch_reduce_mid(c, op, nan_mode, x, 1, n, 1, out);
}

}  // namespace rmhip
