// skel_reduce.h -- reduction kernel skeletons, templated on a value functor F (index -> f64).
// Used twice: instantiated ahead of time with an identity functor (reduce_kernels.hip:
// reduce_sum / reduce_sum_dim / reduce_mean / min / max / prod, reference semantics
// crates/runmat-accelerate/src/simple_provider.rs:6728-6806) and embedded as text into the
// hipRTC source of every fused reduction (the functor is then the folded producer expression of
// crates/runmat-accelerate/src/fusion.rs:1765-2077).
//
// Data view: the tensor is [pre, red, post] column-major and the middle extent is reduced;
// element (i, r, j) lives at i + pre*(r + red*j); output slice id = i + pre*j.
//
// Determinism: no atomics anywhere. Every partial is produced by a fixed thread in a fixed
// order and combined in a fixed order, so results are run-to-run reproducible. NaNs are counted,
// not propagated through the sum, so include/omit policies are applied once in finalize
// (CPU: crates/runmat-runtime/src/builtins/math/reduction/sum.rs:1031-1076).
// Requires skel_common.h first. No `#include` here (hipRTC inline text).
#ifndef RMHIP_SKEL_REDUCE
#define RMHIP_SKEL_REDUCE

#define RM_RSUM 0
#define RM_RMEAN 1
#define RM_RMIN 2
#define RM_RMAX 3
#define RM_RPROD 4
#define RM_RBLOCK 256
#define RM_ABLOCK 1024  // kernel A (contiguous slices) streams: 1024-thread blocks measured 8-20 % faster than 256 on
                        // the elementwise kernels (scripts/tune_ew_ab.py), and sum(x,'all') of 512 MiB went 0.105 -> see DESIGN.md

struct RmAcc {
    double v;
    double nan;  // number of NaN inputs seen (exact in f64 up to 2^53)
};

template <int OP>
__device__ __forceinline__ RmAcc rm_acc_init() {
    RmAcc a;
    a.nan = 0.0;
    a.v = (OP == RM_RMIN) ? __builtin_inf() : (OP == RM_RMAX) ? -__builtin_inf() : (OP == RM_RPROD) ? 1.0 : 0.0;
    return a;
}
template <int OP>
__device__ __forceinline__ double rm_combine(double a, double b) {
    if (OP == RM_RMIN) return b < a ? b : a;
    if (OP == RM_RMAX) return b > a ? b : a;
    if (OP == RM_RPROD) return a * b;
    return a + b;
}
template <int OP>
__device__ __forceinline__ void rm_acc_add(RmAcc& a, double x) {
    if (x != x) a.nan += 1.0;
    else a.v = rm_combine<OP>(a.v, x);
}
template <int OP>
__device__ __forceinline__ void rm_acc_merge(RmAcc& a, const RmAcc& b) {
    a.v = rm_combine<OP>(a.v, b.v);
    a.nan += b.nan;
}

// Fixed-order reduction across the RM_ABLOCK threads of a block: wave64 shuffle tree, then wave 0
// folds the wave results in wave order. Result valid in thread 0.
template <int OP>
__device__ __forceinline__ RmAcc rm_block_reduce(RmAcc a, RmAcc* lds /* >= RM_ABLOCK / 64 entries */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        RmAcc o;
        o.v = __shfl_down(a.v, off, 64);
        o.nan = __shfl_down(a.nan, off, 64);
        rm_acc_merge<OP>(a, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) rm_acc_merge<OP>(a, lds[w]);
    }
    return a;
}

// Kernel A: pre == 1 (each slice is `red` contiguous elements). grid = (nsplit, post).
template <int OP, class F>
__device__ __forceinline__ void rm_reduce_contig(const F& f, rm_u64 red, rm_u64 nslices, rm_u64 nsplit,
                                                 double* part_v, double* part_nan) {
    __shared__ RmAcc lds[RM_ABLOCK / 64];
    const rm_u64 slice = blockIdx.y + (rm_u64)gridDim.y * blockIdx.z;
    if (slice >= nslices) return;  // padding blocks of the (y, z) slice grid; uniform per block
    const rm_u64 split = blockIdx.x;
    rm_u64 chunk = (red + nsplit - 1) / nsplit;
    const rm_u64 bs = blockDim.x;  // RM_ABLOCK for long slices, RM_RBLOCK for short ones (host: reduce_plan.h)
    chunk = (chunk + bs - 1) / bs * bs;  // block-aligned chunks
    const rm_u64 begin = split * chunk;
    rm_u64 end = begin + chunk;
    if (end > red) end = red;
    const rm_u64 base = slice * red;
    RmAcc a0 = rm_acc_init<OP>(), a1 = rm_acc_init<OP>(), a2 = rm_acc_init<OP>(), a3 = rm_acc_init<OP>();
    rm_u64 r = begin + threadIdx.x;
    for (; r + 3 * bs < end; r += 4 * bs) {
        const double x0 = f(base + r), x1 = f(base + r + bs), x2 = f(base + r + 2 * bs), x3 = f(base + r + 3 * bs);
        rm_acc_add<OP>(a0, x0);
        rm_acc_add<OP>(a1, x1);
        rm_acc_add<OP>(a2, x2);
        rm_acc_add<OP>(a3, x3);
    }
    for (; r < end; r += bs) rm_acc_add<OP>(a0, f(base + r));
    rm_acc_merge<OP>(a0, a1);
    rm_acc_merge<OP>(a2, a3);
    rm_acc_merge<OP>(a0, a2);
    a0 = rm_block_reduce<OP>(a0, lds);
    if (threadIdx.x == 0) {
        part_v[slice * nsplit + split] = a0.v;
        part_nan[slice * nsplit + split] = a0.nan;
    }
}

// Kernel B: pre > 1. Threads run along `pre` (coalesced); each thread walks its slice's `red`
// extent in ascending order, so with nsplit == 1 and tx == RM_RBLOCK the per-output summation
// order equals the CPU's. grid = (ceil(pre/tx), nsplit, post); tx is a power of two <= 256.
template <int OP, class F>
__device__ __forceinline__ void rm_reduce_strided(const F& f, rm_u64 pre, rm_u64 red, rm_u64 nsplit, int tx,
                                                  double* part_v, double* part_nan) {
    __shared__ RmAcc lds[RM_RBLOCK];
    const int lx = threadIdx.x & (tx - 1);
    const int ly = threadIdx.x / tx;
    const int ty = RM_RBLOCK / tx;
    const rm_u64 i = (rm_u64)blockIdx.x * tx + lx;
    const rm_u64 split = blockIdx.y;
    const rm_u64 j = blockIdx.z;
    rm_u64 chunk = (red + nsplit - 1) / nsplit;
    const rm_u64 begin = split * chunk;
    rm_u64 end = begin + chunk;
    if (end > red) end = red;
    RmAcc a0 = rm_acc_init<OP>(), a1 = rm_acc_init<OP>();
    if (i < pre) {
        const rm_u64 base = i + pre * red * j;
        rm_u64 r = begin + ly;
        if (ty == 1) {  // sequential order; loads issued in groups of eight, adds stay in order
            for (; r + 7 < end; r += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = f(base + pre * (r + u));
#pragma unroll
                for (int u = 0; u < 8; ++u) rm_acc_add<OP>(a0, x[u]);
            }
            for (; r < end; ++r) rm_acc_add<OP>(a0, f(base + pre * r));
        } else {
            for (; r + ty < end; r += 2 * ty) {
                const double x0 = f(base + pre * r), x1 = f(base + pre * (r + ty));
                rm_acc_add<OP>(a0, x0);
                rm_acc_add<OP>(a1, x1);
            }
            for (; r < end; r += ty) rm_acc_add<OP>(a0, f(base + pre * r));
            rm_acc_merge<OP>(a0, a1);
        }
    }
    if (ty > 1) {
        lds[threadIdx.x] = a0;
        __syncthreads();
        if (ly == 0) {
            for (int y = 1; y < ty; ++y) rm_acc_merge<OP>(a0, lds[y * tx + lx]);
        }
    }
    if (ly == 0 && i < pre) {
        const rm_u64 slice = i + pre * j;
        part_v[slice * nsplit + split] = a0.v;
        part_nan[slice * nsplit + split] = a0.nan;
    }
}

// Finalize: one wave per slice folds its nsplit partials (lane-strided, then a shuffle tree) and
// applies NaN policy + scaling.  mode: RM_RSUM => * scale (scale == 1 for plain sums);
// RM_RMEAN => / count (CPU mean divides: mean.rs:1134-1151).
template <int OP>
__device__ __forceinline__ double rm_finalize_value(const RmAcc& a, rm_u64 red, int mean, int omitnan, double scale) {
    double r = a.v;
    const double cnt = (double)red - a.nan;
    if (OP == RM_RMIN || OP == RM_RMAX) {
        if ((!omitnan && a.nan > 0.0) || cnt <= 0.0) r = rm_nan();
    } else if (mean) {
        if (omitnan) r = cnt > 0.0 ? r / cnt : rm_nan();
        else r = a.nan > 0.0 ? rm_nan() : r / (double)red;
    } else {
        if (!omitnan && a.nan > 0.0) r = rm_nan();
        else r = r * scale;
    }
    return r;
}
template <int OP>
__device__ __forceinline__ void rm_reduce_finalize(const double* part_v, const double* part_nan, rm_u64 nslices,
                                                   rm_u64 nsplit, rm_u64 red, int mean, int omitnan, double scale,
                                                   double* out) {
    const rm_u64 slice = (rm_u64)blockIdx.x * (RM_RBLOCK / 64) + (threadIdx.x >> 6);
    if (slice >= nslices) return;
    const int lane = threadIdx.x & 63;
    RmAcc a = rm_acc_init<OP>();
    rm_u64 s = lane;
    const double* pv = part_v + slice * nsplit;
    const double* pn = part_nan + slice * nsplit;
    // four trips' loads in flight, merged in the same order as the plain loop (one dependent load per trip made
    // `sum(x,'all')`'s 2048 partials cost ~10 us of pure latency)
    for (; s + 192 < nsplit; s += 256) {
        RmAcc p0, p1, p2, p3;
        p0.v = pv[s];
        p1.v = pv[s + 64];
        p2.v = pv[s + 128];
        p3.v = pv[s + 192];
        p0.nan = pn[s];
        p1.nan = pn[s + 64];
        p2.nan = pn[s + 128];
        p3.nan = pn[s + 192];
        rm_acc_merge<OP>(a, p0);
        rm_acc_merge<OP>(a, p1);
        rm_acc_merge<OP>(a, p2);
        rm_acc_merge<OP>(a, p3);
    }
    for (; s < nsplit; s += 64) {
        RmAcc p;
        p.v = pv[s];
        p.nan = pn[s];
        rm_acc_merge<OP>(a, p);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        RmAcc o;
        o.v = __shfl_down(a.v, off, 64);
        o.nan = __shfl_down(a.nan, off, 64);
        rm_acc_merge<OP>(a, o);
    }
    if (lane == 0) out[slice] = rm_finalize_value<OP>(a, red, mean, omitnan, scale);
}

// The same with ONE THREAD per slice, its few partials merged in split order: for many slices with one or a handful of partials each
// (sum(x,2) of a 524288 x 32 matrix: 524288 slices, one partial) a wave per slice is 64 times the launch the work needs - there the
// finalize took longer than the reduction itself.
template <int OP>
__device__ __forceinline__ void rm_reduce_finalize_flat(const double* part_v, const double* part_nan, rm_u64 nslices, rm_u64 nsplit, rm_u64 red,
                                                        int mean, int omitnan, double scale, double* out) {
    const rm_u64 slice = (rm_u64)blockIdx.x * RM_RBLOCK + threadIdx.x;
    if (slice >= nslices) return;
    RmAcc a = rm_acc_init<OP>();
    for (rm_u64 s = 0; s < nsplit; ++s) {
        RmAcc p;
        p.v = part_v[slice * nsplit + s];
        p.nan = part_nan[slice * nsplit + s];
        rm_acc_merge<OP>(a, p);
    }
    out[slice] = rm_finalize_value<OP>(a, red, mean, omitnan, scale);
}

// ---- kernel A over 16-byte vectors: the functor returns two adjacent elements of a slice per call (plain tensors in
// reduce_kernels.hip, generated fused reductions through `RmVal2`); `red` even, or ODD with the functor's unaligned accessors.
typedef double rm_rv2 __attribute__((ext_vector_type(2)));
// ODD: `red` is odd (or the base only element-aligned): slice s starts at element s * red, its red / 2 pairs are loaded with
// unaligned 16-byte loads and the last element joins the accumulator of the thread that would own the next pair.
template <int OP, bool ODD = false, class F2>
__device__ __forceinline__ void rm_reduce_contig_v2(const F2& f2, rm_u64 red, rm_u64 nslices, rm_u64 nsplit, double* pv,
                                                    double* pn) {
    __shared__ RmAcc lds[RM_ABLOCK / 64];
    const rm_u64 slice = blockIdx.y + (rm_u64)gridDim.y * blockIdx.z;
    if (slice >= nslices) return;
    const rm_u64 split = blockIdx.x, bs = blockDim.x, red2 = red >> 1;
    rm_u64 chunk = (red2 + nsplit - 1) / nsplit;
    chunk = (chunk + bs - 1) / bs * bs;
    const rm_u64 begin = split * chunk;
    rm_u64 end = begin + chunk;
    if (end > red2) end = red2;
    const rm_u64 base = slice * red2, ebase = slice * red;
    auto ld = [&](rm_u64 rr) -> rm_rv2 {
        if constexpr (ODD) return f2.pair_at(ebase + 2 * rr);
        else return f2(base + rr);
    };
    RmAcc a0 = rm_acc_init<OP>(), a1 = rm_acc_init<OP>(), a2 = rm_acc_init<OP>(), a3 = rm_acc_init<OP>();
    rm_u64 r = begin + threadIdx.x;
    for (; r + 3 * bs < end; r += 4 * bs) {
        const rm_rv2 x0 = ld(r), x1 = ld(r + bs), x2 = ld(r + 2 * bs), x3 = ld(r + 3 * bs);
        rm_acc_add<OP>(a0, x0.x);
        rm_acc_add<OP>(a1, x1.x);
        rm_acc_add<OP>(a2, x2.x);
        rm_acc_add<OP>(a3, x3.x);
        rm_acc_add<OP>(a0, x0.y);
        rm_acc_add<OP>(a1, x1.y);
        rm_acc_add<OP>(a2, x2.y);
        rm_acc_add<OP>(a3, x3.y);
    }
    for (; r < end; r += bs) {
        const rm_rv2 v = ld(r);
        rm_acc_add<OP>(a0, v.x);
        rm_acc_add<OP>(a0, v.y);
    }
    if constexpr (ODD) {
        // the leftover element: pair index red2 would be its pair - the chunk that contains that index owns it, and exactly one of
        // its threads ends its walk there
        const rm_u64 owner = red2 / chunk < nsplit - 1 ? red2 / chunk : nsplit - 1;
        if ((red & 1) && split == owner && r == red2) rm_acc_add<OP>(a0, f2.one_at(ebase + red - 1));
    }
    rm_acc_merge<OP>(a0, a1);
    rm_acc_merge<OP>(a2, a3);
    rm_acc_merge<OP>(a0, a2);
    a0 = rm_block_reduce<OP>(a0, lds);
    if (threadIdx.x == 0) {
        pv[slice * nsplit + split] = a0.v;
        pn[slice * nsplit + split] = a0.nan;
    }
}

// ---- kernel B over 16-byte vectors for generated fused reductions: a thread owns TWO adjacent slices (pair index along `pre`) and
// walks its chunk of the reduced extent in ascending order with U vector loads in flight - the per-slice summation order of
// rm_reduce_strided with ty == 1.  Geometry from reduce_plan.h (plan_strided_wide): blockIdx.x = window of `win` pairs, blockIdx.y =
// chunk, blockIdx.z = post.  (The plain-tensor kernel k_reduce_strided_v2 in reduce_kernels.hip is the same walk plus the odd-extent form.)
template <int OP, int U, class F2>
__device__ __forceinline__ void rm_reduce_strided_v2(const F2& f2, rm_u64 pre, rm_u64 red, rm_u64 nsplit, unsigned win, double* pv, double* pn) {
    const rm_u64 i2 = (rm_u64)blockIdx.x * win + threadIdx.x;
    const rm_u64 pre2 = pre >> 1;
    if (threadIdx.x >= win || i2 >= pre2) return;
    const rm_u64 split = blockIdx.y, j = blockIdx.z;
    const rm_u64 chunk = (red + nsplit - 1) / nsplit;
    const rm_u64 begin = split * chunk;
    rm_u64 end = begin + chunk;
    if (end > red) end = red;
    RmAcc a0 = rm_acc_init<OP>(), a1 = rm_acc_init<OP>();
    const rm_u64 base2 = i2 + pre2 * red * j;
    rm_u64 r = begin;
    for (; r + U <= end; r += U) {
        rm_rv2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = f2(base2 + pre2 * (r + u));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rm_acc_add<OP>(a0, v[u].x);
            rm_acc_add<OP>(a1, v[u].y);
        }
    }
    for (; r < end; ++r) {
        const rm_rv2 v = f2(base2 + pre2 * r);
        rm_acc_add<OP>(a0, v.x);
        rm_acc_add<OP>(a1, v.y);
    }
    const rm_u64 slice = 2 * i2 + pre * j;
    pv[slice * nsplit + split] = a0.v;
    pn[slice * nsplit + split] = a0.nan;
    pv[(slice + 1) * nsplit + split] = a1.v;
    pn[(slice + 1) * nsplit + split] = a1.nan;
}

#endif  // RMHIP_SKEL_REDUCE
