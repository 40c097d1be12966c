// reduce2.hip -- the reduction hooks next to sum / mean / min / max on the provider trait
// (crates/runmat-accelerate-api/src/lib.rs):
//   reduce_min_dim / reduce_max_dim -> ReduceDimResult{values, indices}   :2864-2883, :513-517
//   reduce_std / reduce_std_dim                                           :2786-2802
//   reduce_nnz(_dim), reduce_any(_dim), reduce_all(_dim)                  :2730-2742, :2803-2850
//   cumsum_scan / cumprod_scan                                            :2884-2891, :2908-2915
// Semantics are the CPU builtins' (crates/runmat-runtime/src/builtins/math/reduction/{min,max,std,nnz,any,all,cumsum,
// cumprod}.rs; cited at each accumulator), not the in-process test provider's where the two differ (its reduce_min_dim
// ignores NaNs, simple_provider.rs:7301-7345; the runtime only calls the hook in "includenan" mode, min.rs:795-800, and
// expects what its host path gives).
//
// One skeleton for everything that reduces: the tensor is viewed as [pre, red, post] (column-major, the middle extent is
// reduced), every slice is cut into nsplit chunks so that the grid covers the chip several times (reduce_plan.h), a chunk
// folds into an accumulator, and a final kernel merges the chunks of a slice IN CHUNK ORDER - no atomics, results do not
// depend on scheduling.  HBM-bound: one 8-byte read per element.
#include "common.h"
#include "reduce_plan.h"

namespace rmhip {

typedef unsigned long long u64;

__device__ __forceinline__ double r2_nan() { return __longlong_as_double(0x7ff8000000000000ll); }  // f64::NAN

// ---- accumulators ---------------------------------------------------------------------------------------------------------
// arg-min / arg-max.  Values map to integer keys whose unsigned order is the builtin's order, smaller = better:
//   * total order of the doubles with -0 below +0 (min.rs:1519-1531 prefers -0 over +0, max.rs:1715-1727 +0 over -0);
//     for max the key is complemented;
//   * ties keep the FIRST index (a later equal value never replaces, same functions);
//   * includenan: the first NaN fixes the result - value NaN, index of that NaN (min.rs:1443-1455): NaN gets key 0;
//     omitnan: NaNs never win (key ~0); a slice of NaNs only gives value NaN and index NaN (min.rs:1060-1064).
// No real value maps to 0 or ~0 (only NaN bit patterns do), so the two sentinels are unambiguous.
template <bool MAX, bool OMIT>
struct ArgAcc {
    u64 key, idx;
    __device__ __forceinline__ void init() {
        key = ~0ull;
        idx = ~0ull;
    }
    // every thread feeds its elements in ASCENDING index order, so a strict compare keeps the first occurrence (the index only breaks
    // ties between threads / chunks, in merge)
    __device__ __forceinline__ void add(u64 k, double v) {
        const long long b = __double_as_longlong(v);
        u64 ord = (u64)b ^ ((u64)(b >> 63) | 0x8000000000000000ull);  // negative: ~b, else b with the sign bit set
        if (MAX) ord = ~ord;
        if (v != v) ord = OMIT ? ~0ull : 0ull;
        if (ord < key) {
            key = ord;
            idx = k;
        }
    }
    __device__ __forceinline__ void merge(const ArgAcc& o) {
        if (o.key < key || (o.key == key && o.idx < idx)) {
            key = o.key;
            idx = o.idx;
        }
    }
};
template <bool MAX>
__device__ __forceinline__ void arg_decode(u64 key, u64 idx, double* value, double* index) {
    if (key == ~0ull) {  // nothing but NaNs under omitnan (or an empty slice)
        *value = r2_nan();
        *index = r2_nan();
        return;
    }
    *index = (double)(idx + 1);  // 1-based (min.rs:1067-1074)
    if (key == 0) {
        *value = r2_nan();
        return;
    }
    const u64 ord = MAX ? ~key : key;
    const u64 b = (ord >> 63) ? (ord & 0x7fffffffffffffffull) : ~ord;
    *value = __longlong_as_double((long long)b);
}

// count / mean / M2 (std.rs:858-935: Welford's update element by element, NaNs counted apart).  Here: batches of eight by the
// two-pass formula, batches and chunks merged with Chan's formula in a fixed order (tails and NaN-carrying batches element by
// element) - the same quantities to rounding, not the CPU's operation sequence.
struct MomAcc {
    double n, mean, m2, nan;
    __device__ __forceinline__ void init() { n = mean = m2 = nan = 0.0; }
    __device__ __forceinline__ void add(u64, double v) {
        if (v != v) {
            nan += 1.0;
            return;
        }
        n += 1.0;
        const double delta = v - mean;
        // delta / n with the hardware reciprocal + two Newton steps (n is a small exact integer: the quotient differs from the CPU's
        // division by at most an ulp, far inside what merging chunks changes anyway); an IEEE division is ~35 instructions per
        // element and made the kernel VALU-bound at 2.8 TB/s
        double rn = __builtin_amdgcn_rcp(n);
        rn = __builtin_fma(rn, __builtin_fma(-n, rn, 1.0), rn);
        rn = __builtin_fma(rn, __builtin_fma(-n, rn, 1.0), rn);
        mean += delta * rn;
        const double delta2 = v - mean;
        m2 += delta * delta2;
    }
    // eight values at once: their own mean and M2 by the two-pass formula in registers (independent operations), then ONE Chan
    // merge - a third of the instructions of eight Welford updates and no eight-deep dependent chain (the per-element form ran at
    // 3 TB/s).  A batch holding a NaN goes element by element.
    __device__ __forceinline__ void add8(const double (&v)[8]) {
        bool clean = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) clean = clean && (v[u] == v[u]);
        if (!clean) {
#pragma unroll
            for (int u = 0; u < 8; ++u) add(0, v[u]);
            return;
        }
        const double s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const double mb = s * 0.125;
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            const double d0 = v[u] - mb, d1 = v[u + 1] - mb;
            q0 = __builtin_fma(d0, d0, q0);
            q1 = __builtin_fma(d1, d1, q1);
        }
        const double tot = n + 8.0, delta = mb - mean;
        double rt = __builtin_amdgcn_rcp(tot);
        rt = __builtin_fma(rt, __builtin_fma(-tot, rt, 1.0), rt);
        rt = __builtin_fma(rt, __builtin_fma(-tot, rt, 1.0), rt);
        const double f = 8.0 * rt;  // n_b / tot
        mean = __builtin_fma(delta, f, mean);
        m2 += (q0 + q1) + delta * delta * (n * f);
        n = tot;
    }
    __device__ __forceinline__ void merge(const MomAcc& o) {
        nan += o.nan;
        if (o.n == 0.0) return;
        if (n == 0.0) {
            n = o.n;
            mean = o.mean;
            m2 = o.m2;
            return;
        }
        const double tot = n + o.n, delta = o.mean - mean;
        mean += delta * (o.n / tot);
        m2 += o.m2 + delta * delta * (n * (o.n / tot));
        n = tot;
    }
};

// truth counts: nonzero non-NaN elements and NaNs (nnz.rs:358 `is_nan() || v != 0`; any.rs:620-621, 722-733;
// all.rs:568-569, 671-703)
struct TruthAcc {
    u64 nz, nan;
    __device__ __forceinline__ void init() { nz = nan = 0; }
    __device__ __forceinline__ void add(u64, double v) {
        if (v != v) ++nan;
        else if (v != 0.0) ++nz;
    }
    __device__ __forceinline__ void merge(const TruthAcc& o) {
        nz += o.nz;
        nan += o.nan;
    }
};

// sum and sum of squares in one pass (reduce_moments_nd: E[x] and E[x^2] along a dimension); NaNs propagate through both sums
struct SqAcc {
    double s, q;
    __device__ __forceinline__ void init() { s = q = 0.0; }
    __device__ __forceinline__ void add(u64, double v) {
        s += v;
        q += v * v;
    }
    __device__ __forceinline__ void merge(const SqAcc& o) {
        s += o.s;
        q += o.q;
    }
};

// ---- stage 1 ----------------------------------------------------------------------------------------------------------------
// eight elements of a thread's run, in ascending index order
template <class Acc>
__device__ __forceinline__ void r2_fold8(Acc& a, const u64 (&k)[8], const double (&v)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a.add(k[u], v[u]);
}
template <>
__device__ __forceinline__ void r2_fold8<MomAcc>(MomAcc& a, const u64 (&)[8], const double (&v)[8]) {
    a.add8(v);
}
template <>
__device__ __forceinline__ void r2_fold8<SqAcc>(SqAcc& a, const u64 (&)[8], const double (&v)[8]) {
    // pairwise inside the batch (independent adds instead of two eight-deep chains), then one add into the running sums
    a.s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    a.q += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
}

static constexpr int R2_BLOCK = 256;

// pre == 1: slice s is `red` contiguous elements.  grid (nsplit, slices in y, z)
template <class Acc, class T = double>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_contig(const T* __restrict__ x, u64 red, u64 nslices, u64 nsplit, Acc* __restrict__ part) {
    __shared__ Acc lds[R2_BLOCK];
    const u64 slice = blockIdx.y + (u64)gridDim.y * blockIdx.z;
    if (slice >= nslices) return;
    const u64 split = blockIdx.x;
    u64 chunk = (red + nsplit - 1) / nsplit;
    chunk = (chunk + R2_BLOCK - 1) / R2_BLOCK * R2_BLOCK;
    const u64 begin = split * chunk;
    u64 end = begin + chunk;
    if (end > red) end = red;
    const T* xs = x + slice * red;
    Acc a;
    a.init();
    u64 r = begin + threadIdx.x;
    for (; r + 7 * R2_BLOCK < end; r += 8 * R2_BLOCK) {  // eight loads in flight; folded in index order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (double)__builtin_nontemporal_load(xs + r + u * R2_BLOCK);
        u64 k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = r + u * R2_BLOCK;
        r2_fold8(a, k, v);
    }
    for (; r < end; r += R2_BLOCK) a.add(r, (double)__builtin_nontemporal_load(xs + r));
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int s = R2_BLOCK / 2; s > 0; s >>= 1) {  // fixed tree
        if ((int)threadIdx.x < s) {
            Acc m = lds[threadIdx.x];
            m.merge(lds[threadIdx.x + s]);
            lds[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[slice * nsplit + split] = lds[0];
}

// the same with 16-byte loads (even `red`, 16-byte aligned base: every slice starts on a pair) and the next batch requested before
// the current one is folded - the fold of an arg accumulator is a chain of dependent compare / selects during which a wave would
// otherwise have nothing in flight.  A thread's elements still arrive in ascending index order.
// ODD: odd `red` or an element-aligned base - the pairs are loaded from 8-byte aligned addresses and the slice's last element is folded
// by the thread whose walk ends at its pair index (still ascending within the thread).
// T = storage type: float on a precision-32 provider (widened in registers: exact and order preserving), a pair is then 8 bytes.
typedef double r2_d2 __attribute__((ext_vector_type(2)));
template <class T>
struct R2Pair;
template <>
struct R2Pair<double> {
    typedef double v2 __attribute__((ext_vector_type(2)));
    typedef v2 v2u __attribute__((aligned(8)));
};
template <>
struct R2Pair<float> {
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef v2 v2u __attribute__((aligned(4)));
};
// the pair that starts at element pointer p
template <class T, bool ODD>
__device__ __forceinline__ r2_d2 r2_ld2(const T* p) {
    typedef typename R2Pair<T>::v2 V;
    typedef typename R2Pair<T>::v2u VU;
    V v;
    if constexpr (ODD) v = (V)__builtin_nontemporal_load(reinterpret_cast<const VU*>(p));
    else v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    return r2_d2{(double)v.x, (double)v.y};
}
template <class Acc, bool ODD = false, class T = double>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_contig_v2(const T* __restrict__ x, u64 red, u64 nslices, u64 nsplit, Acc* __restrict__ part) {
    __shared__ Acc lds[R2_BLOCK];
    const u64 slice = blockIdx.y + (u64)gridDim.y * blockIdx.z;
    if (slice >= nslices) return;
    const u64 split = blockIdx.x, red2 = red >> 1;
    u64 chunk = (red2 + nsplit - 1) / nsplit;
    chunk = (chunk + R2_BLOCK - 1) / R2_BLOCK * R2_BLOCK;
    const u64 begin = split * chunk;
    u64 end = begin + chunk;
    if (end > red2) end = red2;
    const T* xs = x + slice * red;  // pair q starts at element 2 q
    Acc a;
    a.init();
    u64 r = begin + threadIdx.x;
    constexpr int U = 4;
    if (r + (U - 1) * R2_BLOCK < end) {
        r2_d2 cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = r2_ld2<T, ODD>(xs + 2 * (r + u * R2_BLOCK));
        for (; r + (2 * U - 1) * R2_BLOCK < end; r += U * R2_BLOCK) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = r2_ld2<T, ODD>(xs + 2 * (r + (U + u) * R2_BLOCK));
            {
                u64 k[8];
                double w[8];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    k[2 * u] = 2 * (r + u * R2_BLOCK);
                    k[2 * u + 1] = 2 * (r + u * R2_BLOCK) + 1;
                    w[2 * u] = cur[u].x;
                    w[2 * u + 1] = cur[u].y;
                }
                r2_fold8(a, k, w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
        {
            u64 k[8];
            double w[8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                k[2 * u] = 2 * (r + u * R2_BLOCK);
                k[2 * u + 1] = 2 * (r + u * R2_BLOCK) + 1;
                w[2 * u] = cur[u].x;
                w[2 * u + 1] = cur[u].y;
            }
            r2_fold8(a, k, w);
        }
        r += U * R2_BLOCK;
    }
    for (; r < end; r += R2_BLOCK) {
        const r2_d2 v = r2_ld2<T, ODD>(xs + 2 * r);
        a.add(2 * r, v.x);
        a.add(2 * r + 1, v.y);
    }
    if constexpr (ODD) {  // the leftover element of an odd slice: owned by the chunk that contains pair index red2
        const u64 owner = red2 / chunk < nsplit - 1 ? red2 / chunk : nsplit - 1;
        if ((red & 1) && split == owner && r == red2) a.add(red - 1, (double)__builtin_nontemporal_load(x + slice * red + red - 1));
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int s = R2_BLOCK / 2; s > 0; s >>= 1) {  // fixed tree
        if ((int)threadIdx.x < s) {
            Acc m = lds[threadIdx.x];
            m.merge(lds[threadIdx.x + s]);
            lds[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[slice * nsplit + split] = lds[0];
}

// many short contiguous slices (red < 256): a block stages a tile of consecutive slices in LDS with coalesced loads and thread t folds
// slice t in ascending order (reduce_kernels.hip: k_reduce_short) - one block per slice is half a million blocks for min(x,[],1) of a
// 32 x 524288 matrix
static constexpr int R2_SHORT_TILE = 4096;
__device__ __forceinline__ int r2_short_pad(int i) { return i + (i >> 5); }
template <class Acc, class T = double>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_short(const T* __restrict__ x, u64 red, u64 nslices, unsigned per_block, Acc* __restrict__ part) {
    __shared__ double tile[R2_SHORT_TILE + R2_SHORT_TILE / 32 + 1];
    const u64 s0 = (u64)blockIdx.x * per_block;
    const u64 ns = nslices - s0 < per_block ? nslices - s0 : per_block;
    const u64 count = ns * red;
    const T* src = x + s0 * red;
    for (u64 i = threadIdx.x; i < count; i += R2_BLOCK) tile[r2_short_pad((int)i)] = (double)__builtin_nontemporal_load(src + i);
    __syncthreads();
    if (threadIdx.x >= ns) return;
    Acc a;
    a.init();
    const int b = (int)(threadIdx.x * red);
    for (int r = 0; r < (int)red; ++r) a.add((u64)r, tile[r2_short_pad(b + r)]);
    part[s0 + threadIdx.x] = a;
}

// pre > 1: threads run along `pre` (coalesced), each walks its chunk of `red` in ascending order.  grid (ceil(pre / 256), nsplit, post)
template <class Acc, class T = double>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_strided(const T* __restrict__ x, u64 pre, u64 red, u64 nsplit, Acc* __restrict__ part) {
    const u64 i = (u64)blockIdx.x * R2_BLOCK + threadIdx.x;
    if (i >= pre) return;
    const u64 split = blockIdx.y, j = blockIdx.z;
    const u64 chunk = (red + nsplit - 1) / nsplit;
    const u64 begin = split * chunk;
    u64 end = begin + chunk;
    if (end > red) end = red;
    const T* xs = x + i + pre * red * j;
    Acc a;
    a.init();
    u64 r = begin;
    for (; r + 8 <= end; r += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (double)__builtin_nontemporal_load(xs + pre * (r + u));
        u64 k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = r + u;
        r2_fold8(a, k, v);
    }
    if (r < end) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + u < end) v[u] = (double)__builtin_nontemporal_load(xs + pre * (r + u));
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + u < end) a.add(r + u, v[u]);
    }
    part[(i + pre * j) * nsplit + split] = a;
}

// the same with 16-byte loads: a thread owns two adjacent lines (even `pre`, 16-byte aligned base).  As for sum(x,2)
// (reduce_kernels.hip) what decides the rate of these lock-step column walks is the number of blocks: three per CU.
// ODD: odd `pre` (or an element-aligned base): unaligned pairs, the last line alone in its pair.
template <class Acc, bool ODD = false, class T = double>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_strided_v2(const T* __restrict__ x, u64 pre, u64 red, u64 nsplit, unsigned win,
                                                            Acc* __restrict__ part) {
    const u64 i2 = (u64)blockIdx.x * win + threadIdx.x, pre2 = ODD ? (pre + 1) >> 1 : pre >> 1;  // balanced windows, their number a multiple of the XCD count (run_r2)
    if (threadIdx.x >= win || i2 >= pre2) return;
    const bool single = ODD && 2 * i2 + 1 >= pre;
    const u64 split = blockIdx.y, j = blockIdx.z;
    const u64 chunk = (red + nsplit - 1) / nsplit;
    const u64 begin = split * chunk;
    u64 end = begin + chunk;
    if (end > red) end = red;
    const T* const xo = x + 2 * i2 + pre * red * j;  // element offsets: with an odd `pre` the lines alternate between 16- and 8-byte alignment
    auto ldp = [&](u64 rr) -> r2_d2 {
        const T* q = xo + pre * rr;
        if (ODD && single) return r2_d2{(double)__builtin_nontemporal_load(q), 0.0};
        return r2_ld2<T, ODD>(q);
    };
    Acc a0, a1;
    a0.init();
    a1.init();
    u64 r = begin;
    if (r + 8 <= end) {  // two batches in flight: the next one is requested before the current one is folded
        r2_d2 cur[8], nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = ldp(r + u);
        for (; r + 16 <= end; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = ldp(r + 8 + u);
            {
                u64 k[8];
                double w0[8], w1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    k[u] = r + u;
                    w0[u] = cur[u].x;
                    w1[u] = cur[u].y;
                }
                r2_fold8(a0, k, w0);
                if (!single) r2_fold8(a1, k, w1);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
        {
            u64 k[8];
            double w0[8], w1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                k[u] = r + u;
                w0[u] = cur[u].x;
                w1[u] = cur[u].y;
            }
            r2_fold8(a0, k, w0);
            if (!single) r2_fold8(a1, k, w1);
        }
        r += 8;
    }
    if (r < end) {
        r2_d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + u < end) v[u] = ldp(r + u);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + u < end) {
                a0.add(r + u, v[u].x);
                if (!single) a1.add(r + u, v[u].y);
            }
    }
    const u64 line = 2 * i2 + pre * j;
    part[line * nsplit + split] = a0;
    if (!single) part[(line + 1) * nsplit + split] = a1;
}

// ---- stage 2: one wave per slice merges the chunks - lane l folds a contiguous run in chunk order, the 64 lane results merge in a
// fixed shuffle tree (lower lanes = earlier chunks on the left of every merge).  (First version: lane 0 folded the 64 lane
// results one after the other out of LDS - 28 us at 8192 slices, a third of the whole min / max call.)
template <class Acc>
__device__ __forceinline__ Acc r2_shfl_down(const Acc& a, int off) {
    static_assert(sizeof(Acc) % 4 == 0, "accumulators are shuffled word by word");
    constexpr int W = sizeof(Acc) / 4;
    int w[W];
    __builtin_memcpy(w, &a, sizeof(Acc));
#pragma unroll
    for (int i = 0; i < W; ++i) w[i] = __shfl_down(w[i], off, 64);
    Acc o;
    __builtin_memcpy(&o, w, sizeof(Acc));
    return o;
}
template <class Acc, class Fin>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_final(const Acc* __restrict__ part, u64 nslices, u64 nsplit, Fin fin) {
    const u64 slice = (u64)blockIdx.x * (R2_BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slice >= nslices) return;  // whole waves leave together
    Acc a;
    a.init();
    const u64 per = (nsplit + 63) / 64;
    const u64 b = (u64)lane * per;
    u64 e = b + per;
    if (e > nsplit) e = nsplit;
    for (u64 s = b; s < e; ++s) a.merge(part[slice * nsplit + s]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Acc o = r2_shfl_down(a, off);
        if ((lane & (2 * off - 1)) == 0) a.merge(o);  // lane, lane + off: the earlier chunks stay on the left
    }
    if (lane == 0) fin(slice, a);
}

// one thread per slice, its few chunks merged in chunk order (many slices with one or a handful of chunks each)
template <class Acc, class Fin>
__global__ void __launch_bounds__(R2_BLOCK) k_r2_final_flat(const Acc* __restrict__ part, u64 nslices, u64 nsplit, Fin fin) {
    const u64 slice = (u64)blockIdx.x * R2_BLOCK + threadIdx.x;
    if (slice >= nslices) return;
    Acc a;
    a.init();
    for (u64 s = 0; s < nsplit; ++s) a.merge(part[slice * nsplit + s]);
    fin(slice, a);
}

template <bool MAX>
struct ArgFin {
    double* values;
    double* indices;
    template <class Acc>
    __device__ __forceinline__ void operator()(u64 slice, const Acc& a) const {
        arg_decode<MAX>(a.key, a.idx, values + slice, indices + slice);
    }
};
struct StdFin {  // std.rs:918-935
    double* out;
    int population, omitnan;
    __device__ __forceinline__ void operator()(u64 slice, const MomAcc& a) const {
        double r;
        if ((a.nan > 0.0 && !omitnan) || a.n == 0.0) r = r2_nan();
        else {
            double var;
            if (population) var = a.m2 / a.n;
            else var = a.n > 1.0 ? a.m2 / (a.n - 1.0) : 0.0;
            r = sqrt(var > 0.0 ? var : 0.0);
        }
        out[slice] = r;
    }
};
struct TruthFin {
    double* out;
    int op, omitnan;  // RMHIP_TNNZ / RMHIP_TANY / RMHIP_TALL
    u64 red;
    __device__ __forceinline__ void operator()(u64 slice, const TruthAcc& a) const {
        double r;
        if (op == RMHIP_TNNZ) r = (double)(a.nz + a.nan);
        else if (op == RMHIP_TANY) r = (omitnan ? a.nz : a.nz + a.nan) > 0 ? 1.0 : 0.0;
        else r = (red - a.nz - a.nan) == 0 ? 1.0 : 0.0;  // all: NaNs are skipped in both modes; nothing left counts as true
        out[slice] = r;
    }
};

template <class Acc, class Fin, class T = double>
static int run_r2(Context* c, const T* x, size_t pre, size_t red, size_t post, const Fin& fin, const char* what) {
    if (pre == 0 || post == 0 || red == 0) return RMHIP_OK;
    ReducePlan p = plan_reduction(pre, red, post, c->num_cus, (unsigned)sizeof(T));
    if (!p.valid) return fail(RMHIP_ERR_UNSUPPORTED, "%s: geometry [%zu,%zu,%zu] exceeds launch limits", what, pre, red, post);
    u64 nsplit = p.nsplit;
    unsigned gx = p.gx;
    const bool wide = !p.contiguous && pre >= 512;
    const bool wide_odd = wide && ((pre & 1) != 0 || (((uintptr_t)x) & (2 * sizeof(T) - 1)) != 0);  // a pair of T
    unsigned win = R2_BLOCK, threads = R2_BLOCK;
    if (!p.contiguous) {  // these kernels keep up to 256 threads along `pre`
        if (wide) {  // as for sum(x,2): a window count that is a multiple of the XCD count pins every window to one XCD (reduce_plan.h)
            const StridedWidePlan w = plan_strided_wide(pre, red, post, c->num_cus, c->num_xcc, (unsigned)sizeof(T));
            gx = w.bx;
            win = w.win;
            threads = w.threads;
            nsplit = w.nsplit;
        } else {
            gx = (unsigned)ceil_div_u64(pre, R2_BLOCK);
            u64 want = ceil_div_u64((u64)c->num_cus * 8, (u64)gx * post);
            u64 max_split = ceil_div_u64(red, 16);
            nsplit = want < 1 ? 1 : (want > max_split ? max_split : want);
            nsplit = dealias_nsplit(red, nsplit, pre * sizeof(T), max_split);
            if (nsplit > 65535) nsplit = 65535;
        }
    }
    const bool short_a = p.contiguous && red < 256 && p.nslices >= 1024;
    if (short_a) nsplit = 1;
    const size_t nparts = (size_t)(p.nslices * nsplit);
    RMHIP_TRY(c->ensure_scratch(nparts * sizeof(Acc)));
    Acc* part = reinterpret_cast<Acc*>(c->scratch);
    if (short_a) {
        unsigned per_block = (unsigned)(R2_SHORT_TILE / red);
        if (per_block > R2_BLOCK) per_block = R2_BLOCK;
        hipLaunchKernelGGL((k_r2_short<Acc, T>), dim3((unsigned)ceil_div_u64(p.nslices, per_block)), dim3(R2_BLOCK), 0, c->stream, x, (u64)red, (u64)p.nslices,
                           per_block, part);
    } else if (p.contiguous && (red & 1) == 0 && red >= 4 * R2_BLOCK && (((uintptr_t)x) & (2 * sizeof(T) - 1)) == 0)
        hipLaunchKernelGGL((k_r2_contig_v2<Acc, false, T>), dim3((unsigned)nsplit, p.gy, p.gz), dim3(R2_BLOCK), 0, c->stream, x, (u64)red, (u64)p.nslices, nsplit, part);
    else if (p.contiguous && red >= 4 * R2_BLOCK)  // odd slice length or element-aligned base
        hipLaunchKernelGGL((k_r2_contig_v2<Acc, true, T>), dim3((unsigned)nsplit, p.gy, p.gz), dim3(R2_BLOCK), 0, c->stream, x, (u64)red, (u64)p.nslices, nsplit, part);
    else if (p.contiguous)
        hipLaunchKernelGGL((k_r2_contig<Acc, T>), dim3((unsigned)nsplit, p.gy, p.gz), dim3(R2_BLOCK), 0, c->stream, x, (u64)red, (u64)p.nslices, nsplit, part);
    else if (wide_odd)
        hipLaunchKernelGGL((k_r2_strided_v2<Acc, true, T>), dim3(gx, (unsigned)nsplit, (unsigned)post), dim3(threads), 0, c->stream, x, (u64)pre, (u64)red, nsplit, win, part);
    else if (wide)
        hipLaunchKernelGGL((k_r2_strided_v2<Acc, false, T>), dim3(gx, (unsigned)nsplit, (unsigned)post), dim3(threads), 0, c->stream, x, (u64)pre, (u64)red, nsplit, win, part);
    else
        hipLaunchKernelGGL((k_r2_strided<Acc, T>), dim3(gx, (unsigned)nsplit, (unsigned)post), dim3(R2_BLOCK), 0, c->stream, x, (u64)pre, (u64)red, nsplit, part);
    RMHIP_HIP_CHECK(hipGetLastError());
    // many slices with a handful of partials each: one thread per slice (a wave per slice would idle 50 of its lanes; 65536 slices
    // x 14 partials took 37 us that way)
    if ((nsplit <= 8 && p.nslices >= 1024) || (nsplit <= 32 && p.nslices >= 16384))
        hipLaunchKernelGGL((k_r2_final_flat<Acc, Fin>), dim3((unsigned)ceil_div_u64(p.nslices, R2_BLOCK)), dim3(R2_BLOCK), 0, c->stream, part,
                           (u64)p.nslices, nsplit, fin);
    else
        hipLaunchKernelGGL((k_r2_final<Acc, Fin>), dim3((unsigned)ceil_div_u64(p.nslices, R2_BLOCK / 64)), dim3(R2_BLOCK), 0, c->stream, part,
                           (u64)p.nslices, nsplit, fin);
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}

template <class T>
static int argreduce_any(Context* c, int op, int nan_mode, const T* x, size_t pre, size_t red, size_t post, double* values, double* indices) {
    const bool mx = op == RMHIP_RMAX;
    if (mx) {
        ArgFin<true> fin{values, indices};
        return nan_mode ? run_r2<ArgAcc<true, true>, ArgFin<true>, T>(c, x, pre, red, post, fin, "reduce_max_dim")
                        : run_r2<ArgAcc<true, false>, ArgFin<true>, T>(c, x, pre, red, post, fin, "reduce_max_dim");
    }
    ArgFin<false> fin{values, indices};
    return nan_mode ? run_r2<ArgAcc<false, true>, ArgFin<false>, T>(c, x, pre, red, post, fin, "reduce_min_dim")
                    : run_r2<ArgAcc<false, false>, ArgFin<false>, T>(c, x, pre, red, post, fin, "reduce_min_dim");
}
int launch_argreduce(Context* c, int op, int nan_mode, const double* x, size_t pre, size_t red, size_t post, double* values, double* indices) {
    return argreduce_any(c, op, nan_mode, x, pre, red, post, values, indices);
}
// f32 storage read in place (a precision-32 provider): the same accumulators on values widened in registers - no widened copy
int launch_argreduce_f32(Context* c, int op, int nan_mode, const float* x, size_t pre, size_t red, size_t post, double* values, double* indices) {
    return argreduce_any(c, op, nan_mode, x, pre, red, post, values, indices);
}
int launch_reduce_std(Context* c, int population, int nan_mode, const double* x, size_t pre, size_t red, size_t post, double* out) {
    StdFin fin{out, population, nan_mode};
    return run_r2<MomAcc>(c, x, pre, red, post, fin, "reduce_std");
}
int launch_reduce_std_f32(Context* c, int population, int nan_mode, const float* x, size_t pre, size_t red, size_t post, double* out) {
    StdFin fin{out, population, nan_mode};
    return run_r2<MomAcc, StdFin, float>(c, x, pre, red, post, fin, "reduce_std");
}
// image_normalize with more planes than special.hip's block layout takes (batch > 256): the tensor is a batch x plane matrix and the
// statistics are a moments reduction along its second dimension - many short-stride lines, the strided kernels' home ground.
// stats[b] = mean, stats[batch + b] = 1 / sqrt(M2 / plane + eps) (0 when that is not positive: simple_provider.rs:7961-7963)
struct PlaneStatFin {
    double* stats;
    u64 batch;
    double plane, eps;
    __device__ __forceinline__ void operator()(u64 slice, const MomAcc& a) const {
        double mean = a.mean, inv = 0.0;
        if (a.nan > 0.0) mean = r2_nan();  // the CPU's running sum turns NaN; its sigma test then fails and inv stays 0
        else {
            const double sigma = sqrt(a.m2 / plane + eps);
            inv = sigma > 0.0 ? 1.0 / sigma : 0.0;
        }
        stats[slice] = mean;
        stats[batch + slice] = inv;
    }
};
int launch_plane_stats(Context* c, const double* x, size_t batch, size_t plane, double eps, double* stats) {
    PlaneStatFin fin{stats, (u64)batch, (double)plane, eps};
    return run_r2<MomAcc>(c, x, batch, plane, 1, fin, "image_normalize");
}
// mean(x, dim) and mean(x .* x, dim) from one pass over x (the first, full-size step of reduce_moments_nd)
struct MomentsFin {
    double* mean;
    double* ex2;
    double count;
    __device__ __forceinline__ void operator()(u64 slice, const SqAcc& a) const {
        mean[slice] = a.s / count;
        ex2[slice] = a.q / count;
    }
};
int launch_reduce_moments(Context* c, const double* x, size_t pre, size_t red, size_t post, double* mean, double* ex2) {
    MomentsFin fin{mean, ex2, (double)red};
    return run_r2<SqAcc>(c, x, pre, red, post, fin, "reduce_moments_nd");
}
int launch_reduce_truth(Context* c, int op, int omit_nan, const double* x, size_t pre, size_t red, size_t post, double* out) {
    TruthFin fin{out, op, omit_nan, (u64)red};
    return run_r2<TruthAcc>(c, x, pre, red, post, fin, "reduce_truth");
}
int launch_reduce_truth_f32(Context* c, int op, int omit_nan, const float* x, size_t pre, size_t red, size_t post, double* out) {
    TruthFin fin{out, op, omit_nan, (u64)red};
    return run_r2<TruthAcc, TruthFin, float>(c, x, pre, red, post, fin, "reduce_truth");
}

// ---- cumulative sum / product along one dimension (cumsum.rs:559-650, cumprod.rs:581-670) ----------------------------------------
// includenan: from the first NaN on every output is NaN (what the running value does by itself; the outputs are canonical
// NaNs like the CPU's); omitnan: a NaN leaves the running value unchanged.  Reverse runs from the last element.
template <bool PROD>
__device__ __forceinline__ double scan_op(double a, double b) { return PROD ? a * b : a + b; }
template <bool PROD>
__device__ __forceinline__ double scan_in(double v, int omit) { return (omit && v != v) ? (PROD ? 1.0 : 0.0) : v; }
__device__ __forceinline__ double scan_out(double v) { return v != v ? r2_nan() : v; }

// pre > 1 (or short lines): one thread per line, the CPU's own sequence of operations - bit-identical results.  Threads run along
// `pre`, so every step is a coalesced read and write.  The chain itself is cheap (one dependent operation per element); what the
// kernel waits for is memory, so a thread keeps two batches of U loads in flight - the next batch is requested before the current
// one is consumed - and few lines are spread over many small blocks (BLOCK = 64: 8192 lines reach 128 CUs instead of 32).
template <bool PROD, int BLOCK, int U>
__global__ void __launch_bounds__(BLOCK) k_scan_lines(const double* __restrict__ x, double* __restrict__ y, u64 pre, u64 len, u64 post, int reverse,
                                                      int omit) {
    const u64 line = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (line >= pre * post) return;
    const u64 i = line % pre, j = line / pre;
    const u64 base = i + pre * len * j;
    auto at = [&](u64 k) { return base + pre * (reverse ? len - 1 - k : k); };
    double run = PROD ? 1.0 : 0.0;
    double cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = (u64)u < len ? __builtin_nontemporal_load(x + at(u)) : 0.0;
    for (u64 k0 = 0; k0 < len; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 k = k0 + U + u;
            nxt[u] = k < len ? __builtin_nontemporal_load(x + at(k)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u64 k = k0 + u;
            if (k < len) {
                run = scan_op<PROD>(run, scan_in<PROD>(cur[u], omit));
                __builtin_nontemporal_store(scan_out(run), y + at(k));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

// Few long strided lines (8192 x 8192 along dim 2: 8192 lines, 128 waves): one thread per line cannot keep enough loads in flight to
// cover the memory latency - 1.09 ms where the bytes would take 0.2.  Here a block of eight waves owns 64 adjacent lines and walks them
// in tiles of 64 steps: ALL waves fetch the next tile (64 lines x 64 steps, 32 KiB, coalesced rows of 512 bytes) while wave 0 runs the
// 64 chains over the current one in LDS - still each line's own left-to-right sequence, so the result stays bit-identical to the
// CPU's - then all waves write the tile out and drop the fetched one into the cells they just emptied (no barrier in between: a thread
// stores and refills exactly its own cells).
// LINES = 64, or 32 when 64 would leave CUs without a block (8192 lines: 128 blocks on 256 CUs).
static constexpr int ST_STEPS = 64, ST_THREADS = 512;
// Few LONG lines (512 lines of 65536 steps: 16 workgroups, each a chain of 65536 - 3 ms) are cut into chunks along the line
// (blockIdx.z): a first launch (TOT) only forms every chunk's total, k_scan_carries turns the totals of a line into the value carried
// into each chunk, and the second launch scans every chunk from its carry - the scheme of the contiguous path below.  chunk_steps is
// a multiple of ST_STEPS; one chunk (carries == nullptr) is the single-launch form.
template <bool PROD, int LINES, bool TOT>
__global__ void __launch_bounds__(ST_THREADS) k_scan_lines_staged(const double* __restrict__ x, double* __restrict__ y, u64 pre, u64 len, u64 post,
                                                                  int reverse, int omit, u64 chunk_steps, u64 nchunks,
                                                                  const double* __restrict__ carries, double* __restrict__ totals) {
    __shared__ double buf[ST_STEPS][LINES];
    (void)post;
    constexpr int RP = ST_THREADS / LINES;   // rows of a tile the block touches per pass (8 or 16)
    constexpr int ROWS = ST_STEPS / RP;      // passes = rows per thread (8 or 4)
    const int line = threadIdx.x % LINES, row0 = threadIdx.x / LINES;
    const u64 i = (u64)blockIdx.x * LINES + line, j = blockIdx.y;
    const bool live = i < pre;
    const u64 base = i + pre * len * j;
    const u64 chunk = blockIdx.z, kbeg = chunk * chunk_steps;
    const u64 kend = kbeg + chunk_steps < len ? kbeg + chunk_steps : len;
    const u64 ntiles = (kend - kbeg + ST_STEPS - 1) / ST_STEPS;
    const double ident = PROD ? 1.0 : 0.0;
    double regs[ROWS], regs2[ROWS];  // tiles t + 1 and t + 2 in flight (few blocks run this kernel: a tile takes its full memory latency)
    auto at = [&](u64 k) { return base + pre * (reverse ? len - 1 - k : k); };
    auto gload = [&](u64 t, double (&r)[ROWS]) {
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const u64 k = kbeg + t * ST_STEPS + (u64)(u * RP + row0);
            r[u] = (live && k < kend) ? __builtin_nontemporal_load(x + at(k)) : ident;
        }
    };
    gload(0, regs);
#pragma unroll
    for (int u = 0; u < ROWS; ++u) buf[u * RP + row0][line] = regs[u];
    if (ntiles > 1) gload(1, regs);
    __syncthreads();
    double run = (carries && live && threadIdx.x < LINES) ? carries[(i + pre * j) * nchunks + chunk] : ident;
    for (u64 t = 0; t < ntiles; ++t) {
        if (t + 2 < ntiles) gload(t + 2, regs2);  // in flight while the first LINES threads scan this tile and the next
        if (threadIdx.x < LINES) {
            double v[ST_STEPS];
#pragma unroll
            for (int k = 0; k < ST_STEPS; ++k) v[k] = scan_in<PROD>(buf[k][line], omit);  // (the NaN policies stay off the dependent chain)
#pragma unroll
            for (int k = 0; k < ST_STEPS; ++k) {  // rows beyond `len` hold the identity: they leave the running value alone and are not stored
                run = scan_op<PROD>(run, v[k]);
                v[k] = run;
            }
#pragma unroll
            for (int k = 0; k < ST_STEPS; ++k) buf[k][line] = scan_out(v[k]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const u64 k = kbeg + t * ST_STEPS + (u64)(u * RP + row0);
            if (!TOT && live && k < kend) __builtin_nontemporal_store(buf[u * RP + row0][line], y + at(k));
            buf[u * RP + row0][line] = regs[u];  // tile t + 1 (garbage after the last one: nobody reads it)
            regs[u] = regs2[u];
        }
        __syncthreads();
    }
    if (TOT && live && threadIdx.x < LINES) totals[(i + pre * j) * nchunks + chunk] = run;
}

// pre == 1, SHORT contiguous lines (len < 256: cumsum(x,1) of a 32 x N matrix): one thread per line would read with a stride of one
// line between lanes.  A block stages a tile of consecutive lines - one contiguous piece of memory - in LDS with coalesced loads
// (padded: the per-thread walks fall on distinct banks), thread t scans line t in place - the CPU's own sequence - and the tile goes
// back the same way.
static constexpr int SS_TILE = 4096;
__device__ __forceinline__ int ss_pad(int i) { return i + (i >> 5); }
template <bool PROD>
__global__ void __launch_bounds__(R2_BLOCK) k_scan_short(const double* __restrict__ x, double* __restrict__ y, u64 len, u64 nlines, unsigned per_block,
                                                         int reverse, int omit) {
    __shared__ double tile[SS_TILE + SS_TILE / 32 + 1];
    const u64 l0 = (u64)blockIdx.x * per_block;
    const u64 nl = nlines - l0 < per_block ? nlines - l0 : per_block;
    const u64 count = nl * len;
    const double* src = x + l0 * len;
    double* dst = y + l0 * len;
    for (u64 i = threadIdx.x; i < count; i += R2_BLOCK) tile[ss_pad((int)i)] = scan_in<PROD>(__builtin_nontemporal_load(src + i), omit);
    __syncthreads();
    if (threadIdx.x < nl) {
        const int b = (int)(threadIdx.x * len);
        double run = PROD ? 1.0 : 0.0;
        for (int k = 0; k < (int)len; ++k) {
            const int idx = ss_pad(b + (reverse ? (int)len - 1 - k : k));
            run = scan_op<PROD>(run, tile[idx]);
            tile[idx] = scan_out(run);
        }
    }
    __syncthreads();
    for (u64 i = threadIdx.x; i < count; i += R2_BLOCK) __builtin_nontemporal_store(tile[ss_pad((int)i)], dst + i);
}

// pre == 1, long lines: three passes over chunks of SCAN_CHUNK elements - chunk totals, a serial scan of the totals per line,
// then every chunk scans itself from its carry (a line of one chunk skips the first two).  Inside a chunk a block scans tiles of
// 256 x 8 elements: the tile is read COALESCED (lane l of a load takes element l + 256 u), turned through LDS so that a thread
// owns eight consecutive elements, scanned serially in the thread, the thread totals scanned by shuffles inside a wave and through
// LDS across the four waves, and written back the same way.  (First version: every thread loaded its own eight consecutive
// elements - 64-byte strides between lanes; the counters showed 1.4-1.5x the algorithmic bytes, profiles/r03_pmc_summary.json.)
// The grouping differs from the CPU's left-to-right sequence: results agree to rounding (exactly, for integer-valued data below
// 2^53).
static constexpr int SCAN_PER_THREAD = 8;
static constexpr int SCAN_TILE = R2_BLOCK * SCAN_PER_THREAD;
static constexpr u64 SCAN_CHUNK = 8 * SCAN_TILE;  // 16384 elements: a 6.7e7-element vector gives 4096 blocks (with 32 tiles per block, 1024 blocks, that vector took 508 us against 332)
static constexpr int SCAN_LDS = SCAN_TILE + SCAN_TILE / 8;  // one pad per eight: the transposed accesses spread over the banks
__device__ __forceinline__ int scan_pad(int i) { return i + (i >> 3); }
static_assert(R2_BLOCK == 256, "the scan kernels assume four waves per block");

// exclusive prefix of the threads' totals in thread order, and the block total
template <bool PROD>
__device__ __forceinline__ double block_exclusive(double total, double* wsum /*[4]*/, double* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double n = __shfl_up(incl, off, 64);
        if (lane >= off) incl = scan_op<PROD>(n, incl);
    }
    double prev = __shfl_up(incl, 1, 64);
    if (lane == 0) prev = PROD ? 1.0 : 0.0;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    double lead = PROD ? 1.0 : 0.0, all = PROD ? 1.0 : 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const double s = wsum[w];
        if (w < wave) lead = scan_op<PROD>(lead, s);
        all = scan_op<PROD>(all, s);
    }
    __syncthreads();  // wsum is reused by the next tile
    *block_total = all;
    return scan_op<PROD>(lead, prev);
}

// the tile's elements (identity beyond `e`), coalesced, into v[] = the thread's eight consecutive ones
template <bool PROD>
__device__ __forceinline__ void scan_tile_in(const double* __restrict__ xs, u64 len, u64 tile, u64 e, int reverse, int omit, double* tl,
                                             double (&v)[SCAN_PER_THREAD]) {
    const int t = threadIdx.x;
    double in[SCAN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) {
        const u64 k = tile + (u64)(t + R2_BLOCK * u);
        in[u] = PROD ? 1.0 : 0.0;
        if (k < e) in[u] = scan_in<PROD>(__builtin_nontemporal_load(xs + (reverse ? len - 1 - k : k)), omit);
    }
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) tl[scan_pad(t + R2_BLOCK * u)] = in[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) v[u] = tl[scan_pad(t * SCAN_PER_THREAD + u)];
}

// pass 1: the chunk's total.  Any grouping will do - the carries only have to be right to rounding - so this is a plain streaming
// reduction: eight independent accumulators per thread over coalesced loads, folded in a fixed order (deterministic).
template <bool PROD>
__global__ void __launch_bounds__(R2_BLOCK) k_scan_chunk_totals(const double* __restrict__ x, u64 len, u64 nchunks, u64 nlines, int reverse, int omit,
                                                                double* __restrict__ totals) {
    __shared__ double wsum[4];
    const u64 chunk = blockIdx.x, line = blockIdx.y + (u64)gridDim.y * blockIdx.z;  // (more than 65535 lines spill into z)
    if (line >= nlines) return;
    const u64 b = chunk * SCAN_CHUNK;
    u64 e = b + SCAN_CHUNK;
    if (e > len) e = len;
    const double* xs = x + line * len;
    double acc[SCAN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) acc[u] = PROD ? 1.0 : 0.0;
    for (u64 tile = b; tile < e; tile += SCAN_TILE) {
#pragma unroll
        for (int u = 0; u < SCAN_PER_THREAD; ++u) {
            const u64 k = tile + (u64)(threadIdx.x + R2_BLOCK * u);
            if (k < e) acc[u] = scan_op<PROD>(acc[u], scan_in<PROD>(__builtin_nontemporal_load(xs + (reverse ? len - 1 - k : k)), omit));
        }
    }
    double tot = acc[0];
#pragma unroll
    for (int u = 1; u < SCAN_PER_THREAD; ++u) tot = scan_op<PROD>(tot, acc[u]);
    double bt;
    (void)block_exclusive<PROD>(tot, wsum, &bt);
    if (threadIdx.x == 0) totals[line * nchunks + chunk] = bt;
}
// pass 2: totals[c] <- carry into chunk c.  One block per line, 256 totals per trip, the same shuffle scan as inside a tile.  (First
// version: one THREAD per line walking its chunks - a 6.7e7-element vector has 1024 of them and the walk, a dependent global
// round trip per chunk, took longer than both streaming passes together.)
template <bool PROD>
__global__ void __launch_bounds__(R2_BLOCK) k_scan_carries(double* __restrict__ totals, u64 nchunks, u64 nlines) {
    __shared__ double wsum[4];
    double* tl = totals + (u64)blockIdx.x * nchunks;
    double carry = PROD ? 1.0 : 0.0;
    for (u64 c0 = 0; c0 < nchunks; c0 += R2_BLOCK) {
        const u64 c = c0 + threadIdx.x;
        const double v = c < nchunks ? tl[c] : (PROD ? 1.0 : 0.0);
        double bt;
        const double excl = block_exclusive<PROD>(v, wsum, &bt);
        if (c < nchunks) tl[c] = scan_op<PROD>(carry, excl);
        carry = scan_op<PROD>(carry, bt);
    }
}
// carries == nullptr: one chunk per line, nothing carried in
template <bool PROD>
__global__ void __launch_bounds__(R2_BLOCK) k_scan_chunks(const double* __restrict__ x, double* __restrict__ y, u64 len, u64 nchunks, u64 nlines,
                                                          int reverse, int omit, const double* __restrict__ carries) {
    __shared__ double tl[SCAN_LDS];
    __shared__ double wsum[4];
    const int t = threadIdx.x;
    const u64 chunk = blockIdx.x, line = blockIdx.y + (u64)gridDim.y * blockIdx.z;
    if (line >= nlines) return;
    const u64 b = chunk * SCAN_CHUNK;
    u64 e = b + SCAN_CHUNK;
    if (e > len) e = len;
    const double* xs = x + line * len;
    double* ys = y + line * len;
    double carry = carries ? carries[line * nchunks + chunk] : (PROD ? 1.0 : 0.0);
    for (u64 tile = b; tile < e; tile += SCAN_TILE) {
        double v[SCAN_PER_THREAD];
        scan_tile_in<PROD>(xs, len, tile, e, reverse, omit, tl, v);
        double tot = PROD ? 1.0 : 0.0;
#pragma unroll
        for (int u = 0; u < SCAN_PER_THREAD; ++u) {
            tot = scan_op<PROD>(tot, v[u]);
            v[u] = tot;  // inclusive within the thread
        }
        double bt;
        const double excl = block_exclusive<PROD>(tot, wsum, &bt);  // (barriers: every thread has read its eight by now)
        const double lead = scan_op<PROD>(carry, excl);
#pragma unroll
        for (int u = 0; u < SCAN_PER_THREAD; ++u) tl[scan_pad(t * SCAN_PER_THREAD + u)] = scan_out(scan_op<PROD>(lead, v[u]));
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SCAN_PER_THREAD; ++u) {
            const u64 k = tile + (u64)(t + R2_BLOCK * u);
            if (k < e) __builtin_nontemporal_store(tl[scan_pad(t + R2_BLOCK * u)], ys + (reverse ? len - 1 - k : k));
        }
        __syncthreads();  // the next tile overwrites tl
        carry = scan_op<PROD>(carry, bt);
    }
}

int launch_cumulative(Context* c, int prod, int reverse, int omit, const double* x, size_t pre, size_t len, size_t post, double* y) {
    if (pre == 0 || len == 0 || post == 0) return RMHIP_OK;
    const size_t lines = pre * post;
    if (pre > 1) {  // lines along a strided dimension: every line is the CPU's own chain - except the chunked form below (few lines of >= 4096 steps:
                    // chunk totals carried forward, equal to rounding; include/rmhip.h states the contract and the overflow caveat)
        if (pre >= 8 && len >= 256 && lines < (size_t)c->num_cus * 256 && post <= 65535) {  // few long lines: staged tiles
            const bool half = pre < 64 || ceil_div_u64(pre, 64) * post < (u64)c->num_cus;   // 64 lines per block would leave CUs (or lanes) idle
            const u64 line_blocks = ceil_div_u64(pre, half ? 32 : 64) * post;
            u64 nchunks = 1, chunk_steps = ceil_div_u64(len, ST_STEPS) * ST_STEPS;
            if (line_blocks * 4 <= (u64)c->num_cus && len >= 4096) {  // a quarter of the CUs or fewer would run chains of len steps: chunks along the line
                u64 want = ceil_div_u64((u64)c->num_cus * 4, line_blocks), most = len / 1024;  // chunks of >= 1024 steps
                if (want > most) want = most;
                if (want > 65535) want = 65535;
                if (want > 1) {
                    chunk_steps = ceil_div_u64(ceil_div_u64(len, want), ST_STEPS) * ST_STEPS;
                    nchunks = ceil_div_u64(len, chunk_steps);
                }
            }
            double* totals = nullptr;
            if (nchunks > 1) {
                RMHIP_TRY(c->ensure_scratch(sizeof(double) * lines * nchunks));
                totals = c->scratch;
            }
            const dim3 sgrid((unsigned)ceil_div_u64(pre, half ? 32 : 64), (unsigned)post, (unsigned)nchunks);
#define RMHIP_STAGED(P, L, T, CAR, TOTP)                                                                                                       \
    hipLaunchKernelGGL((k_scan_lines_staged<P, L, T>), sgrid, dim3(ST_THREADS), 0, c->stream, x, y, (u64)pre, (u64)len, (u64)post, reverse, omit, \
                       chunk_steps, nchunks, (const double*)(CAR), (double*)(TOTP))
#define RMHIP_STAGED_PL(T, CAR, TOTP)                    \
    do {                                                 \
        if (prod && half) RMHIP_STAGED(true, 32, T, CAR, TOTP);   \
        else if (prod) RMHIP_STAGED(true, 64, T, CAR, TOTP);      \
        else if (half) RMHIP_STAGED(false, 32, T, CAR, TOTP);     \
        else RMHIP_STAGED(false, 64, T, CAR, TOTP);               \
    } while (0)
            if (nchunks > 1) {
                RMHIP_STAGED_PL(true, nullptr, totals);
                if (prod) hipLaunchKernelGGL(k_scan_carries<true>, dim3((unsigned)lines), dim3(R2_BLOCK), 0, c->stream, totals, nchunks, (u64)lines);
                else hipLaunchKernelGGL(k_scan_carries<false>, dim3((unsigned)lines), dim3(R2_BLOCK), 0, c->stream, totals, nchunks, (u64)lines);
                c->tel.kernel_launches += 2;
            }
            RMHIP_STAGED_PL(false, totals, nullptr);
#undef RMHIP_STAGED_PL
#undef RMHIP_STAGED
        } else if (lines >= (size_t)c->num_cus * 1024) {  // plenty of lines: four waves per block, eight loads deep
            const unsigned grid = (unsigned)ceil_div_u64(lines, 256);
            if (prod) hipLaunchKernelGGL((k_scan_lines<true, 256, 8>), dim3(grid), dim3(256), 0, c->stream, x, y, (u64)pre, (u64)len, (u64)post, reverse, omit);
            else hipLaunchKernelGGL((k_scan_lines<false, 256, 8>), dim3(grid), dim3(256), 0, c->stream, x, y, (u64)pre, (u64)len, (u64)post, reverse, omit);
        } else {
            const unsigned grid = (unsigned)ceil_div_u64(lines, 64);
            if (prod) hipLaunchKernelGGL((k_scan_lines<true, 64, 32>), dim3(grid), dim3(64), 0, c->stream, x, y, (u64)pre, (u64)len, (u64)post, reverse, omit);
            else hipLaunchKernelGGL((k_scan_lines<false, 64, 32>), dim3(grid), dim3(64), 0, c->stream, x, y, (u64)pre, (u64)len, (u64)post, reverse, omit);
        }
        RMHIP_HIP_CHECK(hipGetLastError());
        c->tel.kernel_launches++;
        return RMHIP_OK;
    }
    if (len < 256) {  // short contiguous lines: a tile of lines per block, a thread per line
        unsigned per_block = (unsigned)(SS_TILE / len);
        if (per_block > R2_BLOCK) per_block = R2_BLOCK;
        const unsigned grid = (unsigned)ceil_div_u64(lines, per_block);
        if (prod) hipLaunchKernelGGL(k_scan_short<true>, dim3(grid), dim3(R2_BLOCK), 0, c->stream, x, y, (u64)len, (u64)lines, per_block, reverse, omit);
        else hipLaunchKernelGGL(k_scan_short<false>, dim3(grid), dim3(R2_BLOCK), 0, c->stream, x, y, (u64)len, (u64)lines, per_block, reverse, omit);
        RMHIP_HIP_CHECK(hipGetLastError());
        c->tel.kernel_launches++;
        return RMHIP_OK;
    }
    // long contiguous lines: blocked scans over chunks (a line of one chunk: the last pass only)
    const u64 gy = lines < 65535 ? lines : 65535, gz = ceil_div_u64(lines, gy);
    if (gz > 65535) return fail(RMHIP_ERR_UNSUPPORTED, "cumulative scan: %zu contiguous lines exceed the launch limits", lines);
    const u64 nchunks = ceil_div_u64(len, SCAN_CHUNK);
    RMHIP_TRY(c->ensure_scratch(lines * nchunks * sizeof(double)));
    double* totals = c->scratch;
    const dim3 grid((unsigned)nchunks, (unsigned)gy, (unsigned)gz);
    const double* carries = nchunks > 1 ? totals : nullptr;
    if (prod) {
        if (carries) {
            hipLaunchKernelGGL(k_scan_chunk_totals<true>, grid, dim3(R2_BLOCK), 0, c->stream, x, (u64)len, nchunks, (u64)lines, reverse, omit, totals);
            hipLaunchKernelGGL(k_scan_carries<true>, dim3((unsigned)lines), dim3(R2_BLOCK), 0, c->stream, totals, nchunks, (u64)lines);
        }
        hipLaunchKernelGGL(k_scan_chunks<true>, grid, dim3(R2_BLOCK), 0, c->stream, x, y, (u64)len, nchunks, (u64)lines, reverse, omit, carries);
    } else {
        if (carries) {
            hipLaunchKernelGGL(k_scan_chunk_totals<false>, grid, dim3(R2_BLOCK), 0, c->stream, x, (u64)len, nchunks, (u64)lines, reverse, omit, totals);
            hipLaunchKernelGGL(k_scan_carries<false>, dim3((unsigned)lines), dim3(R2_BLOCK), 0, c->stream, totals, nchunks, (u64)lines);
        }
        hipLaunchKernelGGL(k_scan_chunks<false>, grid, dim3(R2_BLOCK), 0, c->stream, x, y, (u64)len, nchunks, (u64)lines, reverse, omit, carries);
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += carries ? 3 : 1;
    return RMHIP_OK;
}

}  // namespace rmhip
