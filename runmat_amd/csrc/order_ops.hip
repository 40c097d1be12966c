// order_ops.hip -- the order-statistics hooks of the reduction / sorting builtins
//   cummin_scan / cummax_scan     crates/runmat-accelerate-api/src/lib.rs:2918-2935   (CPU semantics cummin.rs:719-876, cummax.rs)
//   diff_dim                      lib.rs:2596-2603    (diff.rs:439-507; the provider form simple_provider.rs:6474-6496)
//   reduce_median(_dim)           lib.rs:2833-2845    (median.rs:644-741; simple_provider.rs:7167-7270)
//   sort_dim                      lib.rs:2358-2366    (sorting_sets/sort.rs:413-468, 538-574)
// Integer / comparison work on f64 data: every result is a copy of an input element, a position, or one rounded operation
// (a difference, the mean of two middle elements) - bit-exact against the oracle.
//   * running extremes: the (value, first position, NaN state) triple is associative, so lines are scanned in any grouping - a wave per
//     contiguous line piece (shuffle scan over rows of 64), a thread per strided line piece (coalesced across the lines), long lines cut
//     into chunks with a carry pass in between.
//   * sort / median: every line becomes (u64 key, u32 position) pairs in a workspace padded to a power of two - the key orders the
//     values exactly as compare_real_values does (NaNs last / first, |x| then x, -0 == +0) and the position breaks ties, which IS the
//     stable order - and is sorted by a bitonic network: all steps below 2048 elements in LDS, the wider ones in global passes.
#include <algorithm>
#include <cstring>
#include <limits>

#include "common.h"

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c);                                             \
    NarrowScope _ns(c)

namespace rmhip {
namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

struct Lines {  // element (line, k) of the operand sits at (line / pre) * pre * len + line % pre + k * pre
    u64 pre, len, post;
};

std::vector<size_t> matrix_shape(const std::vector<size_t>& s) {
    if (s.empty()) return {1, 1};
    if (s.size() == 1) return {s[0], 1};
    return s;
}

__device__ __forceinline__ double quiet_nan() { return __longlong_as_double(0x7ff8000000000000ll); }

// ---------------------------------------------------------------------------------------------------------------------------------
// cummin / cummax
// ---------------------------------------------------------------------------------------------------------------------------------
struct Ext {
    double v;
    u32 pos;   // 1-based position along the dimension
    int kind;  // 0 nothing yet, 1 a number, 2 fixed at a NaN (include mode)
};

template <bool MAX>
__device__ __forceinline__ Ext ext_combine(const Ext& l, const Ext& r) {  // l covers the earlier part of the scan
    if (l.kind == 2 || r.kind == 0) return l;
    if (l.kind == 0 || r.kind == 2) return r;
    return (MAX ? r.v > l.v : r.v < l.v) ? r : l;  // strict: the first occurrence keeps a tie
}
__device__ __forceinline__ Ext ext_of(double x, u32 pos, int omit) {
    Ext e;
    e.v = x;
    e.pos = pos;
    e.kind = (x != x) ? (omit ? 0 : 2) : 1;
    return e;
}
__device__ __forceinline__ void ext_store(const Ext& e, double* v, double* idx) {
    __builtin_nontemporal_store(e.kind == 1 ? e.v : quiet_nan(), v);
    __builtin_nontemporal_store(e.kind == 0 ? quiet_nan() : (double)e.pos, idx);
}
__device__ __forceinline__ Ext ext_shfl(const Ext& e, int src_lane) {
    Ext o;
    o.v = __shfl(e.v, src_lane);
    o.pos = __shfl(e.pos, src_lane);
    o.kind = __shfl(e.kind, src_lane);
    return o;
}
__device__ __forceinline__ Ext ext_shfl_up(const Ext& e, int d) {
    Ext o;
    o.v = __shfl_up(e.v, d);
    o.pos = __shfl_up(e.pos, d);
    o.kind = __shfl_up(e.kind, d);
    return o;
}

struct ExtSum {  // chunk summaries / carries
    double* v;
    u32* pos;
    int* kind;
};

// pre == 1: one wave per (line, chunk); MODE 0 = summary of the chunk only, 1 = scan from the carry and write.  WE rows of 64
// consecutive elements per trip: loaded together (every access a contiguous 512 bytes of the line), scanned by WE independent shuffle
// chains, their totals combined in order.  Inside the wave the state is ONE ordered 64-bit key and the position: the value's bits
// mapped monotonically (inverted for max, both zeros on one key), 0 = "fixed at a NaN" (beats everything, the earlier one on a tie)
// and ~0 = "nothing yet" (loses to everything) - combine is `right key < left key ? right : left`, three shuffles per step.
constexpr int WE = 4;
constexpr u64 KEY_ZERO = 0x8000000000000000ull;
template <bool MAX>
__device__ __forceinline__ u64 ext_key(double x, int omit) {
    if (x != x) return omit ? ~0ull : 0ull;
    if (x == 0.0) x = 0.0;
    const u64 u = (u64)__double_as_longlong(x);
    const u64 b = (u >> 63) ? ~u : (u | KEY_ZERO);
    return MAX ? ~b : b;
}
template <bool MAX>
__device__ __forceinline__ u64 ext_key_of(const Ext& e) { return e.kind == 0 ? ~0ull : (e.kind == 2 ? 0ull : ext_key<MAX>(e.v, 0)); }
template <bool MAX>
__device__ __forceinline__ Ext ext_from_key(u64 key, u32 pos, const double* line_data) {
    Ext e;
    e.pos = pos;
    e.kind = key == 0ull ? 2 : (key == ~0ull ? 0 : 1);
    const u64 b = MAX ? ~key : key;
    e.v = 0.0;
    if (e.kind == 1) e.v = b == KEY_ZERO ? line_data[pos - 1] : __longlong_as_double((long long)((b >> 63) ? (b & ~KEY_ZERO) : ~b));  // a zero: its own sign
    return e;
}
template <bool MAX, int MODE>
__global__ void __launch_bounds__(256) k_cumext_wave(const double* __restrict__ x, double* __restrict__ vals, double* __restrict__ idxs, u64 len,
                                                     u64 nlines, u64 chunk_len, u64 nchunks, int reverse, int omit, ExtSum sum) {
    const u64 w = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= nlines * nchunks) return;
    const u64 line = w / nchunks, ch = w % nchunks;
    const u64 s0 = ch * chunk_len, s1 = (s0 + chunk_len < len) ? s0 + chunk_len : len;
    const double* xs = x + line * len;
    u64 ckey = ~0ull;
    u32 cpos = 0;
    if (MODE == 1 && nchunks > 1) {
        Ext c0;
        c0.v = sum.v[w];
        c0.pos = sum.pos[w];
        c0.kind = sum.kind[w];
        ckey = ext_key_of<MAX>(c0);
        cpos = c0.pos;
    }
    for (u64 s = s0; s < s1; s += 64 * WE) {
        u64 key[WE];
        u32 pos[WE];
#pragma unroll
        for (int u = 0; u < WE; ++u) {
            const u64 me = s + (u64)u * 64 + lane;
            const u64 k = reverse ? len - 1 - me : me;
            key[u] = ~0ull;
            pos[u] = 0;
            if (me < s1) {
                key[u] = ext_key<MAX>(__builtin_nontemporal_load(xs + k), omit);
                pos[u] = (u32)(k + 1);
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int u = 0; u < WE; ++u) {
                const u64 uk = __shfl_up(key[u], d);
                const u32 up = __shfl_up(pos[u], d);
                if (lane >= d && !(key[u] < uk)) {  // the earlier one unless the later is strictly better
                    key[u] = uk;
                    pos[u] = up;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WE; ++u) {
            const u64 me = s + (u64)u * 64 + lane;
            const u64 k = reverse ? len - 1 - me : me;
            if (!(key[u] < ckey)) {
                key[u] = ckey;
                pos[u] = cpos;
            }
            if (MODE == 1 && me < s1) ext_store(ext_from_key<MAX>(key[u], pos[u], xs), vals + line * len + k, idxs + line * len + k);
            ckey = __shfl(key[u], 63);
            cpos = __shfl(pos[u], 63);
        }
    }
    if (MODE == 0 && lane == 0) {
        const Ext e = ext_from_key<MAX>(ckey, cpos, xs);
        sum.v[w] = e.v;
        sum.pos[w] = e.pos;
        sum.kind[w] = e.kind;
    }
}

// pre > 1: one thread per (line, chunk), consecutive threads on consecutive lines of the same `after` block: coalesced across the lines
template <bool MAX, int MODE>
__global__ void __launch_bounds__(256) k_cumext_thread(const double* __restrict__ x, double* __restrict__ vals, double* __restrict__ idxs, Lines g,
                                                       u64 chunk_len, u64 nchunks, int reverse, int omit, ExtSum sum) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 nlines = g.pre * g.post;
    if (t >= nlines * nchunks) return;
    const u64 line = t % nlines, ch = t / nlines;  // (chunk-major: the threads of a wave share their chunk)
    const u64 base = (line / g.pre) * g.pre * g.len + line % g.pre;
    const u64 s0 = ch * chunk_len, s1 = (s0 + chunk_len < g.len) ? s0 + chunk_len : g.len;
    const u64 slot = line * nchunks + ch;
    Ext run;
    run.v = 0.0;
    run.pos = 0;
    run.kind = 0;
    if (MODE == 1 && nchunks > 1) {
        run.v = sum.v[slot];
        run.pos = sum.pos[slot];
        run.kind = sum.kind[slot];
    }
    constexpr int TU = 8;  // loads of TU steps in flight before the dependent chain consumes them
    for (u64 s = s0; s < s1; s += TU) {
        double v[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const u64 sc = s + u < s1 ? s + u : s1 - 1;
            const u64 k = reverse ? g.len - 1 - sc : sc;
            v[u] = __builtin_nontemporal_load(x + base + k * g.pre);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            if (s + u < s1) {
                const u64 k = reverse ? g.len - 1 - (s + u) : s + u;
                const u64 at = base + k * g.pre;
                run = ext_combine<MAX>(run, ext_of(v[u], (u32)(k + 1), omit));
                if (MODE == 1) ext_store(run, vals + at, idxs + at);
            }
        }
    }
    if (MODE == 0) {
        sum.v[slot] = run.v;
        sum.pos[slot] = run.pos;
        sum.kind[slot] = run.kind;
    }
}

// summaries[line][chunk] -> what is carried INTO the chunk.  One block per line: thread t owns a run of consecutive chunks, the runs'
// totals are scanned across the block in LDS, then every thread walks its run again from what precedes it.
template <bool MAX>
__global__ void __launch_bounds__(256) k_cumext_carries(ExtSum sum, u64 nlines, u64 nchunks) {
    __shared__ double sv[256];
    __shared__ u32 sp[256];
    __shared__ int sk[256];
    const u64 line = blockIdx.x;
    const int t = threadIdx.x;
    const u64 per = (nchunks + 255) / 256;
    const u64 c0 = (u64)t * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;
    Ext run;
    run.v = 0.0;
    run.pos = 0;
    run.kind = 0;
    for (u64 ch = c0; ch < c1; ++ch) {
        const u64 slot = line * nchunks + ch;
        Ext e;
        e.v = sum.v[slot];
        e.pos = sum.pos[slot];
        e.kind = sum.kind[slot];
        run = ext_combine<MAX>(run, e);
    }
    sv[t] = run.v;
    sp[t] = run.pos;
    sk[t] = run.kind;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of the run totals
        Ext up, me;
        me.v = sv[t];
        me.pos = sp[t];
        me.kind = sk[t];
        if (t >= d) {
            up.v = sv[t - d];
            up.pos = sp[t - d];
            up.kind = sk[t - d];
            me = ext_combine<MAX>(up, me);
        }
        __syncthreads();
        sv[t] = me.v;
        sp[t] = me.pos;
        sk[t] = me.kind;
        __syncthreads();
    }
    run.v = 0.0;
    run.pos = 0;
    run.kind = 0;
    if (t > 0) {
        run.v = sv[t - 1];
        run.pos = sp[t - 1];
        run.kind = sk[t - 1];
    }
    for (u64 ch = c0; ch < c1; ++ch) {
        const u64 slot = line * nchunks + ch;
        Ext e;
        e.v = sum.v[slot];
        e.pos = sum.pos[slot];
        e.kind = sum.kind[slot];
        sum.v[slot] = run.v;
        sum.pos[slot] = run.pos;
        sum.kind[slot] = run.kind;
        run = ext_combine<MAX>(run, e);
    }
}

template <bool MAX>
int launch_cumextreme(Context* c, const double* x, Lines g, int reverse, int omit, double* vals, double* idxs) {
    const u64 nlines = g.pre * g.post;
    if (nlines == 0 || g.len == 0) return RMHIP_OK;
    if (g.len >= 0xffffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "cummin / cummax: a dimension of %llu elements", g.len);
    const bool wave = g.pre == 1;
    // enough independent pieces for the device (two waves per SIMD of threads, or eight waves per CU of wave-pieces), pieces of >= 256
    const u64 want = wave ? (u64)c->num_cus * 8 : (u64)c->num_cus * 512;
    u64 nchunks = 1;
    if (nlines < want && g.len >= 512) {
        nchunks = std::min<u64>((want + nlines - 1) / nlines, g.len / 256);
        nchunks = std::min<u64>(nchunks, 8192);
    }
    u64 chunk_len = (g.len + nchunks - 1) / nchunks;
    if (wave) chunk_len = (chunk_len + 64 * WE - 1) / (64 * WE) * (64 * WE);
    nchunks = (g.len + chunk_len - 1) / chunk_len;
    ExtSum sum{nullptr, nullptr, nullptr};
    std::shared_ptr<Allocation> ws;
    if (nchunks > 1) {
        const u64 slots = nlines * nchunks;
        RMHIP_TRY(c->alloc_device(slots * 2, &ws));  // v: slots doubles; pos + kind: slots * (4 + 4) bytes
        sum.v = ws->ptr;
        sum.pos = (u32*)(ws->ptr + slots);
        sum.kind = (int*)(sum.pos + slots);
    }
    const u64 pieces = nlines * nchunks;
    const unsigned grid = (unsigned)(wave ? (pieces + 3) / 4 : (pieces + 255) / 256);
    if (nchunks > 1) {
        if (wave) hipLaunchKernelGGL((k_cumext_wave<MAX, 0>), dim3(grid), dim3(256), 0, c->stream, x, vals, idxs, g.len, nlines, chunk_len, nchunks, reverse, omit, sum);
        else hipLaunchKernelGGL((k_cumext_thread<MAX, 0>), dim3(grid), dim3(256), 0, c->stream, x, vals, idxs, g, chunk_len, nchunks, reverse, omit, sum);
        hipLaunchKernelGGL(k_cumext_carries<MAX>, dim3((unsigned)nlines), dim3(256), 0, c->stream, sum, nlines, nchunks);
        c->tel.kernel_launches += 2;
    }
    if (wave) hipLaunchKernelGGL((k_cumext_wave<MAX, 1>), dim3(grid), dim3(256), 0, c->stream, x, vals, idxs, g.len, nlines, chunk_len, nchunks, reverse, omit, sum);
    else hipLaunchKernelGGL((k_cumext_thread<MAX, 1>), dim3(grid), dim3(256), 0, c->stream, x, vals, idxs, g, chunk_len, nchunks, reverse, omit, sum);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// diff
// ---------------------------------------------------------------------------------------------------------------------------------
// one thread per output element, output order = memory order of the result: COLMAJ: (before, k, after); else the reference's (k, before, after)
template <bool COLMAJ>
__global__ void __launch_bounds__(256) k_diff(const double* __restrict__ x, double* __restrict__ y, Lines g, u64 total) {
    const u64 o = (u64)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const u64 m = g.len - 1;
    u64 before, k, after;
    if (COLMAJ) {
        before = o % g.pre;
        k = (o / g.pre) % m;
        after = o / (g.pre * m);
    } else {
        k = o % m;
        before = (o / m) % g.pre;
        after = o / (m * g.pre);
    }
    const u64 i0 = before + after * g.pre * g.len + k * g.pre;
    __builtin_nontemporal_store(x[i0 + g.pre] - x[i0], y + o);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// sort / median
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int SORT_C = 2048;       // elements sorted inside one workgroup's LDS
constexpr int SORT_THREADS = 512;  // two compare-exchanges per thread and step

__device__ __forceinline__ u64 sort_key(double x, int descend, int by_abs) {
    if (x != x) return descend ? 0ull : ~0ull;  // NaNs last ascending, first descending (sort.rs:538-551)
    u64 b;
    if (by_abs) {  // |x| first, then x itself (sort.rs:553-574): the sign bit of a nonzero x as the lowest key bit, both zeros alike
        b = ((u64)__double_as_longlong(fabs(x)) << 1) | (x > 0.0 ? 1ull : 0ull);
    } else {
        if (x == 0.0) x = 0.0;  // -0 == +0 under partial_cmp
        const u64 u = (u64)__double_as_longlong(x);
        b = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    return descend ? ~b : b;
}

// workspace[line][i], i < lp: keys + positions of the line's elements, max-key padding beyond len (and beyond the last line)
__global__ void __launch_bounds__(256) k_sort_keys(const double* __restrict__ x, u64* __restrict__ keys, u32* __restrict__ pos, Lines g, u64 lp, u64 ws_total,
                                                   int descend, int by_abs, int before_fastest) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= ws_total) return;
    const u64 nlines = g.pre * g.post;
    u64 line, i;
    if (before_fastest && t < nlines * lp) {  // strided lines: consecutive threads read consecutive lines (coalesced reads, scattered writes)
        const u64 blk = t / (g.pre * lp), r = t % (g.pre * lp);
        line = blk * g.pre + r % g.pre;
        i = r / g.pre;
    } else {
        line = t / lp;
        i = t % lp;
    }
    u64 key = ~0ull;
    u32 p = 0xffffffffu;
    if (line < nlines && i < g.len) {
        key = sort_key(x[(line / g.pre) * g.pre * g.len + line % g.pre + i * g.pre], descend, by_abs);
        p = (u32)i;
    }
    keys[line * lp + i] = key;
    pos[line * lp + i] = p;
}

__device__ __forceinline__ bool pair_after(u64 ka, u32 pa, u64 kb, u32 pb) { return ka > kb || (ka == kb && pa > pb); }

// FULL: every step of the network with k <= min(SORT_C, lp); else the steps j = SORT_C / 2 ... 1 of the given k (> SORT_C)
template <bool FULL>
__global__ void __launch_bounds__(SORT_THREADS) k_bitonic_local(u64* __restrict__ keys, u32* __restrict__ pos, u64 lp, u64 kk) {
    __shared__ u64 sk[SORT_C];
    __shared__ u32 sp[SORT_C];
    const u64 g0 = (u64)blockIdx.x * SORT_C;
    for (int i = threadIdx.x; i < SORT_C; i += SORT_THREADS) {
        sk[i] = keys[g0 + i];
        sp[i] = pos[g0 + i];
    }
    __syncthreads();
    const u64 kmax = FULL ? (lp < (u64)SORT_C ? lp : (u64)SORT_C) : kk;
    for (u64 k = FULL ? 2 : kk; k <= kmax; k <<= 1) {
        for (u32 j = (u32)((FULL ? k : (u64)SORT_C) >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < SORT_C / 2; t += SORT_THREADS) {
                const u32 i = ((t / j) * 2 * j) + (t % j), l = i + j;
                const bool asc = (((g0 + i) & (lp - 1)) & k) == 0;  // the position INSIDE the line decides the direction
                const u64 ki = sk[i], kl = sk[l];
                const u32 pi = sp[i], pl = sp[l];
                if (pair_after(ki, pi, kl, pl) == asc) {
                    sk[i] = kl;
                    sk[l] = ki;
                    sp[i] = pl;
                    sp[l] = pi;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < SORT_C; i += SORT_THREADS) {
        keys[g0 + i] = sk[i];
        pos[g0 + i] = sp[i];
    }
}

__global__ void __launch_bounds__(256) k_bitonic_global(u64* __restrict__ keys, u32* __restrict__ pos, u64 lp, u64 k, u64 j, u64 half_total) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= half_total) return;
    const u64 i = (t / j) * 2 * j + (t % j), l = i + j;
    const bool asc = ((i & (lp - 1)) & k) == 0;
    const u64 ki = keys[i], kl = keys[l];
    const u32 pi = pos[i], pl = pos[l];
    if (pair_after(ki, pi, kl, pl) == asc) {
        keys[i] = kl;
        keys[l] = ki;
        pos[i] = pl;
        pos[l] = pi;
    }
}

// sorted[line][r] = x[line][pos[r]], indices = pos + 1 (output in the operand's layout)
__global__ void __launch_bounds__(256) k_sort_emit(const double* __restrict__ x, const u32* __restrict__ pos, Lines g, u64 lp, double* __restrict__ sorted,
                                                   double* __restrict__ indices, u64 total) {
    const u64 o = (u64)blockIdx.x * 256 + threadIdx.x;  // output element in memory order
    if (o >= total) return;
    const u64 before = o % g.pre, r = (o / g.pre) % g.len, after = o / (g.pre * g.len);
    const u64 line = after * g.pre + before;
    const u32 p = pos[line * lp + r];
    const u64 base = after * g.pre * g.len + before;
    __builtin_nontemporal_store(x[base + (u64)p * g.pre], sorted + o);
    __builtin_nontemporal_store((double)(p + 1), indices + o);
}

// median of every sorted line: NaN keys sort last, so one look at the last element tells whether the line held a NaN
__global__ void __launch_bounds__(256) k_median_emit(const double* __restrict__ x, const u64* __restrict__ keys, const u32* __restrict__ pos, Lines g, u64 lp,
                                                     double* __restrict__ out) {
    const u64 line = (u64)blockIdx.x * 256 + threadIdx.x;
    if (line >= g.pre * g.post) return;
    const u64 base = (line / g.pre) * g.pre * g.len + line % g.pre;
    double m = quiet_nan();
    if (g.len > 0 && keys[line * lp + g.len - 1] != ~0ull) {
        const double hi = x[base + (u64)pos[line * lp + g.len / 2] * g.pre];
        if (g.len & 1) m = hi;
        else m = 0.5 * (x[base + (u64)pos[line * lp + g.len / 2 - 1] * g.pre] + hi);  // median.rs:733-737
    }
    out[line] = m;
}

struct SortSpace {
    std::shared_ptr<Allocation> mem;
    u64* keys = nullptr;
    u32* pos = nullptr;
    u64 lp = 0, total = 0;
};

// workspace for `nlines` lines of `len` pairs (padded to a power of two each, the total to whole workgroup tiles)
int sort_alloc(Context* c, u64 nlines, u64 len, SortSpace* ws) {
    if (len >= 0x7fffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "sort: a dimension of %llu elements", len);
    u64 lp = 2;
    while (lp < len) lp <<= 1;
    u64 total = nlines * lp;
    total = (total + SORT_C - 1) / SORT_C * SORT_C;
    if (total / lp > 0x7fffffffull || total > (1ull << 40)) return fail(RMHIP_ERR_UNSUPPORTED, "sort: workspace of %llu pairs", total);
    RMHIP_TRY(c->alloc_device(total + (total + 1) / 2, &ws->mem));  // 8 + 4 bytes per pair
    ws->keys = (u64*)ws->mem->ptr;
    ws->pos = (u32*)(ws->mem->ptr + total);
    ws->lp = lp;
    ws->total = total;
    return RMHIP_OK;
}

// the bitonic network over the (key, position) pairs already in the workspace
int sort_pairs(Context* c, SortSpace* ws) {
    const u64 lp = ws->lp, total = ws->total;
    hipLaunchKernelGGL(k_bitonic_local<true>, dim3((unsigned)(total / SORT_C)), dim3(SORT_THREADS), 0, c->stream, ws->keys, ws->pos, lp, (u64)0);
    c->tel.kernel_launches++;
    for (u64 k = 2 * (u64)SORT_C; k <= lp; k <<= 1) {
        for (u64 j = k >> 1; j >= (u64)SORT_C; j >>= 1) {
            hipLaunchKernelGGL(k_bitonic_global, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, c->stream, ws->keys, ws->pos, lp, k, j, total / 2);
            c->tel.kernel_launches++;
        }
        hipLaunchKernelGGL(k_bitonic_local<false>, dim3((unsigned)(total / SORT_C)), dim3(SORT_THREADS), 0, c->stream, ws->keys, ws->pos, lp, k);
        c->tel.kernel_launches++;
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// keys + positions of every line, sorted
int sort_lines(Context* c, const double* x, Lines g, int descend, int by_abs, SortSpace* ws) {
    RMHIP_TRY(sort_alloc(c, g.pre * g.post, g.len, ws));
    hipLaunchKernelGGL(k_sort_keys, dim3((unsigned)((ws->total + 255) / 256)), dim3(256), 0, c->stream, x, ws->keys, ws->pos, g, ws->lp, ws->total, descend, by_abs,
                       g.pre > 1 ? 1 : 0);
    c->tel.kernel_launches++;
    return sort_pairs(c, ws);
}

// sortrows (runmat-accelerate/src/sortrows_host.rs:11-140): one stable pass per key column, last key first.  `perm[r]`: the row now at rank r.
__global__ void __launch_bounds__(256) k_rows_keys(const double* __restrict__ x, u64 rows, u64 col, const u32* __restrict__ perm, int descend, int by_abs, u64* __restrict__ keys,
                                                   u32* __restrict__ pos, u64 total) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= total) return;
    u64 key = ~0ull;
    u32 p = 0xffffffffu;
    if (r < rows) key = sort_key(x[(perm ? perm[r] : (u32)r) + col * rows], descend, by_abs), p = (u32)r;
    keys[r] = key, pos[r] = p;
}

__global__ void __launch_bounds__(256) k_iota1(double* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)i + 1.0;
}

__global__ void __launch_bounds__(256) k_rows_compose(const u32* __restrict__ perm, const u32* __restrict__ pos, u64 rows, u32* __restrict__ out) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) out[r] = perm ? perm[pos[r]] : pos[r];
}

__global__ void __launch_bounds__(256) k_rows_emit(const double* __restrict__ x, const u32* __restrict__ perm, u64 rows, u64 total, double* __restrict__ sorted,
                                                   double* __restrict__ indices) {
    const u64 o = (u64)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const u64 r = o % rows, col = o / rows;
    const u32 src = perm ? perm[r] : (u32)r;
    sorted[o] = x[src + col * rows];
    if (col == 0) indices[r] = (double)src + 1.0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// find: ordered stream compaction.  A wave owns FIND_ROWS rows of 64 consecutive scan positions; pass 1 counts its nonzeros, one
// workgroup turns the counts into offsets, pass 2 places every nonzero at offset + (number of nonzeros before it in the wave).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int FIND_ROWS = 16;

__global__ void __launch_bounds__(256) k_find_count(const double* __restrict__ x, u64 n, int last, u32* __restrict__ counts, u64 nchunks) {
    const u64 w = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= nchunks) return;
    u32 cnt = 0;
#pragma unroll 4
    for (int r = 0; r < FIND_ROWS; ++r) {
        const u64 p = (w * FIND_ROWS + r) * 64 + lane;
        const bool nz = p < n && x[last ? n - 1 - p : p] != 0.0;
        cnt += (u32)__popcll(__ballot(nz));
    }
    if (lane == 0) counts[w] = cnt;
}

// exclusive scan of the chunk counts by ONE workgroup: thread t owns a run of consecutive chunks
__global__ void __launch_bounds__(1024) k_find_scan(const u32* __restrict__ counts, u64 nchunks, u64* __restrict__ offsets, u64* __restrict__ total) {
    __shared__ u64 run[1024];
    const int t = threadIdx.x;
    const u64 per = (nchunks + 1023) / 1024;
    const u64 c0 = (u64)t * per, c1 = (c0 + per < nchunks) ? c0 + per : nchunks;
    u64 sum = 0;
    for (u64 ch = c0; ch < c1; ++ch) sum += counts[ch];
    run[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const u64 add = t >= d ? run[t - d] : 0;
        __syncthreads();
        run[t] += add;
        __syncthreads();
    }
    u64 off = t > 0 ? run[t - 1] : 0;
    for (u64 ch = c0; ch < c1; ++ch) {
        offsets[ch] = off;
        off += counts[ch];
    }
    if (t == 1023) *total = run[1023];
}

__global__ void __launch_bounds__(256) k_find_emit(const double* __restrict__ x, u64 n, int last, const u64* __restrict__ offsets, u64 nchunks, u64 count,
                                                   u64 row_extent, double* __restrict__ linear, double* __restrict__ rows, double* __restrict__ cols,
                                                   double* __restrict__ values) {
    const u64 w = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= nchunks) return;
    u64 off = offsets[w];
    for (int r = 0; r < FIND_ROWS && off < count; ++r) {
        const u64 p = (w * FIND_ROWS + r) * 64 + lane;
        const u64 idx = last ? n - 1 - p : p;
        const double v = p < n ? x[idx] : 0.0;
        const bool nz = p < n && v != 0.0;
        const u64 mask = __ballot(nz);
        const u64 at = off + (u64)__popcll(mask & ((1ull << lane) - 1ull));
        if (nz && at < count) {
            linear[at] = (double)(idx + 1);
            rows[at] = (double)(idx % row_extent + 1);
            cols[at] = (double)(idx / row_extent + 1);
            values[at] = v;
        }
        off += (u64)__popcll(mask);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// median of LONG lines by radix selection: the two middle ranks are located digit by digit (eight passes over the line's keys, one
// byte each, most significant first) instead of sorting the line - ~8 reads of the data against the 100+ passes of the network.
// Per line and target rank the state is (the key's decided high bytes, the rank inside that group); a pass histograms the next byte
// of the elements that match the prefix, a one-wave kernel picks the bucket.  The key decodes to the value except for a zero, whose
// sign is that of the rank-th zero in index order (stable order): found by an ordered count when - and only when - it matters.
// ---------------------------------------------------------------------------------------------------------------------------------
struct SelState {
    u64 prefix[2];  // lower middle, upper middle
    u64 rank[2];
    u32 nan;        // the line holds a NaN
    u32 pad;
};

__global__ void __launch_bounds__(256) k_sel_init(SelState* __restrict__ st, u32* __restrict__ hist, u64 len, u64 nlines) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t < nlines) {
        st[t].prefix[0] = st[t].prefix[1] = 0;
        st[t].rank[0] = (len - 1) / 2;
        st[t].rank[1] = len / 2;
        st[t].nan = 0;
    }
    if (t < nlines * 512) hist[t] = 0;
}

// one (line, piece) per workgroup: byte `shift / 8` of every key whose higher bytes equal the target's prefix
__global__ void __launch_bounds__(256) k_sel_hist(const double* __restrict__ x, Lines g, SelState* __restrict__ st, u32* __restrict__ hist, int shift, u64 piece) {
    __shared__ u32 h[2][256];
    const u64 line = blockIdx.y;
    const int t = threadIdx.x;
    h[0][t] = 0;
    h[1][t] = 0;
    __syncthreads();
    const SelState s = st[line];
    const bool same = s.prefix[0] == s.prefix[1];
    const u64 base = (line / g.pre) * g.pre * g.len + line % g.pre;
    const u64 k0 = (u64)blockIdx.x * piece, k1 = (k0 + piece < g.len) ? k0 + piece : g.len;
    const int hs = shift + 8;  // bits above the byte under the histogram
    u32 saw_nan = 0;
    for (u64 k = k0 + t; k < k1; k += 256) {
        const u64 key = sort_key(__builtin_nontemporal_load(x + base + k * g.pre), 0, 0);
        if (shift == 56 && key == ~0ull) saw_nan = 1;
        const u32 bin = (u32)(key >> shift) & 255u;
        const bool m0 = hs >= 64 || (key >> hs) == (s.prefix[0] >> hs);
        const bool m1 = !same && (hs >= 64 || (key >> hs) == (s.prefix[1] >> hs));
        // wave-aggregated increments: data of one magnitude shares its leading bytes, and same-address LDS atomics serialise
        u64 todo = __ballot(m0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 lb = __shfl(bin, leader);
            const u64 grp = __ballot(m0 && bin == lb) & todo;
            if ((t & 63) == leader) atomicAdd(&h[0][lb], (u32)__popcll(grp));
            todo &= ~grp;
        }
        if (!same) {
            todo = __ballot(m1);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const u32 lb = __shfl(bin, leader);
                const u64 grp = __ballot(m1 && bin == lb) & todo;
                if ((t & 63) == leader) atomicAdd(&h[1][lb], (u32)__popcll(grp));
                todo &= ~grp;
            }
        }
    }
    if (saw_nan) st[line].nan = 1;
    __syncthreads();
    if (h[0][t]) atomicAdd(hist + line * 512 + t, h[0][t]);
    if (!same && h[1][t]) atomicAdd(hist + line * 512 + 256 + t, h[1][t]);
}

// one wave per line: the bucket that holds each target rank; the histograms are cleared for the next pass
__global__ void __launch_bounds__(64) k_sel_pick(SelState* __restrict__ st, u32* __restrict__ hist, int shift) {
    const u64 line = blockIdx.x;
    const int lane = threadIdx.x;
    SelState s = st[line];
    const bool same = s.prefix[0] == s.prefix[1];
    for (int tg = 0; tg < 2; ++tg) {
        const u32* h = hist + line * 512 + ((tg == 1 && !same) ? 256 : 0);
        // lane l owns buckets 4 l .. 4 l + 3
        u64 c[4], mine = 0;
        for (int q = 0; q < 4; ++q) {
            c[q] = h[4 * lane + q];
            mine += c[q];
        }
        u64 incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const u64 up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        u64 before = incl - mine;
        const u64 r = s.rank[tg];
        int found = -1;
        u64 nr = 0;
        for (int q = 0; q < 4; ++q) {
            if (found < 0 && r >= before && r < before + c[q]) {
                found = 4 * lane + q;
                nr = r - before;
            }
            before += c[q];
        }
        const u64 who = __ballot(found >= 0);
        const int src = who ? __ffsll((long long)who) - 1 : 0;
        const int bucket = __shfl(found, src);
        const u64 newrank = __shfl(nr, src);
        if (lane == 0) {
            st[line].prefix[tg] = s.prefix[tg] | ((u64)(u32)bucket << shift);
            st[line].rank[tg] = newrank;
        }
    }
    __syncthreads();
    for (int q = lane; q < 512; q += 64) hist[line * 512 + q] = 0;
}

// the rank-th (0-based) zero of a line in index order: its sign decides a median that lands on a zero
__global__ void __launch_bounds__(256) k_kth_zero(const double* __restrict__ x, Lines g, u64 line, u64 kth, double* __restrict__ out) {
    __shared__ u64 cnt[256];
    const int t = threadIdx.x;
    const u64 base = (line / g.pre) * g.pre * g.len + line % g.pre;
    const u64 per = (g.len + 255) / 256, k0 = (u64)t * per, k1 = (k0 + per < g.len) ? k0 + per : g.len;
    u64 mine = 0;
    for (u64 k = k0; k < k1; ++k) mine += x[base + k * g.pre] == 0.0 ? 1 : 0;
    cnt[t] = mine;
    __syncthreads();
    u64 before = 0;
    for (int i = 0; i < t; ++i) before += cnt[i];
    if (kth >= before && kth < before + mine) {
        u64 seen = before;
        for (u64 k = k0; k < k1; ++k) {
            const double v = x[base + k * g.pre];
            if (v == 0.0) {
                if (seen == kth) {
                    *out = v;
                    break;
                }
                ++seen;
            }
        }
    }
}

__device__ __host__ inline double key_to_value(u64 key) {  // inverse of sort_key (ascending, by value); a zero decodes to +0
    const u64 u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    double v;
    memcpy(&v, &u, sizeof v);
    return v;
}

// medians of g's lines into out[line]; lines are long and few (the caller's choice)
int median_select(Context* c, const double* x, Lines g, double* out) {
    const u64 nlines = g.pre * g.post;
    std::shared_ptr<Allocation> ws;
    const u64 st_doubles = nlines * (sizeof(SelState) / 8), hist_doubles = nlines * 256;  // 512 u32 per line
    RMHIP_TRY(c->alloc_device(st_doubles + hist_doubles + 2, &ws));
    SelState* st = (SelState*)ws->ptr;
    u32* hist = (u32*)(ws->ptr + st_doubles);
    double* zero_out = ws->ptr + st_doubles + hist_doubles;
    hipLaunchKernelGGL(k_sel_init, dim3((unsigned)((nlines * 512 + 255) / 256)), dim3(256), 0, c->stream, st, hist, g.len, nlines);
    // pieces of >= 16 Ki elements, enough workgroups for the device
    u64 pieces = std::max<u64>(1, std::min<u64>((u64)c->num_cus * 8 / std::max<u64>(nlines, 1), g.len / 16384));
    const u64 piece = (g.len + pieces - 1) / pieces;
    pieces = (g.len + piece - 1) / piece;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)pieces, (unsigned)nlines), dim3(256), 0, c->stream, x, g, st, hist, shift, piece);
        hipLaunchKernelGGL(k_sel_pick, dim3((unsigned)nlines), dim3(64), 0, c->stream, st, hist, shift);
        c->tel.kernel_launches += 2;
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    std::vector<SelState> host(nlines);
    RMHIP_HIP_CHECK(hipMemcpyAsync(host.data(), st, nlines * sizeof(SelState), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<double> med(nlines);
    for (u64 l = 0; l < nlines; ++l) {
        if (host[l].nan) {
            med[l] = std::numeric_limits<double>::quiet_NaN();
            continue;
        }
        double v[2];
        for (int tg = 0; tg < 2; ++tg) {
            v[tg] = key_to_value(host[l].prefix[tg]);
            if (host[l].prefix[tg] == 0x8000000000000000ull && (tg == 1 || (g.len & 1) == 0)) {  // a zero: which one?
                hipLaunchKernelGGL(k_kth_zero, dim3(1), dim3(256), 0, c->stream, x, g, l, host[l].rank[tg], zero_out);
                RMHIP_HIP_CHECK(hipMemcpyAsync(&v[tg], zero_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
                RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
            }
        }
        med[l] = (g.len & 1) ? v[1] : 0.5 * (v[0] + v[1]);  // median.rs:733-737
    }
    RMHIP_HIP_CHECK(hipMemcpyAsync(out, med.data(), nlines * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // `med` leaves scope
    return RMHIP_OK;
}

int lines_of(const std::vector<size_t>& shape, int dim, const char* what, Lines* g) {
    if (dim < 0 || (size_t)dim >= shape.size()) return fail(RMHIP_ERR_UNSUPPORTED, "%s: dim %d out of range for rank %zu", what, dim, shape.size());
    g->pre = g->post = 1;
    for (int d = 0; d < dim; ++d) g->pre *= shape[d];
    for (size_t d = dim + 1; d < shape.size(); ++d) g->post *= shape[d];
    g->len = shape[dim];
    return RMHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// unique / ismember (elements): the CPU's hash maps keyed by `canonicalize_f64` (unique.rs:1347-1355: every NaN one key, both zeros one
// key) become the sorted (key, position) pairs of the whole tensor - equal keys are neighbours, ordered by position, so a group's head
// is its FIRST occurrence and its tail the LAST.  Group ids are a scan of the head flags.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_CHUNK = 1024;  // flags per workgroup (256 threads x 4)

__global__ void __launch_bounds__(256) k_group_heads(const u64* __restrict__ keys, u64 n, u32* __restrict__ flags, u32* __restrict__ counts) {
    __shared__ u32 part[4];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
    u32 cnt = 0;
    for (int e = 0; e < 4; ++e) {
        const u64 i = base + e;
        u32 f = 0;
        if (i < n) f = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
        if (i < n) flags[i] = f;
        cnt += f;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the chunk counts by one workgroup; total[0] = number of groups
__global__ void __launch_bounds__(1024) k_chunk_offsets(const u32* __restrict__ counts, u64 nchunks, u32* __restrict__ offsets, u32* __restrict__ total) {
    __shared__ u32 warp_sums[16];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nchunks; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u32 v = i < nchunks ? counts[i] : 0;
        u32 incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 up = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += up;
        }
        if ((threadIdx.x & 63) == 63) warp_sums[threadIdx.x >> 6] = incl;
        __syncthreads();
        u32 before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += warp_sums[w];
        if (i < nchunks) offsets[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

// gid[i] = (heads at or before i) - 1; the head of group g records its sorted index, the tail its element position
__global__ void __launch_bounds__(256) k_group_ids(const u32* __restrict__ flags, const u32* __restrict__ offsets, const u32* __restrict__ pos, u64 n,
                                                   u32* __restrict__ gid, u32* __restrict__ first_pos, u32* __restrict__ last_pos) {
    __shared__ u32 wave_tot[4];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
    u32 f[4], mine = 0;
    for (int e = 0; e < 4; ++e) {
        f[e] = base + e < n ? flags[base + e] : 0;
        mine += f[e];
    }
    u32 incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 up = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 run = offsets[blockIdx.x] + incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wave_tot[w];
    for (int e = 0; e < 4; ++e) {
        const u64 i = base + e;
        if (i >= n) break;
        run += f[e];
        const u32 g = run - 1;
        gid[i] = g;
        if (f[e]) first_pos[g] = pos[i];
        if (i + 1 == n || flags[i + 1]) last_pos[g] = pos[i];
    }
}

__global__ void __launch_bounds__(256) k_u32_to_double(const u32* __restrict__ src, u64 n, double* __restrict__ dst) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

// rank_of[group] = r for the r-th group in output order (`order[r]` = group, or null: sorted order, rank = group)
__global__ void __launch_bounds__(256) k_unique_outputs(const double* __restrict__ x, const u32* __restrict__ order, const u32* __restrict__ first_pos,
                                                        const u32* __restrict__ last_pos, u64 groups, int take_last, double* __restrict__ values, double* __restrict__ ia,
                                                        u32* __restrict__ rank_of) {
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= groups) return;
    const u32 g = order ? order[r] : (u32)r;
    values[r] = x[first_pos[g]];  // the entry keeps the value of its first occurrence (unique.rs:505-511)
    ia[r] = (double)(take_last ? last_pos[g] : first_pos[g]) + 1.0;
    rank_of[g] = (u32)r;
}

__global__ void __launch_bounds__(256) k_unique_inverse(const u32* __restrict__ pos, const u32* __restrict__ gid, const u32* __restrict__ rank_of, u64 n, double* __restrict__ ic) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ic[pos[i]] = (double)rank_of[gid[i]] + 1.0;
}

// mask / loc of every element of a against the sorted pairs of b: the first pair with the element's key holds b's lowest position
__global__ void __launch_bounds__(256) k_ismember(const double* __restrict__ a, u64 na, const u64* __restrict__ keys, const u32* __restrict__ pos, u64 nb,
                                                  unsigned char* __restrict__ mask, double* __restrict__ loc) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= na) return;
    const u64 key = sort_key(a[i], 0, 0);
    u64 lo = 0, hi = nb;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    const bool hit = lo < nb && keys[lo] == key;
    mask[i] = hit ? 1 : 0;
    loc[i] = hit ? (double)pos[lo] + 1.0 : 0.0;
}

}  // namespace
}  // namespace rmhip

int rmhip_cumextreme(rmhip_ctx* ctx, int is_max, rmhip_buf a, int dim, int reverse, int nan_mode, rmhip_buf* values, rmhip_buf* indices) {
    CTX_OR_FAIL(ctx);
    if (!values || !indices) return fail(RMHIP_ERR_INVALID, "cumextreme: null output");
    *values = *indices = 0;
    Buffer ab, vb, ib;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = matrix_shape(ab.shape);
    Lines g;
    RMHIP_TRY(lines_of(shape, dim, "cummin / cummax", &g));
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), values, &vb));
    int rc = c->new_buffer(shape.data(), shape.size(), indices, &ib);
    if (rc == RMHIP_OK && ab.numel > 0)
        rc = is_max ? launch_cumextreme<true>(c, ab.data(), g, reverse ? 1 : 0, nan_mode ? 1 : 0, vb.data(), ib.data())
                    : launch_cumextreme<false>(c, ab.data(), g, reverse ? 1 : 0, nan_mode ? 1 : 0, vb.data(), ib.data());
    if (rc) {
        rmhip_free(ctx, *values);
        if (*indices) rmhip_free(ctx, *indices);
    }
    return rc;
}

int rmhip_diff_dim(rmhip_ctx* ctx, rmhip_buf a, size_t order, int dim, int column_major, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "diff_dim: dim must be >= 0");
    Buffer cur;
    RMHIP_TRY(c->get(a, &cur));
    std::vector<size_t> shape = cur.shape;  // diff_tensor_once: the shape is extended with ones up to the dimension (diff.rs:479-481)
    while (shape.size() <= (size_t)dim) shape.push_back(1);
    if (order == 0) {  // simple_provider.rs:6480-6482: the operand itself
        Buffer ob;
        RMHIP_TRY(c->new_buffer(cur.shape.data(), cur.shape.size(), out, &ob));
        if (cur.numel) RMHIP_HIP_CHECK(hipMemcpyAsync(ob.data(), cur.data(), cur.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return RMHIP_OK;
    }
    rmhip_buf cur_id = 0;
    for (size_t pass = 0; pass < order; ++pass) {
        Lines g;
        RMHIP_TRY(lines_of(shape, dim, "diff_dim", &g));
        std::vector<size_t> oshape = shape;
        oshape[dim] = g.len > 0 ? g.len - 1 : 0;
        rmhip_buf next = 0;
        Buffer nb;
        const int rc = c->new_buffer(oshape.data(), oshape.size(), &next, &nb);
        if (rc == RMHIP_OK && nb.numel > 0) {
            if (column_major || g.pre == 1) hipLaunchKernelGGL(k_diff<true>, dim3((unsigned)((nb.numel + 255) / 256)), dim3(256), 0, c->stream, cur.data(), nb.data(), g, (u64)nb.numel);
            else hipLaunchKernelGGL(k_diff<false>, dim3((unsigned)((nb.numel + 255) / 256)), dim3(256), 0, c->stream, cur.data(), nb.data(), g, (u64)nb.numel);
            c->tel.kernel_launches++;
        }
        if (cur_id) rmhip_free(ctx, cur_id);
        if (rc) return rc;
        cur_id = next;
        cur = nb;
        shape = oshape;
        if (nb.numel == 0) break;  // diff_tensor_host stops at the first empty result (diff.rs:444-446)
    }
    RMHIP_HIP_CHECK(hipGetLastError());
    *out = cur_id;
    return RMHIP_OK;
}

int rmhip_sort_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, int descend, int by_abs, rmhip_buf* sorted, rmhip_buf* indices) {
    CTX_OR_FAIL(ctx);
    if (!sorted || !indices) return fail(RMHIP_ERR_INVALID, "sort_dim: null output");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "sort_dim: dim must be >= 0");
    *sorted = *indices = 0;
    Buffer ab, sb, ib;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = matrix_shape(ab.shape);
    Lines g{ab.numel, 1, 1};  // a dimension beyond the rank: lines of one element (sort.rs:428-437: values unchanged, indices all one)
    if ((size_t)dim < shape.size()) RMHIP_TRY(lines_of(shape, dim, "sort_dim", &g));
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), sorted, &sb));
    int rc = c->new_buffer(shape.data(), shape.size(), indices, &ib);
    if (rc == RMHIP_OK && ab.numel > 0 && g.len <= 1) {  // sort.rs:428-437: nothing to order - the values themselves, every index one
        RMHIP_HIP_CHECK(hipMemcpyAsync(sb.data(), ab.data(), ab.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        rc = launch_fill(c, ib.data(), ib.numel, 1.0);
    } else if (rc == RMHIP_OK && ab.numel > 0) {
        SortSpace ws;
        rc = sort_lines(c, ab.data(), g, descend ? 1 : 0, by_abs ? 1 : 0, &ws);
        if (rc == RMHIP_OK) {
            hipLaunchKernelGGL(k_sort_emit, dim3((unsigned)((ab.numel + 255) / 256)), dim3(256), 0, c->stream, ab.data(), ws.pos, g, ws.lp, sb.data(), ib.data(), (u64)ab.numel);
            c->tel.kernel_launches++;
            if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "sort_dim: launch failed");
        }
    }
    if (rc) {
        rmhip_free(ctx, *sorted);
        if (*indices) rmhip_free(ctx, *indices);
    }
    return rc;
}

int rmhip_reduce_median(rmhip_ctx* ctx, rmhip_buf a, int dim, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, ob;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = matrix_shape(ab.shape);
    Lines g{1, ab.numel, 1};  // dim < 0: every element (reduce_median, simple_provider.rs:7167-7193) -> [1, 1]
    std::vector<size_t> oshape{1, 1};
    if (dim >= 0) {
        RMHIP_TRY(lines_of(shape, dim, "reduce_median_dim", &g));
        oshape = shape;
        oshape[dim] = 1;
    }
    RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    int rc = RMHIP_OK;
    if (g.len == 0) {  // empty slices: NaN (median.rs:668-672)
        rc = launch_fill(c, ob.data(), ob.numel, std::numeric_limits<double>::quiet_NaN());
    } else if (g.len == 1) {
        RMHIP_HIP_CHECK(hipMemcpyAsync(ob.data(), ab.data(), ab.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else if (g.len >= (1u << 17) && g.pre * g.post <= 64 && !std::getenv("RMHIP_MEDIAN_SORT")) {
        rc = median_select(c, ab.data(), g, ob.data());  // long lines, few of them: selection instead of a full sort
    } else {
        SortSpace ws;
        rc = sort_lines(c, ab.data(), g, 0, 0, &ws);
        if (rc == RMHIP_OK) {
            hipLaunchKernelGGL(k_median_emit, dim3((unsigned)((ob.numel + 255) / 256)), dim3(256), 0, c->stream, ab.data(), ws.keys, ws.pos, g, ws.lp, ob.data());
            c->tel.kernel_launches++;
            if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "reduce_median: launch failed");
        }
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_find(rmhip_ctx* ctx, rmhip_buf a, long long limit_or_neg, int last, rmhip_buf* linear, rmhip_buf* rows, rmhip_buf* cols, rmhip_buf* values) {
    CTX_OR_FAIL(ctx);
    if (!linear || !rows || !cols || !values) return fail(RMHIP_ERR_INVALID, "find: null output");
    *linear = *rows = *cols = *values = 0;
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const u64 n = ab.numel;
    // simple_provider.rs:7513-7517: first -> limit or everything, last -> limit or ONE
    u64 cap = limit_or_neg >= 0 ? (u64)limit_or_neg : (last ? 1ull : n);
    if (cap > n) cap = n;
    u64 count = 0;
    std::shared_ptr<Allocation> ws;
    u64* offsets = nullptr;
    const u64 nchunks = (n + 64 * FIND_ROWS - 1) / (64 * FIND_ROWS);
    if (cap > 0) {
        RMHIP_TRY(c->alloc_device(nchunks + (nchunks + 1) / 2 + 2, &ws));  // offsets (u64), total (u64), counts (u32)
        offsets = (u64*)ws->ptr;
        u64* total = offsets + nchunks;
        u32* counts = (u32*)(total + 1);
        hipLaunchKernelGGL(k_find_count, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, c->stream, ab.data(), n, last ? 1 : 0, counts, nchunks);
        hipLaunchKernelGGL(k_find_scan, dim3(1), dim3(1024), 0, c->stream, counts, nchunks, offsets, total);
        c->tel.kernel_launches += 2;
        RMHIP_HIP_CHECK(hipGetLastError());
        unsigned long long found = 0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&found, total, sizeof(found), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the outputs' size is the answer
        count = found < cap ? found : cap;
    }
    const size_t oshape[2] = {(size_t)count, 1};
    Buffer lb, rb, cb, vb;
    rmhip_buf* outs[4] = {linear, rows, cols, values};
    Buffer* bufs[4] = {&lb, &rb, &cb, &vb};
    int rc = RMHIP_OK;
    for (int i = 0; i < 4 && rc == RMHIP_OK; ++i) rc = c->new_buffer(oshape, 2, outs[i], bufs[i]);
    if (rc == RMHIP_OK && count > 0) {
        const u64 row_extent = ab.shape.empty() || ab.shape[0] == 0 ? 1 : ab.shape[0];
        hipLaunchKernelGGL(k_find_emit, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, c->stream, ab.data(), n, last ? 1 : 0, offsets, nchunks, count, row_extent,
                           lb.data(), rb.data(), cb.data(), vb.data());
        c->tel.kernel_launches++;
        if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "find: launch failed");
    }
    if (rc)
        for (int i = 0; i < 4; ++i)
            if (*outs[i]) rmhip_free(ctx, *outs[i]);
    return rc;
}

namespace rmhip {
namespace {
struct UniqueDev {
    std::shared_ptr<Allocation> outs;  // values | ia | ic
    double *dv = nullptr, *dia = nullptr, *dic = nullptr;
    u64 groups = 0;
};

// the distinct values of x[0 .. n) (n > 0) with their first / last positions and the inverse map, left on the device
int unique_device(Context* c, const double* x, u64 n, int stable, int last_occurrence, UniqueDev* r) {
    SortSpace ws;
    RMHIP_TRY(sort_lines(c, x, Lines{1, n, 1}, 0, 0, &ws));
    const u64 nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    // u32 work arrays: flags n | gid n | first n | last n | rank n | counts nchunks | offsets nchunks | total 2
    std::shared_ptr<Allocation> wk;
    RMHIP_TRY(c->alloc_device((5 * n + 2 * nchunks + 2 + 1) / 2 + 1, &wk));
    u32* flags = (u32*)wk->ptr;
    u32 *gid = flags + n, *first_pos = gid + n, *last_pos = first_pos + n, *rank_of = last_pos + n, *counts = rank_of + n, *offsets = counts + nchunks, *total = offsets + nchunks;
    hipLaunchKernelGGL(k_group_heads, dim3((unsigned)nchunks), dim3(256), 0, c->stream, ws.keys, n, flags, counts);
    hipLaunchKernelGGL(k_chunk_offsets, dim3(1), dim3(1024), 0, c->stream, counts, nchunks, offsets, total);
    hipLaunchKernelGGL(k_group_ids, dim3((unsigned)nchunks), dim3(256), 0, c->stream, flags, offsets, ws.pos, n, gid, first_pos, last_pos);
    c->tel.kernel_launches += 3;
    RMHIP_HIP_CHECK(hipGetLastError());
    u32 groups32 = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&groups32, total, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the outputs' size is part of the answer
    const u64 groups = groups32;
    RMHIP_TRY(c->alloc_device(2 * groups + n, &r->outs));
    r->dv = r->outs->ptr, r->dia = r->dv + groups, r->dic = r->dia + groups, r->groups = groups;
    const u32* order = nullptr;
    SortSpace ws2;
    std::shared_ptr<Allocation> fp;
    if (stable && groups > 1) {  // groups in order of their first occurrence (unique.rs:516-519: `order` stays the insertion order)
        RMHIP_TRY(c->alloc_device(groups, &fp));
        hipLaunchKernelGGL(k_u32_to_double, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, c->stream, first_pos, groups, fp->ptr);
        c->tel.kernel_launches++;
        RMHIP_TRY(sort_lines(c, fp->ptr, Lines{1, groups, 1}, 0, 0, &ws2));
        order = ws2.pos;
    }
    hipLaunchKernelGGL(k_unique_outputs, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, c->stream, x, order, first_pos, last_pos, groups, last_occurrence ? 1 : 0, r->dv,
                       r->dia, rank_of);
    hipLaunchKernelGGL(k_unique_inverse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ws.pos, gid, rank_of, n, r->dic);
    c->tel.kernel_launches += 2;
    RMHIP_HIP_CHECK(hipGetLastError());
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // (the work arrays above are released on return)
    return RMHIP_OK;
}
}  // namespace
}  // namespace rmhip

int rmhip_unique(rmhip_ctx* ctx, rmhip_buf a, int stable, int last_occurrence, size_t* count, double* values_host, double* ia_host, double* ic_host) {
    CTX_OR_FAIL(ctx);
    if (!count) return fail(RMHIP_ERR_INVALID, "unique: null count");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const u64 n = ab.numel;
    *count = 0;
    if (n == 0) return RMHIP_OK;
    if (!values_host || !ia_host || !ic_host) return fail(RMHIP_ERR_INVALID, "unique: null output");
    UniqueDev r;
    RMHIP_TRY(unique_device(c, ab.data(), n, stable, last_occurrence, &r));
    RMHIP_HIP_CHECK(hipMemcpyAsync(values_host, r.dv, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(ia_host, r.dia, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(ic_host, r.dic, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->tel.download_bytes += (2 * r.groups + n) * sizeof(double);
    *count = r.groups;
    return RMHIP_OK;
}

int rmhip_union(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int stable, size_t* count, double* values_host, size_t* ia_count, double* ia_host, size_t* ib_count,
                double* ib_host) {
    CTX_OR_FAIL(ctx);
    if (!count || !ia_count || !ib_count) return fail(RMHIP_ERR_INVALID, "union: null count");
    Buffer ab, bb;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    const u64 na = ab.numel, nb = bb.numel, n = na + nb;
    *count = *ia_count = *ib_count = 0;
    if (n == 0) return RMHIP_OK;
    if (!values_host || (na && !ia_host) || (nb && !ib_host)) return fail(RMHIP_ERR_INVALID, "union: null output");
    // union.rs:491-544 + 1238-1279: the CPU's map over a's elements, then b's, is `unique` of the two in sequence, first occurrences;
    // a value whose first occurrence lies in a reports that position in ia, the others their position in b in ib - both in output order
    std::shared_ptr<Allocation> cat;
    RMHIP_TRY(c->alloc_device(n, &cat));
    if (na) RMHIP_HIP_CHECK(hipMemcpyAsync(cat->ptr, ab.data(), na * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (nb) RMHIP_HIP_CHECK(hipMemcpyAsync(cat->ptr + na, bb.data(), nb * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    UniqueDev r;
    RMHIP_TRY(unique_device(c, cat->ptr, n, stable, 0, &r));
    std::vector<double> first(r.groups);
    RMHIP_HIP_CHECK(hipMemcpyAsync(values_host, r.dv, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(first.data(), r.dia, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    size_t ia = 0, ib = 0;
    for (u64 g = 0; g < r.groups; ++g) {
        if (first[g] <= (double)na) ia_host[ia++] = first[g];
        else ib_host[ib++] = first[g] - (double)na;
    }
    c->tel.download_bytes += 2 * r.groups * sizeof(double);
    *count = r.groups, *ia_count = ia, *ib_count = ib;
    return RMHIP_OK;
}

int rmhip_setdiff(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int stable, size_t* count, double* values_host, double* ia_host) {
    CTX_OR_FAIL(ctx);
    if (!count) return fail(RMHIP_ERR_INVALID, "setdiff: null count");
    Buffer ab, bb;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    const u64 na = ab.numel, nb = bb.numel;
    *count = 0;
    if (na == 0) return RMHIP_OK;
    if (!values_host || !ia_host) return fail(RMHIP_ERR_INVALID, "setdiff: null output");
    // setdiff.rs:463-496: a's distinct values (first occurrences, in the requested order) whose key does not occur in b
    UniqueDev r;
    RMHIP_TRY(unique_device(c, ab.data(), na, stable, 0, &r));
    std::vector<double> v(r.groups), first(r.groups);
    std::vector<unsigned char> hit(r.groups, 0);
    if (nb) {
        SortSpace ws;
        RMHIP_TRY(sort_lines(c, bb.data(), Lines{1, nb, 1}, 0, 0, &ws));
        std::shared_ptr<Allocation> m;
        RMHIP_TRY(c->alloc_device(r.groups + (r.groups + 7) / 8, &m));
        unsigned char* dmask = (unsigned char*)(m->ptr + r.groups);
        hipLaunchKernelGGL(k_ismember, dim3((unsigned)((r.groups + 255) / 256)), dim3(256), 0, c->stream, r.dv, r.groups, ws.keys, ws.pos, nb, dmask, m->ptr);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        RMHIP_HIP_CHECK(hipMemcpyAsync(hit.data(), dmask, r.groups, hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    RMHIP_HIP_CHECK(hipMemcpyAsync(v.data(), r.dv, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(first.data(), r.dia, r.groups * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    size_t kept = 0;
    for (u64 g = 0; g < r.groups; ++g)
        if (!hit[g]) values_host[kept] = v[g], ia_host[kept] = first[g], ++kept;
    c->tel.download_bytes += 2 * r.groups * sizeof(double);
    *count = kept;
    return RMHIP_OK;
}

int rmhip_ismember(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, unsigned char* mask_host, double* loc_host) {
    CTX_OR_FAIL(ctx);
    Buffer ab, bb;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    const u64 na = ab.numel, nb = bb.numel;
    if (na == 0) return RMHIP_OK;
    if (!mask_host || !loc_host) return fail(RMHIP_ERR_INVALID, "ismember: null output");
    if (nb == 0) {
        std::memset(mask_host, 0, na);
        for (u64 i = 0; i < na; ++i) loc_host[i] = 0.0;
        return RMHIP_OK;
    }
    SortSpace ws;
    RMHIP_TRY(sort_lines(c, bb.data(), Lines{1, nb, 1}, 0, 0, &ws));
    std::shared_ptr<Allocation> outs;
    RMHIP_TRY(c->alloc_device(na + (na + 7) / 8, &outs));  // loc (f64) | mask (bytes)
    double* dloc = outs->ptr;
    unsigned char* dmask = (unsigned char*)(dloc + na);
    hipLaunchKernelGGL(k_ismember, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, c->stream, ab.data(), na, ws.keys, ws.pos, nb, dmask, dloc);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    RMHIP_HIP_CHECK(hipMemcpyAsync(loc_host, dloc, na * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(mask_host, dmask, na, hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->tel.download_bytes += na * 9;
    return RMHIP_OK;
}

int rmhip_sort_rows(rmhip_ctx* ctx, rmhip_buf a, const size_t* column_index, const int* column_descend, size_t n_columns, int by_abs, rmhip_buf* sorted,
                    rmhip_buf* indices) {
    CTX_OR_FAIL(ctx);
    if (!sorted || !indices || (n_columns && (!column_index || !column_descend))) return fail(RMHIP_ERR_INVALID, "sort_rows: null argument");
    *sorted = *indices = 0;
    Buffer ab, sb, ib;
    RMHIP_TRY(c->get(a, &ab));
    for (size_t d = 2; d < ab.shape.size(); ++d)
        if (ab.shape[d] != 1) return fail(RMHIP_ERR_INVALID, "sortrows: input must be a 2-D matrix on the provider path");
    // rows_cols_for_shape, sortrows_host.rs:67-73
    const u64 rows = ab.shape.empty() ? 1 : (ab.shape.size() == 1 ? std::max<size_t>(ab.shape[0], 1) : ab.shape[0]);
    const u64 cols = ab.shape.size() < 2 ? 1 : ab.shape[1];
    if (rows * cols != ab.numel) return fail(RMHIP_ERR_SHAPE, "sortrows: tensor data length %zu does not match its shape", ab.numel);
    RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), sorted, &sb));
    const size_t ishape[2] = {(size_t)rows, 1};
    int rc = c->new_buffer(ishape, 2, indices, &ib);
    std::shared_ptr<Allocation> pm;
    u32* perm = nullptr;
    if (rc == RMHIP_OK && rows > 1 && cols > 0 && ab.numel > 0) {
        SortSpace ws;
        rc = sort_alloc(c, 1, rows, &ws);
        if (rc == RMHIP_OK) rc = c->alloc_device(rows, &pm);  // two u32 arrays of `rows`
        u32 *cur = nullptr, *nxt = pm ? (u32*)pm->ptr : nullptr;
        for (size_t s = n_columns; s-- > 0 && rc == RMHIP_OK;) {  // the last key first: every pass is stable (ties keep the ranks of the pass before)
            if (column_index[s] >= cols) continue;               // sortrows_host.rs:86-88
            hipLaunchKernelGGL(k_rows_keys, dim3((unsigned)((ws.total + 255) / 256)), dim3(256), 0, c->stream, ab.data(), rows, (u64)column_index[s], cur, column_descend[s] ? 1 : 0,
                               by_abs ? 1 : 0, ws.keys, ws.pos, ws.total);
            c->tel.kernel_launches++;
            rc = sort_pairs(c, &ws);
            if (rc != RMHIP_OK) break;
            hipLaunchKernelGGL(k_rows_compose, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, cur, ws.pos, rows, nxt);
            c->tel.kernel_launches++;
            u32* was = cur;
            cur = nxt;
            nxt = was ? was : (u32*)pm->ptr + rows;
        }
        perm = cur;
        if (rc == RMHIP_OK && perm) {
            hipLaunchKernelGGL(k_rows_emit, dim3((unsigned)((ab.numel + 255) / 256)), dim3(256), 0, c->stream, ab.data(), perm, rows, (u64)ab.numel, sb.data(), ib.data());
            c->tel.kernel_launches++;
            if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "sort_rows: launch failed");
            RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the permutation arrays are released on return
        }
    }
    if (rc == RMHIP_OK && !perm && ab.numel > 0) {  // nothing to order: the rows as they are, identity indices (sortrows_host.rs:33-39)
        hipLaunchKernelGGL(k_rows_emit, dim3((unsigned)((ab.numel + 255) / 256)), dim3(256), 0, c->stream, ab.data(), (const u32*)nullptr, rows, (u64)ab.numel, sb.data(), ib.data());
        c->tel.kernel_launches++;
    } else if (rc == RMHIP_OK && ab.numel == 0 && ib.numel > 0) {  // rows without columns: identity indices
        hipLaunchKernelGGL(k_iota1, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, ib.data(), rows);
        c->tel.kernel_launches++;
    }
    if (rc != RMHIP_OK) {
        rmhip_free(ctx, *sorted);
        if (*indices) rmhip_free(ctx, *indices);
        *sorted = *indices = 0;
    }
    return rc;
}
