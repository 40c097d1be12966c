// signal_ops.hip -- convolutions and window generators of the signal builtins.
//   conv1d            crates/runmat-accelerate-api/src/lib.rs:2535-2542   (builtins/math/signal/conv.rs:481-517; simple_provider.rs:1780-1842, 6015-6064)
//   conv2d            lib.rs:2543-2550    (builtins/math/signal/conv2.rs:595-640; simple_provider.rs:6065-6154)
//   hann_window / hamming_window / blackman_window   lib.rs:1797-1807   (simple_provider.rs:95-120, 6453-6472)
//   iir_filter        lib.rs:2551-2559    (builtins/math/signal/filter.rs:1119-1222, 1311-1319, 1370-1460)
//   imfilter          lib.rs:1809-1817    (builtins/image/filters/imfilter.rs:476-545, 620-783)
//   interp1           lib.rs:2458-2463    (runmat-accelerate/src/simple_provider.rs:1396-1472, 8135-8204)
//   polyval           lib.rs:1652-1660    (builtins/math/poly/polyval.rs:886-905, 352-435)
//   moving_window     lib.rs:2852-2857    (builtins/math/reduction/moving.rs:737-825, 929-1003, 1198-1237, 1282-1323)
// The convolutions are DIRECT sums in the CPU's order (output n receives a[i] * b[n - i] for i ascending, every product rounded before it
// is added - this file keeps contraction off): bit-exact against the oracle.  One thread per output point, neighbouring threads read
// neighbouring signal points, the kernel operand is staged in LDS when it fits.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

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
constexpr int kB = 256;
constexpr u64 LDS_TAPS = 4096;  // kernel operands up to this many points are staged in LDS (32 KiB)

// out[o] = sum over i ascending of a[i] * b[n - i], n = start + o
__global__ void __launch_bounds__(kB) k_conv1d(const double* __restrict__ a, u64 la, const double* __restrict__ b, u64 lb, u64 start, u64 len, int b_in_lds,
                                               double* __restrict__ out) {
    extern __shared__ double taps[];
    if (b_in_lds) {
        for (u64 j = threadIdx.x; j < lb; j += kB) taps[j] = b[j];
        __syncthreads();
    }
    const double* bb = b_in_lds ? taps : b;
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= len) return;
    const u64 n = start + o;
    const u64 lo = n >= lb - 1 ? n - (lb - 1) : 0, hi = n < la - 1 ? n : la - 1;
    double acc = 0.0;
    // (32-bit trip count, two running pointers: the loop was 20 instructions per tap with 64-bit index arithmetic)
    const double* ap = a + lo;
    const double* bp = bb + (n - lo);
    const unsigned cnt = (unsigned)(hi - lo + 1);
#pragma unroll 4
    for (unsigned k = 0; k < cnt; ++k) {
        const double p = ap[k] * *(bp - k);
        acc = acc + p;
    }
    out[o] = acc;
}

// out(r, c) of the window starting at (r0, c0) of the full result: sum over ac, ar ascending of a(ar, ac) * b(B_r - 1 - (R - ar), B_c - 1 - (C - ac))
__global__ void __launch_bounds__(kB) k_conv2d(const double* __restrict__ a, u64 ar_n, u64 ac_n, const double* __restrict__ b, u64 br_n, u64 bc_n, u64 r0, u64 c0,
                                               u64 rows, u64 cols, int b_in_lds, double* __restrict__ out) {
    extern __shared__ double taps[];
    if (b_in_lds) {
        for (u64 j = threadIdx.x; j < br_n * bc_n; j += kB) taps[j] = b[j];
        __syncthreads();
    }
    const double* bb = b_in_lds ? taps : b;
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= rows * cols) return;
    u64 R, Cc;
    if (rows * cols <= 0xffffffffull) {
        const unsigned q = (unsigned)o / (unsigned)rows;
        R = r0 + ((unsigned)o - q * (unsigned)rows), Cc = c0 + q;
    } else {
        R = r0 + o % rows, Cc = c0 + o / rows;
    }
    const u64 ac_lo = Cc >= bc_n - 1 ? Cc - (bc_n - 1) : 0, ac_hi = Cc < ac_n - 1 ? Cc : ac_n - 1;
    const u64 ar_lo = R >= br_n - 1 ? R - (br_n - 1) : 0, ar_hi = R < ar_n - 1 ? R : ar_n - 1;
    double acc = 0.0;
    const unsigned ncol = (unsigned)(ac_hi - ac_lo + 1), nrow = (unsigned)(ar_hi - ar_lo + 1);
    const double* acol = a + ac_lo * ar_n + ar_lo;
    const double* bcol = bb + (bc_n - 1 - (Cc - ac_lo)) * br_n + (br_n - 1 - (R - ar_lo));  // the tap of (ar_lo, ac_lo); rows and columns both ascend
    for (unsigned c = 0; c < ncol; ++c) {
#pragma unroll 4
        for (unsigned k = 0; k < nrow; ++k) {
            const double p = acol[k] * bcol[k];
            acc = acc + p;
        }
        acol += ar_n;
        bcol += br_n;
    }
    out[o] = acc;
}

// The same sums on an LDS-staged patch (see k_imfilter_tile below for why): 64 rows x 16 columns of outputs per workgroup, four per thread.  A
// term whose sample lies outside `a` is skipped, not added as a zero - the CPU never forms that product (0 * Inf would be NaN).
constexpr int CONV_TX = 64, CONV_TY = 16;
__global__ void __launch_bounds__(kB) k_conv2d_tile(const double* __restrict__ a, u64 ar_n, u64 ac_n, const double* __restrict__ b, int br_n, int bc_n, u64 r0, u64 c0,
                                                    u64 rows, u64 cols, double* __restrict__ out) {
    extern __shared__ double patch[];
    const int W = CONV_TX + br_n - 1, H = CONV_TY + bc_n - 1;
    // full-result coordinates of this tile's first output, and the sample its first tap reads (negative: outside)
    const long long R0 = (long long)(r0 + (u64)blockIdx.x * CONV_TX), C0 = (long long)(c0 + (u64)blockIdx.y * CONV_TY);
    const long long ar0 = R0 - (br_n - 1), ac0 = C0 - (bc_n - 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ly = wave; ly < H; ly += kB / 64) {
        const long long ac = ac0 + ly;
        const bool col_in = ac >= 0 && ac < (long long)ac_n;
        for (int lx = lane; lx < W; lx += 64) {
            const long long ar = ar0 + lx;
            patch[ly * W + lx] = (col_in && ar >= 0 && ar < (long long)ar_n) ? a[(u64)ac * ar_n + (u64)ar] : 0.0;
        }
    }
    __syncthreads();
    constexpr int OUTS = CONV_TY / (kB / 64);
    double acc[OUTS];
#pragma unroll
    for (int j = 0; j < OUTS; ++j) acc[j] = 0.0;
    // the taps whose sample exists: kr in [kr_lo, kr_hi] for this thread's row, kc in [kc_lo, kc_hi] for each of its columns
    const long long ar_first = ar0 + lane;
    const int kr_lo = ar_first < 0 ? (int)(-ar_first) : 0;
    const long long kr_room = (long long)ar_n - 1 - ar_first;
    const int kr_hi = kr_room < br_n - 1 ? (int)kr_room : br_n - 1;
    for (int kc = 0; kc < bc_n; ++kc) {
        bool col_ok[OUTS];
#pragma unroll
        for (int j = 0; j < OUTS; ++j) {
            const long long ac = ac0 + wave + j * (kB / 64) + kc;
            col_ok[j] = ac >= 0 && ac < (long long)ac_n;
        }
        const double* row = patch + (wave + kc) * W + lane;
        for (int kr = 0; kr < br_n; ++kr) {
            const double bv = b[kc * br_n + kr];  // uniform: a scalar load
            const bool row_ok = kr >= kr_lo && kr <= kr_hi;
#pragma unroll
            for (int j = 0; j < OUTS; ++j) {
                const double p = row[j * (kB / 64) * W + kr] * bv;
                const double next = acc[j] + p;
                acc[j] = (row_ok && col_ok[j]) ? next : acc[j];
            }
        }
    }
    const u64 orow = (u64)blockIdx.x * CONV_TX + lane;
    if (orow >= rows) return;
#pragma unroll
    for (int j = 0; j < OUTS; ++j) {
        const u64 ocol = (u64)blockIdx.y * CONV_TY + wave + j * (kB / 64);
        if (ocol < cols) out[ocol * rows + orow] = acc[j];
    }
}

__global__ void __launch_bounds__(kB) k_window(int kind, u64 len, double denom, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= len) return;
    const double phase = 2.0 * 3.14159265358979323846 * (double)i / denom;
    double v;
    if (kind == 0) v = 0.5 - 0.5 * cos(phase);
    else if (kind == 1) v = 0.54 - 0.46 * cos(phase);
    else v = 0.42 - 0.5 * cos(phase) + 0.08 * cos(2.0 * phase);
    out[i] = v;
}

// moving-window statistics (moving.rs:737-825, 929-1003, 1198-1237, 1282-1323): one thread per output element walks its window in
// ascending position - the CPU's order, every operation rounded as written (sums fold from -0.0, products from 1.0, Welford's running
// mean / M2 for std and var) - so results are bit-exact.  Neighbouring threads are neighbouring lines: coalesced for a window along any
// dimension but the first; along the first, neighbouring outputs share all but two window points (cache hits).
constexpr int MED_MAX = 64;  // median: the window's values are insertion-sorted in a per-thread array
struct MovingArgs {
    u64 pre, len, post, out_len, before, after, total;
    int op, endpoints, nan_omit, population;
    double fill;
};

__device__ __forceinline__ double kth_fill(const double* sorted, int n, double fill, u64 fill_count, u64 k) {
    int less = 0;
    while (less < n && sorted[less] < fill) ++less;
    int equal = 0;
    while (less + equal < n && sorted[less + equal] == fill) ++equal;
    if (k < (u64)less) return sorted[k];
    if (k < (u64)(less + equal) + fill_count) return fill;
    return sorted[k - fill_count];
}

template <int OP>
__global__ void __launch_bounds__(kB) k_moving(const double* __restrict__ x, MovingArgs A, double* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * kB + threadIdx.x;
    if (e >= A.total) return;
    u64 i, p, o;
    if (A.total <= 0xffffffffull) {  // 32-bit divisions when the output has fewer than 2^32 elements (a 64-bit one is ~40 instructions)
        const unsigned e32 = (unsigned)e, pre32 = (unsigned)A.pre, ol32 = (unsigned)A.out_len, r32 = e32 / pre32;
        i = e32 - r32 * pre32, o = r32 / ol32, p = r32 - (unsigned)o * ol32;
    } else {
        const u64 r = e / A.pre;
        i = e % A.pre, p = r % A.out_len, o = r / A.out_len;
    }
    const long long center = (long long)(A.endpoints == 1 ? p + A.before : p);
    const long long start = center - (long long)A.before, end = center + (long long)A.after;
    long long s0 = start < 0 ? 0 : start, e0 = end + 1;
    if (s0 > (long long)A.len) s0 = (long long)A.len;
    if (e0 < 0) e0 = 0;
    if (e0 > (long long)A.len) e0 = (long long)A.len;
    u64 fc = 0;
    if (A.endpoints == 2) fc = (u64)(start < 0 ? -start : 0) + (u64)(end >= (long long)A.len ? end - (long long)A.len + 1 : 0);
    const double* src = x + i + o * A.pre * A.len;
    double sum = -0.0, prod = 1.0, mn = INFINITY, mx = -INFINITY, mean = 0.0, m2 = 0.0;
    double med[OP == 5 ? MED_MAX : 1];
    unsigned n = 0;  // (window lengths fit 32 bits: checked by the entry point)
    bool saw_nan = false;
    const double* wp = src + (u64)s0 * A.pre;  // a running pointer and a 32-bit trip count instead of 64-bit index arithmetic per point
    const unsigned wlen = (unsigned)(e0 - s0);
    // the window in batches of eight loads issued together: with one load per trip every point waited out a memory latency (the NaN test
    // between two loads keeps the compiler from batching them itself)
    for (unsigned w0 = 0; w0 < wlen && !saw_nan; w0 += 8, wp += 8 * A.pre) {
        double batch[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) batch[u] = w0 + u < wlen ? wp[(u64)u * A.pre] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
        if (saw_nan || w0 + u >= wlen) continue;
        const double v = batch[u];
        if (isnan(v)) {
            if (!A.nan_omit) saw_nan = true;
            continue;
        }
        ++n;
        if (OP == 0 || OP == 1) sum = sum + v;
        if (OP == 2) prod = prod * v;
        if (OP == 3) mn = fmin(mn, v);
        if (OP == 4) mx = fmax(mx, v);
        if (OP == 5) {  // insertion into the sorted prefix (a stable sort of non-NaN values: equal values keep their order, as the CPU's)
            int k = (int)n - 1;
            while (k > 0 && med[k - 1] > v) med[k] = med[k - 1], --k;
            med[k] = v;
        }
        if (OP == 6 || OP == 7) {
            const double d = v - mean;
            mean = mean + d / (double)n;
            const double d2 = v - mean;
            m2 = m2 + d * d2;
        }
    }
    }
    double res;
    if (fc && (saw_nan || (isnan(A.fill) && !A.nan_omit))) {
        res = NAN;
    } else {
        if (fc && isnan(A.fill)) fc = 0, saw_nan = false;  // omitted NaN padding: the values alone
        if (saw_nan) {
            res = NAN;
        } else if (n == 0 && fc == 0) {
            res = OP == 0 ? 0.0 : (OP == 2 ? 1.0 : NAN);
        } else if (OP == 0) {
            res = fc ? sum + A.fill * (double)fc : sum;
        } else if (OP == 1) {
            res = fc ? (sum + A.fill * (double)fc) / (double)(n + fc) : sum / (double)n;
        } else if (OP == 2) {
            res = fc ? prod * A.fill : prod;  // (the entry point only lets fills through whose power is the fill itself: 0, 1)
        } else if (OP == 3) {
            res = fc ? fmin(mn, A.fill) : mn;
        } else if (OP == 4) {
            res = fc ? fmax(mx, A.fill) : mx;
        } else if (OP == 5) {
            const u64 total = n + fc, mid = total / 2;
            if (fc == 0) res = total % 2 ? med[mid] : (med[mid - 1] + med[mid]) / 2.0;
            else res = total % 2 ? kth_fill(med, (int)n, A.fill, fc, mid) : (kth_fill(med, (int)n, A.fill, fc, mid - 1) + kth_fill(med, (int)n, A.fill, fc, mid)) / 2.0;
        } else {
            u64 cnt = n;
            if (fc) {
                if (cnt == 0) {
                    cnt = fc, mean = A.fill, m2 = 0.0;
                } else {
                    const u64 total = cnt + fc;
                    const double d = A.fill - mean;
                    mean = mean + d * ((double)fc / (double)total);
                    m2 = m2 + d * d * ((double)cnt * (double)fc / (double)total);
                    cnt = total;
                }
            }
            const double den = A.population ? (double)cnt : (cnt > 1 ? (double)(cnt - 1) : (double)cnt);
            res = m2 / den;
            if (OP == 6) res = sqrt(res);
        }
    }
    out[e] = res;
}

// polyval (builtins/math/poly/polyval.rs:886-905): Horner's rule acc = acc * x + c over the coefficients, product rounded before the sum;
// with `mu` the point is first centred and scaled the way the CPU's complex division by (scale + 0i) does it:
// ((x - mean) * scale) / (scale * scale).  The CPU runs this in complex arithmetic; for real data the real part is this recurrence as long
// as every intermediate stays finite (an infinite one makes the imaginary lane inf * 0 = NaN there): `nonfinite` records that.
__global__ void __launch_bounds__(kB) k_polyval(const double* __restrict__ coef, u64 m, const double* __restrict__ x, u64 n, int has_mu, double mean, double scale,
                                                int c_in_lds, double* __restrict__ out, int* __restrict__ nonfinite) {
    extern __shared__ double taps[];
    if (c_in_lds) {
        for (u64 j = threadIdx.x; j < m; j += kB) taps[j] = coef[j];
        __syncthreads();
    }
    const double* cc = c_in_lds ? taps : coef;
    const u64 e = (u64)blockIdx.x * kB + threadIdx.x;
    if (e >= n) return;
    double v = x[e];
    if (has_mu) {
        const double t = v - mean;
        const double num = t * scale + 0.0, den = scale * scale + 0.0;
        v = num / den;
    }
    double acc = 0.0;
    bool bad = !isfinite(v);
    for (u64 j = 0; j < m; ++j) {
        const double p = acc * v;
        acc = p + cc[j];
        bad |= !isfinite(acc);
    }
    out[e] = acc;
    if (bad) *nonfinite = 1;
}

// filter(b, a, x) along a dimension (builtins/math/signal/filter.rs:1119-1222): direct form II transposed, y = b0 x + s0,
// s[i-1] = (b[i] x + s[i]) - a[i] y - a recurrence along the dimension, so ONE THREAD PER CHANNEL walks it in order with the states in
// registers (order <= 8) or a per-thread array; neighbouring threads are neighbouring channels.  Every operation as the CPU rounds it.
template <int MAXO>
__global__ void __launch_bounds__(kB) k_iir(const double* __restrict__ x, const double* __restrict__ zi, const double* __restrict__ coef, int order, u64 leading,
                                            u64 dim_len, u64 channels, double* __restrict__ y, double* __restrict__ zf) {
    const u64 ch = (u64)blockIdx.x * kB + threadIdx.x;
    if (ch >= channels) return;
    const u64 l = ch % leading, t = ch / leading;
    const u64 state_len = (u64)order - 1;
    double bn[MAXO], an[MAXO], st[MAXO];
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        bn[i] = i < order ? coef[i] : 0.0;
        an[i] = i < order ? coef[order + i] : 0.0;
        st[i] = (zi && (u64)i < state_len) ? zi[l + (u64)i * leading + t * leading * state_len] : 0.0;
    }
    const double* src = x + t * dim_len * leading + l;
    double* dst = y + t * dim_len * leading + l;
    // eight samples are fetched together ahead of the recurrence that consumes them: with one load per step every step waited out a memory
    // latency (4.0 ms for 8192 channels x 8192 samples)
    for (u64 s0 = 0; s0 < dim_len; s0 += 8) {
        double xb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xb[u] = s0 + u < dim_len ? src[(s0 + u) * leading] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u >= dim_len) break;
            const double xn = xb[u];
            const double yv = bn[0] * xn + st[0];
            dst[(s0 + u) * leading] = yv;
#pragma unroll
            for (int i = 1; i < MAXO; ++i) {
                if (i < order) {
                    const double next = (u64)i < state_len ? st[i] : 0.0;
                    const double p = bn[i] * xn, q = an[i] * yv;
                    st[i - 1] = (p + next) - q;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXO; ++i)
        if ((u64)i < state_len) zf[l + (u64)i * leading + t * leading * state_len] = st[i];
}

// interp1 (runmat-accelerate/src/simple_provider.rs:1396-1472, 8135-8204): one thread per (series, query) - a binary search of the strictly
// increasing sample coordinates, then the CPU's four operations (linear) or its nearer-neighbour rule (ties to the left).
__device__ __forceinline__ long long interp_search(const double* __restrict__ x, u64 n, double q, bool* exact) {  // Rust's binary_search_by on distinct keys
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (x[mid] < q) lo = mid + 1;
        else hi = mid;
    }
    *exact = lo < n && x[lo] == q;
    return (long long)lo;  // Ok(lo) when exact, else Err(lo): the insertion point
}

__global__ void __launch_bounds__(kB) k_interp1(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ xq, u64 n, u64 qlen, u64 total, int nearest,
                                                int extrapolation, double fill, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    const double q = xq[o % qlen];
    const double* ys = y + (o / qlen) * n;
    const double oor = extrapolation == 2 ? fill : NAN;  // interp1_out_of_range
    double r;
    if (!isfinite(q)) {
        r = NAN;
    } else if (!nearest) {
        long long piece = -1;
        const u64 last = n - 1;
        if (q < x[0]) piece = extrapolation == 1 ? 0 : -1;
        else if (q > x[last]) piece = extrapolation == 1 ? (long long)last - 1 : -1;
        else if (q == x[last]) piece = (long long)last - 1;
        else {
            bool exact;
            const long long idx = interp_search(x, n, q, &exact);
            if (exact) piece = idx < (long long)last - 1 ? idx : (long long)last - 1;
            else if (idx > 0 && idx < (long long)n) piece = idx - 1;
        }
        if (piece < 0) {
            r = oor;
        } else {
            const double h = x[piece + 1] - x[piece];
            const double t = (q - x[piece]) / h;
            const double d = ys[piece + 1] - ys[piece];
            const double p = t * d;
            r = ys[piece] + p;
        }
    } else if (q < x[0]) {
        r = extrapolation == 1 ? ys[0] : oor;
    } else if (q > x[n - 1]) {
        r = extrapolation == 1 ? ys[n - 1] : oor;
    } else {
        bool exact;
        const long long idx = interp_search(x, n, q, &exact);
        if (exact) {
            r = ys[idx];
        } else {
            const u64 left = idx > 0 ? (u64)idx - 1 : 0, right = (u64)idx < n - 1 ? (u64)idx : n - 1;
            r = fabs(q - x[left]) <= fabs(x[right] - q) ? ys[left] : ys[right];
        }
    }
    out[o] = r;
}

// imfilter (builtins/image/filters/imfilter.rs:476-545, 620-783): N-D correlation / convolution with the four padding rules.  One thread per
// output element walks the kernel's points in their storage order (first dimension fastest) and adds value * sample, the product rounded
// before the sum - the CPU's loop; up to four dimensions.
struct ImfArgs {
    int rank, padding, convolution;  // padding: 0 constant, 1 replicate, 2 symmetric, 3 circular
    long long img[4], ker[4], out[4], base[4], origin[4];
    u64 istride[4], kstride[4], total;
    double constant;
};

__device__ __forceinline__ long long imf_resolve(long long coord, long long len, int padding) {  // -1: outside under constant padding
    if (coord >= 0 && coord < len) return coord;
    if (padding == 0) return -1;
    if (padding == 1) return coord <= 0 ? 0 : len - 1;             // clamp_index
    if (padding == 3) {                                            // wrap_index
        long long c = coord % len;
        return c < 0 ? c + len : c;
    }
    if (len == 1) return 0;                                        // reflect_index
    const long long period = 2 * len - 2;
    long long v = coord % period;
    if (v < 0) v += period;
    return v >= len ? period - v : v;
}

__global__ void __launch_bounds__(kB) k_imfilter(const double* __restrict__ image, const double* __restrict__ kernel, ImfArgs A, int k_in_lds, double* __restrict__ out) {
    extern __shared__ double taps[];
    const u64 ktotal = (u64)(A.ker[0] * A.ker[1] * A.ker[2] * A.ker[3]);
    if (k_in_lds) {
        for (u64 j = threadIdx.x; j < ktotal; j += kB) taps[j] = kernel[j];
        __syncthreads();
    }
    const double* kk = k_in_lds ? taps : kernel;
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= A.total) return;
    long long oc[4];
    u64 r = o;
    for (int d = 0; d < 4; ++d) {
        oc[d] = (long long)(r % (u64)A.out[d]) + A.base[d] - A.origin[d];  // image coordinate of kernel index 0 along d
        r /= (u64)A.out[d];
    }
    double sum = 0.0;
    bool interior = true;
    for (int d = 0; d < 4; ++d) interior = interior && oc[d] >= 0 && oc[d] + A.ker[d] <= A.img[d];
    if (interior) {
        // the kernel's footprint lies inside the image (all but a border of outputs): no padding rule to consult. The tap odometer is
        // wave-uniform, so it lives on the scalar unit (and the coefficients come by scalar loads); sixteen image loads are issued per
        // wait - one dependent load per trip waited out a full memory latency per tap. Same taps, same order.
        const unsigned n0 = (unsigned)A.ker[0], n1 = (unsigned)A.ker[1], n2 = (unsigned)A.ker[2];
        const double* ip = image + (u64)oc[0] + (u64)oc[1] * A.istride[1] + (u64)oc[2] * A.istride[2] + (u64)oc[3] * A.istride[3];
        const long long wrap1 = A.istride[1] - (long long)n0, wrap2 = A.istride[2] - (long long)n1 * A.istride[1],
                        wrap3 = A.istride[3] - (long long)n2 * A.istride[2];
        unsigned k0 = 0, k1 = 0, k2 = 0;
        long long off = 0;
        constexpr int BATCH = 16;
        for (u64 t = 0; t < ktotal; t += BATCH) {
            double v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const bool live = t + j < ktotal;
                v[j] = ip[live ? off : 0];
                ++off;
                if (++k0 == n0) {
                    k0 = 0;
                    off += wrap1;
                    if (++k1 == n1) {
                        k1 = 0;
                        off += wrap2;
                        if (++k2 == n2) {
                            k2 = 0;
                            off += wrap3;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j)
                if (t + j < ktotal) {
                    const double p = kernel[A.convolution ? ktotal - 1 - (t + j) : t + j] * v[j];  // convolution reads the kernel back to front
                    sum = sum + p;
                }
        }
        out[o] = sum;
        return;
    }
    for (long long k3 = 0; k3 < A.ker[3]; ++k3) {
        const long long c3 = imf_resolve(oc[3] + k3, A.img[3], A.padding);
        for (long long k2 = 0; k2 < A.ker[2]; ++k2) {
            const long long c2 = imf_resolve(oc[2] + k2, A.img[2], A.padding);
            for (long long k1 = 0; k1 < A.ker[1]; ++k1) {
                const long long c1 = imf_resolve(oc[1] + k1, A.img[1], A.padding);
                const bool outside_hi = c1 < 0 || c2 < 0 || c3 < 0;
                const u64 ibase = outside_hi ? 0 : (u64)c1 * A.istride[1] + (u64)c2 * A.istride[2] + (u64)c3 * A.istride[3];
                const u64 kbase = A.convolution ? (u64)(A.ker[1] - 1 - k1) * A.kstride[1] + (u64)(A.ker[2] - 1 - k2) * A.kstride[2] + (u64)(A.ker[3] - 1 - k3) * A.kstride[3]
                                                : (u64)k1 * A.kstride[1] + (u64)k2 * A.kstride[2] + (u64)k3 * A.kstride[3];
                for (long long k0 = 0; k0 < A.ker[0]; ++k0) {
                    const long long c0 = imf_resolve(oc[0] + k0, A.img[0], A.padding);
                    const double sample = (outside_hi || c0 < 0) ? A.constant : image[ibase + (u64)c0];
                    const double kv = kk[kbase + (u64)(A.convolution ? A.ker[0] - 1 - k0 : k0)];
                    const double p = kv * sample;
                    sum = sum + p;
                }
            }
        }
    }
    out[o] = sum;
}

// The same sum for a filter of two dimensions (every image filter in practice), tiled: a workgroup stages the (64 + k0 - 1) x (16 + k1 - 1)
// patch of the image its 64 x 16 outputs read into LDS once - the padding rule is applied while staging, so the sum itself has no border case -
// and each output then reads its taps from LDS.  The one-thread-per-output kernel re-reads every sample k0*k1 times through the vector cache,
// which eight resident workgroups overflow: it ran at the L2's bandwidth (8192^2, 5x5: 1.8 ms for 13 GB of cache reads).  Taps in the CPU's order.
constexpr int IMF_TX = 64, IMF_TY = 16;
__global__ void __launch_bounds__(kB) k_imfilter_tile(const double* __restrict__ image, const double* __restrict__ kernel, ImfArgs A, double* __restrict__ out) {
    extern __shared__ double patch[];
    const int n0 = (int)A.ker[0], n1 = (int)A.ker[1], W = IMF_TX + n0 - 1, H = IMF_TY + n1 - 1;
    const long long x0 = (long long)blockIdx.x * IMF_TX, y0 = (long long)blockIdx.y * IMF_TY;
    const double* plane = image + (u64)blockIdx.z * A.istride[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ly = wave; ly < H; ly += kB / 64) {
        const long long c1 = imf_resolve(y0 + A.base[1] - A.origin[1] + ly, A.img[1], A.padding);
        for (int lx = lane; lx < W; lx += 64) {
            const long long c0 = imf_resolve(x0 + A.base[0] - A.origin[0] + lx, A.img[0], A.padding);
            patch[ly * W + lx] = (c0 < 0 || c1 < 0) ? A.constant : plane[(u64)c0 + (u64)c1 * A.istride[1]];
        }
    }
    __syncthreads();
    constexpr int ROWS = IMF_TY / (kB / 64);  // outputs per thread: rows wave, wave + 4, ...
    double sum[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) sum[j] = 0.0;
    const int ktotal = n0 * n1;
    int t = 0;
    for (int k1 = 0; k1 < n1; ++k1) {
        const double* row = patch + (wave + k1) * W + lane;
        for (int k0 = 0; k0 < n0; ++k0, ++t) {
            const double kv = kernel[A.convolution ? ktotal - 1 - t : t];  // uniform: a scalar load
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const double p = kv * row[j * (kB / 64) * W + k0];
                sum[j] = sum[j] + p;
            }
        }
    }
    const long long ox = x0 + lane;
    if (ox >= A.out[0]) return;
    double* oplane = out + (u64)blockIdx.z * (u64)A.out[0] * (u64)A.out[1];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const long long oy = y0 + wave + j * (kB / 64);
        if (oy < A.out[1]) oplane[(u64)ox + (u64)oy * (u64)A.out[0]] = sum[j];
    }
}

// polyder / polyint (simple_provider.rs:320-390, 544-585, 3137-3215): coefficient vectors, highest power first.  One thread per coefficient
// of the untrimmed result, the CPU's sums in the CPU's order (a convolution entry accumulates from 0.0 over the first operand's index
// ascending; the two rule terms are right-aligned and added - or subtracted - to 0.0 in turn).  An empty polynomial is [0].
struct PolyVec {
    const double* x;
    u64 n;  // as stored; the effective length is max(n, 1)
    __device__ u64 len() const { return n ? n : 1; }
    __device__ double at(u64 i) const { return n ? x[i] : 0.0; }
    __device__ u64 dlen() const { return n <= 1 ? 1 : n - 1; }                                      // poly_raw_derivative
    __device__ double dat(u64 i) const { return n <= 1 ? 0.0 : x[i] * (double)(n - 1 - i); }
};

template <bool DA, bool DB>  // entry k of conv(a or a', b or b')
__device__ double poly_conv_entry(const PolyVec& a, const PolyVec& b, u64 k) {
    const u64 la = DA ? a.dlen() : a.len(), lb = DB ? b.dlen() : b.len();
    const u64 lo = k >= lb - 1 ? k - (lb - 1) : 0, hi = k < la - 1 ? k : la - 1;
    double t = 0.0;
    for (u64 i = lo; i <= hi; ++i) {
        const double p = (DA ? a.dat(i) : a.at(i)) * (DB ? b.dat(k - i) : b.at(k - i));
        t = t + p;
    }
    return t;
}

// mode 0: p'; 1: p'q + pq'; 2: u'v - uv' (p = u, q = v); 3: v * v (q = v); 4: the integral of p with `constant` appended
__global__ void __launch_bounds__(kB) k_poly(PolyVec p, PolyVec q, int mode, double constant, u64 len, double* __restrict__ out) {
    const u64 k = (u64)blockIdx.x * kB + threadIdx.x;
    if (k >= len) return;
    double r = 0.0;
    if (mode == 0) {
        r = p.dat(k);
    } else if (mode == 3) {
        r = poly_conv_entry<false, false>(q, q, k);
    } else if (mode == 4) {
        r = k < p.n ? p.x[k] / (double)(p.n - k) : constant;
    } else {
        const u64 l1 = p.dlen() + q.len() - 1, l2 = p.len() + q.dlen() - 1;  // poly_add_real / poly_sub_real right-align the two terms
        if (k >= len - l1) r = r + poly_conv_entry<true, false>(p, q, k - (len - l1));
        if (k >= len - l2) {
            const double t2 = poly_conv_entry<false, true>(p, q, k - (len - l2));
            r = mode == 1 ? r + t2 : r - t2;
        }
    }
    out[k] = r;
}

// poly_trim_slice: the first coefficient with |c| > 1e-12 (a NaN does not count), or len if there is none
__global__ void __launch_bounds__(kB) k_poly_first(const double* __restrict__ x, u64 len, unsigned long long* __restrict__ first) {
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) best = len;
    __syncthreads();
    for (u64 i = threadIdx.x; i < len && i < best; i += kB)
        if (fabs(x[i]) > 1.0e-12) {
            atomicMin(&best, (unsigned long long)i);
            break;
        }
    __syncthreads();
    if (threadIdx.x == 0) *first = best;
}

inline unsigned grid_for(u64 n) { return (unsigned)((n + kB - 1) / kB); }

}  // namespace
}  // namespace rmhip

extern "C" {

int rmhip_conv1d(rmhip_ctx* ctx, rmhip_buf signal, rmhip_buf kernel, int mode, int column, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (mode < 0 || mode > 2) return fail(RMHIP_ERR_INVALID, "conv1d: mode %d", mode);
    Buffer ab, bb;
    RMHIP_TRY(c->get(signal, &ab));
    RMHIP_TRY(c->get(kernel, &bb));
    const u64 la = ab.numel, lb = bb.numel;
    u64 start = 0, len = 0;
    if (la && lb) {
        const u64 full = la + lb - 1;
        start = 0, len = full;
        if (mode == 1) start = (lb - 1) / 2, len = la;
        if (mode == 2) {
            if (la < lb) len = 0;
            else start = lb - 1, len = la - lb + 1;
        }
        if (len && start + len > full) len = full > start ? full - start : 0;
    }
    // conv1d_output_shape (simple_provider.rs:1780-1787): [1, len] or [len, 1], an empty result keeps the orientation
    const size_t shape[2] = {column ? (size_t)len : 1, column ? 1 : (size_t)len};
    if (len > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "conv1d: %llu outputs", len);  // before the output exists
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (len == 0) return RMHIP_OK;
    const int in_lds = lb <= LDS_TAPS;
    hipLaunchKernelGGL(k_conv1d, dim3(grid_for(len)), dim3(kB), in_lds ? lb * sizeof(double) : 0, c->stream, ab.data(), la, bb.data(), lb, start, len, in_lds, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_conv2d(rmhip_ctx* ctx, rmhip_buf signal, rmhip_buf kernel, int mode, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (mode < 0 || mode > 2) return fail(RMHIP_ERR_INVALID, "conv2d: mode %d", mode);
    Buffer ab, bb;
    RMHIP_TRY(c->get(signal, &ab));
    RMHIP_TRY(c->get(kernel, &bb));
    for (const Buffer* x : {&ab, &bb})  // ensure_diag_shape, simple_provider.rs:2394-2400
        for (size_t d = 2; d < x->shape.size(); ++d)
            if (x->shape[d] != 1) return fail(RMHIP_ERR_INVALID, "conv2d: input must be 2-D");
    auto rows_cols = [](const Buffer& x, u64* r, u64* cc) {  // simple_provider.rs:2402-2408
        *r = x.shape.empty() ? 1 : x.shape[0];
        *cc = x.shape.size() < 2 ? 1 : x.shape[1];
    };
    u64 ar_n, ac_n, br_n, bc_n;
    rows_cols(ab, &ar_n, &ac_n);
    rows_cols(bb, &br_n, &bc_n);
    u64 r0 = 0, c0 = 0, rows = 0, cols = 0;
    if (ab.numel == 0 || bb.numel == 0) {  // simple_provider.rs:6079-6093: [0, 0], or the signal's shape for 'same' (no points either way)
        if (mode == 1) rows = ar_n, cols = ac_n;
    } else {
        rows = ar_n + br_n - 1, cols = ac_n + bc_n - 1;
        if (mode == 1) r0 = (br_n - 1) / 2, c0 = (bc_n - 1) / 2, rows = ar_n, cols = ac_n;
        if (mode == 2) {
            if (ar_n < br_n || ac_n < bc_n) rows = cols = 0;
            else r0 = br_n - 1, c0 = bc_n - 1, rows = ar_n - br_n + 1, cols = ac_n - bc_n + 1;
        }
    }
    const size_t shape[2] = {(size_t)rows, (size_t)cols};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (ob.numel > 0x7fffffffull * kB) {  // (the output is registered by now: released here, the caller never learns its id)
        const size_t too_many = ob.numel;
        (void)rmhip_free(ctx, *out);
        *out = 0;
        return fail(RMHIP_ERR_UNSUPPORTED, "conv2d: %zu outputs", too_many);
    }
    const size_t patch_bytes = (size_t)(CONV_TX + br_n - 1) * (size_t)(CONV_TY + bc_n - 1) * sizeof(double);
    const u64 tiles_y = (cols + CONV_TY - 1) / CONV_TY;
    if (br_n <= 64 && bc_n <= 512 && patch_bytes <= 48u * 1024 && tiles_y <= 65535) {
        const dim3 grid((unsigned)((rows + CONV_TX - 1) / CONV_TX), (unsigned)tiles_y);
        hipLaunchKernelGGL(k_conv2d_tile, grid, dim3(kB), patch_bytes, c->stream, ab.data(), ar_n, ac_n, bb.data(), (int)br_n, (int)bc_n, r0, c0, rows, cols, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    const int in_lds = bb.numel <= LDS_TAPS;
    hipLaunchKernelGGL(k_conv2d, dim3(grid_for(ob.numel)), dim3(kB), in_lds ? bb.numel * sizeof(double) : 0, c->stream, ab.data(), ar_n, ac_n, bb.data(), br_n, bc_n, r0, c0,
                       rows, cols, in_lds, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_moving_window(rmhip_ctx* ctx, rmhip_buf a, int dim, size_t before, size_t after, int op, int endpoints, double fill, int nan_omit, int population,
                        const size_t* out_shape, size_t out_rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (out_rank && !out_shape)) return fail(RMHIP_ERR_INVALID, "moving_window: null argument");
    if (dim < 0 || op < 0 || op > 7 || endpoints < 0 || endpoints > 2) return fail(RMHIP_ERR_INVALID, "moving_window: dim %d, op %d, endpoints %d", dim, op, endpoints);
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    std::vector<size_t> shape = ab.shape;
    if ((size_t)dim >= shape.size()) shape.resize(dim + 1, 1);  // simple_provider.rs:1266-1269
    if ((size_t)dim >= out_rank) return fail(RMHIP_ERR_INVALID, "moving_window: dimension exceeds tensor rank");
    // the caller's output shape (moving.rs:1545-1570): the input's, the dimension trimmed by before + after for 'discard'
    std::vector<size_t> want = shape;
    if (endpoints == 1) want[dim] = shape[dim] > before + after ? shape[dim] - before - after : 0;
    std::vector<size_t> given(out_shape, out_shape + out_rank), wanted = want;  // compared without trailing singleton dimensions beyond `dim`
    while (given.size() > (size_t)dim + 1 && given.back() == 1) given.pop_back();
    while (wanted.size() > (size_t)dim + 1 && wanted.back() == 1) wanted.pop_back();
    if (given != wanted) return fail(RMHIP_ERR_SHAPE, "moving_window: output shape does not match the request");
    MovingArgs A{};
    A.pre = 1, A.post = 1;
    for (int k = 0; k < dim; ++k) A.pre *= shape[k];
    for (size_t k = dim + 1; k < shape.size(); ++k) A.post *= shape[k];
    A.len = shape[dim], A.out_len = want[dim], A.before = before, A.after = after;
    A.op = op, A.endpoints = endpoints, A.nan_omit = nan_omit ? 1 : 0, A.population = population ? 1 : 0, A.fill = fill;
    if (before > (1ull << 30) || after > (1ull << 30)) return fail(RMHIP_ERR_UNSUPPORTED, "moving_window: window %zu + %zu", before, after);
    if (op == 5 && std::min<u64>(A.len, before + after + 1) > (u64)MED_MAX)
        return fail(RMHIP_ERR_UNSUPPORTED, "moving_window: median over windows of more than %d points", MED_MAX);
    // prod with padding multiplies by fill^count (`powf`, moving.rs:991): only fills whose every power is the fill itself stay exact
    if (op == 2 && endpoints == 2 && !(fill == 0.0 || fill == 1.0 || std::isnan(fill)))
        return fail(RMHIP_ERR_UNSUPPORTED, "moving_window: product with a padding value other than 0, 1 or NaN");
    Buffer ob;
    RMHIP_TRY(c->new_buffer(out_shape, out_rank, out, &ob));
    A.total = ob.numel;
    if (ob.numel == 0) return RMHIP_OK;
    if (ob.numel > 0x7fffffffull * kB) {  // (the output is registered by now: released here, the caller never learns its id)
        const size_t too_many = ob.numel;
        (void)rmhip_free(ctx, *out);
        *out = 0;
        return fail(RMHIP_ERR_UNSUPPORTED, "moving_window: %zu outputs", too_many);
    }
    const dim3 grid(grid_for(ob.numel)), block(kB);
    switch (op) {
        case 0: hipLaunchKernelGGL(k_moving<0>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 1: hipLaunchKernelGGL(k_moving<1>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 2: hipLaunchKernelGGL(k_moving<2>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 3: hipLaunchKernelGGL(k_moving<3>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 4: hipLaunchKernelGGL(k_moving<4>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 5: hipLaunchKernelGGL(k_moving<5>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        case 6: hipLaunchKernelGGL(k_moving<6>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
        default: hipLaunchKernelGGL(k_moving<7>, grid, block, 0, c->stream, ab.data(), A, ob.data()); break;
    }
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_iir_filter(rmhip_ctx* ctx, rmhip_buf b, rmhip_buf a, rmhip_buf x, int dim, rmhip_buf zi_or_0, int unit_denominator, rmhip_buf* output,
                     rmhip_buf* final_state) {
    CTX_OR_FAIL(ctx);
    if (!output || !final_state) return fail(RMHIP_ERR_INVALID, "iir_filter: null output");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "iir_filter: dim must be >= 0");
    *output = *final_state = 0;
    Buffer bb, ab, xb, zb;
    RMHIP_TRY(c->get(b, &bb));
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(x, &xb));
    const size_t nb = bb.numel, na = unit_denominator ? 1 : ab.numel;
    if (nb == 0) return fail(RMHIP_ERR_INVALID, "iir_filter: numerator coefficients must not be empty");
    if (ab.numel == 0) return fail(RMHIP_ERR_INVALID, "iir_filter: denominator coefficients must not be empty");
    if (unit_denominator && ab.numel != 1) return fail(RMHIP_ERR_INVALID, "iir_filter: unit-denominator FIR path requires scalar denominator");
    const size_t order = std::max(nb, na), state_len = order - 1;
    if (order > 64) return fail(RMHIP_ERR_UNSUPPORTED, "iir_filter: order %zu", order);
    std::vector<size_t> shape = xb.shape;
    if ((size_t)dim >= shape.size()) shape.resize(dim + 1, 1);
    u64 leading = 1, trailing = 1;
    for (int k = 0; k < dim; ++k) leading *= shape[k];
    for (size_t k = dim + 1; k < shape.size(); ++k) trailing *= shape[k];
    const u64 dim_len = shape[dim], channels = leading * trailing;
    // a recurrence per channel: with few channels the chip idles and the host's one core is faster - the caller keeps those
    if (state_len > 0 && channels < 256 && xb.numel > 4096) return fail(RMHIP_ERR_UNSUPPORTED, "iir_filter: %llu channels", channels);
    std::vector<size_t> zshape = shape;  // filter_state_shape, filter.rs:1311-1319
    zshape[dim] = state_len;
    if (zi_or_0) {
        RMHIP_TRY(c->get(zi_or_0, &zb));
        const size_t rk = std::max(zshape.size(), zb.shape.size());
        for (size_t k = 0; k < rk; ++k)
            if ((k < zshape.size() ? zshape[k] : 1) != (k < zb.shape.size() ? zb.shape[k] : 1))
                return fail(RMHIP_ERR_SHAPE, "iir_filter: initial conditions are not compatible with the signal shape");
    }
    // coefficients: read back (<= 128 doubles), normalised as the CPU's complex division by (a0 + 0i) rounds them (filter.rs:1119-1138)
    std::vector<double> hb(nb), ha(ab.numel), coef(2 * order, 0.0);
    RMHIP_HIP_CHECK(hipMemcpyAsync(hb.data(), bb.data(), nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipMemcpyAsync(ha.data(), ab.data(), ab.numel * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    const double a0 = unit_denominator ? 1.0 : ha[0];
    if (a0 == 0.0) return fail(RMHIP_ERR_INVALID, "iir_filter: denominator coefficient a(1) must be non-zero");
    const double den = a0 * a0 + 0.0;
    for (size_t i = 0; i < nb; ++i) coef[i] = (hb[i] * a0 + 0.0) / den;
    coef[order] = 1.0;
    for (size_t i = 1; i < na; ++i) coef[order + i] = (ha[i] * a0 + 0.0) / den;
    Buffer yb, fb;
    RMHIP_TRY(c->new_buffer(xb.shape.data(), xb.shape.size(), output, &yb));
    int rc = c->new_buffer(zshape.data(), zshape.size(), final_state, &fb);
    if (rc == RMHIP_OK && state_len == 0) {
        if (yb.numel) rc = launch_scalar(c, RMHIP_SMUL, xb.data(), coef[0], yb.data(), yb.numel);  // gain * x (filter.rs:1172-1176)
    } else if (rc == RMHIP_OK && zi_or_0 && (dim_len == 0 || channels == 0)) {
        if (fb.numel) RMHIP_HIP_CHECK(hipMemcpyAsync(fb.data(), zb.data(), fb.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else if (rc == RMHIP_OK && channels > 0) {
        std::shared_ptr<Allocation> dc;
        rc = c->alloc_device(2 * order, &dc);
        if (rc == RMHIP_OK) {
            // (pageable source: the copy is staged before the call returns, so `coef` may go out of scope)
            RMHIP_HIP_CHECK(hipMemcpyAsync(dc->ptr, coef.data(), 2 * order * sizeof(double), hipMemcpyHostToDevice, c->stream));
            const double* zi = zi_or_0 ? zb.data() : nullptr;
            if (order <= 8)
                hipLaunchKernelGGL(k_iir<8>, dim3(grid_for(channels)), dim3(kB), 0, c->stream, xb.data(), zi, dc->ptr, (int)order, leading, dim_len, channels, yb.data(), fb.data());
            else
                hipLaunchKernelGGL(k_iir<64>, dim3(grid_for(channels)), dim3(kB), 0, c->stream, xb.data(), zi, dc->ptr, (int)order, leading, dim_len, channels, yb.data(), fb.data());
            c->tel.kernel_launches++;
            if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "iir_filter: launch failed");
            RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the coefficient block is released on return
        }
    }
    if (rc != RMHIP_OK) {
        rmhip_free(ctx, *output);
        if (*final_state) rmhip_free(ctx, *final_state);
        *output = *final_state = 0;
    }
    return rc;
}

int rmhip_imfilter(rmhip_ctx* ctx, rmhip_buf image, rmhip_buf kernel, int padding, double constant_value, int shape_mode, int convolution, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (padding < 0 || padding > 3 || shape_mode < 0 || shape_mode > 2) return fail(RMHIP_ERR_INVALID, "imfilter: padding %d / shape %d", padding, shape_mode);
    Buffer ib, kb, ob;
    RMHIP_TRY(c->get(image, &ib));
    RMHIP_TRY(c->get(kernel, &kb));
    if (kb.numel == 0) return fail(RMHIP_ERR_INVALID, "imfilter: filter must be non-empty along every dimension");  // imfilter.rs:482-487
    std::vector<size_t> ishape = ib.shape.empty() ? std::vector<size_t>{1, 1} : ib.shape, kshape = kb.shape.empty() ? std::vector<size_t>{1, 1} : kb.shape;
    const size_t rank = std::max(ishape.size(), kshape.size());
    if (rank > 4) return fail(RMHIP_ERR_UNSUPPORTED, "imfilter: %zu dimensions", rank);
    for (size_t d = 0; d < kshape.size(); ++d) {  // validate_kernel_shape, imfilter.rs:567-593
        if (d >= ishape.size() && kshape[d] > 1)
            return fail(RMHIP_ERR_INVALID, "imfilter: filter dimension %zu is %zu, but the image has no corresponding axis", d + 1, kshape[d]);
        if ((d < ishape.size() ? ishape[d] : 1) == 0) return fail(RMHIP_ERR_INVALID, "imfilter: image must not have zero-length dimensions");
    }
    ImfArgs A{};
    A.rank = (int)rank, A.padding = padding, A.convolution = convolution ? 1 : 0, A.constant = constant_value;
    std::vector<size_t> oshape(rank);
    u64 is = 1, ks = 1;
    for (size_t d = 0; d < 4; ++d) {
        const long long img = d < ishape.size() ? (long long)ishape[d] : 1, ker = d < kshape.size() ? (long long)kshape[d] : 1;
        A.img[d] = img, A.ker[d] = ker, A.origin[d] = ker / 2;
        A.istride[d] = is, A.kstride[d] = ks;
        is *= (u64)img, ks *= (u64)ker;
        long long o = img, base = 0;                                              // same
        if (shape_mode == 1) o = img + ker - 1, base = A.origin[d] - (ker - 1);   // full
        if (shape_mode == 2) o = img >= ker ? img - ker + 1 : 0, base = A.origin[d];  // valid
        A.out[d] = o, A.base[d] = base;
        if (d < rank) oshape[d] = (size_t)o;
    }
    while (oshape.size() > ishape.size() && oshape.back() == 1) oshape.pop_back();  // imfilter.rs:526-532
    if (oshape.empty()) oshape.push_back(1);
    RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), out, &ob));
    A.total = ob.numel;
    if (ob.numel == 0) return RMHIP_OK;
    if (ob.numel > 0x7fffffffull * kB) {  // (the output is registered by now: released here, the caller never learns its id)
        const size_t too_many = ob.numel;
        (void)rmhip_free(ctx, *out);
        *out = 0;
        return fail(RMHIP_ERR_UNSUPPORTED, "imfilter: %zu outputs", too_many);
    }
    for (int d = 0; d < 4; ++d)
        if (A.out[d] == 0) A.out[d] = 1;  // (never reached with numel > 0)
    const u64 planes = (u64)A.out[2] * (u64)A.out[3];  // with a two-dimensional filter every further image dimension is a batch of planes
    const size_t patch_bytes = (size_t)(IMF_TX + A.ker[0] - 1) * (size_t)(IMF_TY + A.ker[1] - 1) * sizeof(double);
    const u64 tiles_y = ((u64)A.out[1] + IMF_TY - 1) / IMF_TY;
    if (A.ker[2] == 1 && A.ker[3] == 1 && A.ker[0] <= 64 && patch_bytes <= 48u * 1024 && tiles_y <= 65535 && planes <= 65535) {
        const dim3 grid((unsigned)(((u64)A.out[0] + IMF_TX - 1) / IMF_TX), (unsigned)tiles_y, (unsigned)planes);
        hipLaunchKernelGGL(k_imfilter_tile, grid, dim3(kB), patch_bytes, c->stream, ib.data(), kb.data(), A, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    const int in_lds = kb.numel <= LDS_TAPS;
    hipLaunchKernelGGL(k_imfilter, dim3(grid_for(ob.numel)), dim3(kB), in_lds ? kb.numel * sizeof(double) : 0, c->stream, ib.data(), kb.data(), A, in_lds, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_interp1(rmhip_ctx* ctx, rmhip_buf x, rmhip_buf y, rmhip_buf xq, size_t sample_len, size_t series_count, size_t query_len, const size_t* output_shape,
                  size_t out_rank, int nearest, int extrapolation, double extrapolation_value, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (out_rank && !output_shape)) return fail(RMHIP_ERR_INVALID, "interp1: null argument");
    if (extrapolation < 0 || extrapolation > 2) return fail(RMHIP_ERR_INVALID, "interp1: extrapolation %d", extrapolation);
    if (sample_len < 2) return fail(RMHIP_ERR_INVALID, "interp1: sample_len must be at least 2");
    if (series_count < 1) return fail(RMHIP_ERR_INVALID, "interp1: series_count must be positive");
    Buffer xb, yb, qb;
    RMHIP_TRY(c->get(x, &xb));
    RMHIP_TRY(c->get(y, &yb));
    RMHIP_TRY(c->get(xq, &qb));
    u64 olen = 1;
    for (size_t d = 0; d < out_rank; ++d) olen *= output_shape[d];
    if (olen != (u64)query_len * series_count) return fail(RMHIP_ERR_SHAPE, "interp1: output shape does not match query/series count");
    if (xb.numel != sample_len) return fail(RMHIP_ERR_SHAPE, "interp1: X length does not match sample_len");
    if (yb.numel != sample_len * series_count) return fail(RMHIP_ERR_SHAPE, "interp1: Y length does not match sample/series count");
    if (qb.numel != query_len) return fail(RMHIP_ERR_SHAPE, "interp1: Xq length does not match query_len");
    Buffer ob;
    RMHIP_TRY(c->new_buffer(output_shape, out_rank, out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (ob.numel > 0x7fffffffull * kB) {  // (the output is registered by now: released here, the caller never learns its id)
        const size_t too_many = ob.numel;
        (void)rmhip_free(ctx, *out);
        *out = 0;
        return fail(RMHIP_ERR_UNSUPPORTED, "interp1: %zu outputs", too_many);
    }
    hipLaunchKernelGGL(k_interp1, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, xb.data(), yb.data(), qb.data(), (u64)sample_len, (u64)query_len, (u64)ob.numel, nearest ? 1 : 0,
                       extrapolation, extrapolation_value, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_polyval(rmhip_ctx* ctx, rmhip_buf coefficients, rmhip_buf points, int has_mu, double mean, double scale, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer cb, xb;
    RMHIP_TRY(c->get(coefficients, &cb));
    RMHIP_TRY(c->get(points, &xb));
    Buffer ob;
    RMHIP_TRY(c->new_buffer(xb.shape.data(), xb.shape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (ob.numel > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "polyval: %zu points", ob.numel);
    std::shared_ptr<Allocation> flag;
    RMHIP_TRY(c->alloc_device(1, &flag));
    RMHIP_HIP_CHECK(hipMemsetAsync(flag->ptr, 0, sizeof(double), c->stream));
    const int in_lds = cb.numel <= LDS_TAPS;
    hipLaunchKernelGGL(k_polyval, dim3(grid_for(ob.numel)), dim3(kB), in_lds ? cb.numel * sizeof(double) : 0, c->stream, cb.data(), (u64)cb.numel, xb.data(), (u64)ob.numel,
                       has_mu ? 1 : 0, mean, scale, in_lds, ob.data(), (int*)flag->ptr);
    c->tel.kernel_launches++;
    int bad = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&bad, flag->ptr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (bad) {  // the CPU's complex recurrence turns these into NaN + NaN i (a complex result): the caller evaluates on the host
        rmhip_free(ctx, *out);
        *out = 0;
        return fail(RMHIP_ERR_UNSUPPORTED, "polyval: a non-finite intermediate value");
    }
    return RMHIP_OK;
}

namespace {
// poly_orientation_from_shape (simple_provider.rs:320-338): 0 scalar, 1 row, 2 column; more than one extent above 1 is not a vector
int poly_orientation(const std::vector<size_t>& shape, int* orientation) {
    int non_unit = 0;
    *orientation = 0;
    for (size_t d = 0; d < shape.size(); ++d)
        if (shape[d] > 1) ++non_unit, *orientation = d == 0 ? 2 : 1;
    return non_unit > 1 ? rmhip::fail(RMHIP_ERR_INVALID, "polyder: coefficient inputs must be vectors") : RMHIP_OK;
}

// allocate_polynomial / poly_shape_for_len (simple_provider.rs:341-348, 763-770) of work[first .. len), or of [0] when nothing is left
int poly_emit(rmhip::Context* c, const double* work, rmhip::u64 first, rmhip::u64 len, int orientation, rmhip_buf* out) {
    const rmhip::u64 kept = first < len ? len - first : 1;
    const size_t shape[2] = {orientation == 2 && kept > 1 ? (size_t)kept : 1, orientation == 2 || kept <= 1 ? 1 : (size_t)kept};
    rmhip::Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (first < len) RMHIP_HIP_CHECK(hipMemcpyAsync(ob.data(), work + first, kept * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    else RMHIP_HIP_CHECK(hipMemsetAsync(ob.data(), 0, sizeof(double), c->stream));
    return RMHIP_OK;
}

// one untrimmed result of k_poly, trimmed and emitted
int poly_run(rmhip::Context* c, const rmhip::PolyVec& p, const rmhip::PolyVec& q, int mode, rmhip::u64 len, int orientation, rmhip_buf* out) {
    using namespace rmhip;
    if (len > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "polyder: %llu coefficients", (unsigned long long)len);
    std::shared_ptr<Allocation> work;
    RMHIP_TRY(c->alloc_device(len + 1, &work));
    unsigned long long* first_dev = (unsigned long long*)(work->ptr + len);
    hipLaunchKernelGGL(k_poly, dim3(grid_for(len)), dim3(kB), 0, c->stream, p, q, mode, 0.0, len, work->ptr);
    hipLaunchKernelGGL(k_poly_first, dim3(1), dim3(kB), 0, c->stream, work->ptr, len, first_dev);
    c->tel.kernel_launches += 2;
    RMHIP_HIP_CHECK(hipGetLastError());
    unsigned long long first = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&first, first_dev, sizeof(first), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the result's length is part of the answer
    return poly_emit(c, work->ptr, first, len, orientation, out);
}
}  // namespace

int rmhip_polyder(rmhip_ctx* ctx, rmhip_buf p, rmhip_buf q_or_0, int quotient, rmhip_buf* out, rmhip_buf* denominator_or_null) {
    CTX_OR_FAIL(ctx);
    if (!out || (quotient && (!q_or_0 || !denominator_or_null))) return fail(RMHIP_ERR_INVALID, "polyder: null argument");
    Buffer pb, qb;
    RMHIP_TRY(c->get(p, &pb));
    int op = 0, oq = 0;
    RMHIP_TRY(poly_orientation(pb.shape, &op));
    const PolyVec pv{pb.numel ? pb.data() : nullptr, (u64)pb.numel};
    if (!q_or_0) return poly_run(c, pv, pv, 0, pv.n <= 1 ? 1 : pv.n - 1, op, out);
    RMHIP_TRY(c->get(q_or_0, &qb));
    RMHIP_TRY(poly_orientation(qb.shape, &oq));
    const PolyVec qv{qb.numel ? qb.data() : nullptr, (u64)qb.numel};
    const u64 lp = pv.n ? pv.n : 1, lq = qv.n ? qv.n : 1, dp = pv.n <= 1 ? 1 : pv.n - 1, dq = qv.n <= 1 ? 1 : qv.n - 1;
    const u64 len = std::max(dp + lq - 1, lp + dq - 1);
    RMHIP_TRY(poly_run(c, pv, qv, quotient ? 2 : 1, len, op, out));
    if (!quotient) return RMHIP_OK;
    const int rc = poly_run(c, pv, qv, 3, 2 * lq - 1, oq, denominator_or_null);
    if (rc != RMHIP_OK) {
        rmhip_free(ctx, *out);
        *out = 0;
    }
    return rc;
}

int rmhip_polyint(rmhip_ctx* ctx, rmhip_buf p, double constant, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer pb;
    RMHIP_TRY(c->get(p, &pb));
    int orientation = 0;
    RMHIP_TRY(poly_orientation(pb.shape, &orientation));
    const u64 len = (u64)pb.numel + 1;
    if (len > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "polyint: %zu coefficients", pb.numel);
    const size_t shape[2] = {orientation == 2 && len > 1 ? (size_t)len : 1, orientation == 2 || len <= 1 ? 1 : (size_t)len};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    const PolyVec pv{pb.numel ? pb.data() : nullptr, (u64)pb.numel};
    hipLaunchKernelGGL(k_poly, dim3(grid_for(len)), dim3(kB), 0, c->stream, pv, pv, 4, constant, len, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_window(rmhip_ctx* ctx, int kind, size_t len, int periodic, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (kind < 0 || kind > 2) return fail(RMHIP_ERR_INVALID, "window: kind %d", kind);
    const size_t shape[2] = {len, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (len == 0) return RMHIP_OK;
    if (len == 1) return launch_fill(c, ob.data(), 1, 1.0);
    const double denom = (double)((periodic ? len + 1 : len) - 1);
    hipLaunchKernelGGL(k_window, dim3(grid_for(len)), dim3(kB), 0, c->stream, kind, (u64)len, denom, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

}  // extern "C"
