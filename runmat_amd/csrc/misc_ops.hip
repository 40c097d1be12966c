// misc_ops.hip -- small construction / linear-algebra hooks of array and linalg builtins: one thread per output element, one or two
// rounded operations each (bit-exact against the oracle), HBM-bound.
//   diag_from_vector(_sized)          crates/runmat-accelerate-api/src/lib.rs:1600-1623   (simple_provider.rs:3222-3281)
//   kron                              lib.rs:2697-2699    (builtins/array/shape/kron.rs:358-485)
//   cross                             lib.rs:2701-2708    (builtins/math/linalg/ops/cross.rs:332-364, 443-467)
//   gradient_dim(_with_coordinates)   lib.rs:2604-2620    (builtins/math/reduction/gradient.rs:650-720, 814-833)
//   issymmetric                       lib.rs:3115-3124    (builtins/math/linalg/structure/issymmetric.rs:461-487, 517-526)
#include <algorithm>
#include <cmath>
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
constexpr int kB = 256;

inline unsigned grid_for(u64 n) { return (unsigned)((n + kB - 1) / kB); }

__global__ void __launch_bounds__(kB) k_diag_from_vector(const double* __restrict__ v, u64 len, long long offset, u64 rows, u64 cols, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= rows * cols) return;
    const u64 row = o % rows, col = o / rows;
    // (row, col) holds element idx when row == idx + max(-offset, 0) and col == idx + max(offset, 0)
    double val = 0.0;
    const long long idx = offset >= 0 ? (long long)row : (long long)col;
    if ((long long)col - (long long)row == offset && idx >= 0 && (u64)idx < len) val = v[idx];
    __builtin_nontemporal_store(val, out + o);
}

struct KronDims {
    int rank;
    u64 sa[8], sb[8];  // padded extents; strides of the operands follow from them
};
// I = unsigned when the output has fewer than 2^32 elements: a 64-bit division is ~20 times the instructions of a 32-bit one, and there
// are four per dimension and element (8192^2 doubles: 0.24 ms with 64-bit indices)
template <class I>
__global__ void __launch_bounds__(kB) k_kron(const double* __restrict__ a, const double* __restrict__ b, KronDims d, u64 total, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    I rem = (I)o, ia = 0, ib = 0, stra = 1, strb = 1;
    for (int k = 0; k < d.rank; ++k) {
        const I sa = (I)d.sa[k], sb = (I)d.sb[k], ext = sa * sb;
        const I q = rem / ext, cd = rem - q * ext;
        rem = q;
        const I ca = cd / sb, cb = cd - ca * sb;
        ia += ca * stra;
        ib += cb * strb;
        stra *= sa;
        strb *= sb;
    }
    __builtin_nontemporal_store(a[ia] * b[ib], out + o);
}

__global__ void __launch_bounds__(kB) k_cross(const double* __restrict__ a, const double* __restrict__ b, u64 pre, u64 post, double* __restrict__ out) {
    const u64 t = (u64)blockIdx.x * kB + threadIdx.x;
    if (t >= pre * post) return;
    const u64 before = t % pre, after = t / pre;
    const u64 i1 = after * pre * 3 + before, i2 = i1 + pre, i3 = i2 + pre;
    const double a1 = a[i1], a2 = a[i2], a3 = a[i3], b1 = b[i1], b2 = b[i2], b3 = b[i3];
    out[i1] = a2 * b3 - a3 * b2;
    out[i2] = a3 * b1 - a1 * b3;
    out[i3] = a1 * b2 - a2 * b1;
}

__global__ void __launch_bounds__(kB) k_gradient(const double* __restrict__ x, u64 pre, u64 len, u64 total, double spacing, const double* __restrict__ coords,
                                                 double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    const u64 k = (o / pre) % len;
    double num, den;
    if (k == 0) {
        num = x[o + pre] - x[o];
        den = coords ? coords[1] - coords[0] : spacing;
    } else if (k + 1 == len) {
        num = x[o] - x[o - pre];
        den = coords ? coords[len - 1] - coords[len - 2] : spacing;
    } else {
        num = x[o + pre] - x[o - pre];
        den = coords ? coords[k + 1] - coords[k - 1] : 2.0 * spacing;
    }
    __builtin_nontemporal_store(num / den, out + o);
}

// One workgroup per pair of 32 x 32 tiles (ti <= tj): tile (ti, tj) and its mirror (tj, ti) are both read along their columns
// (coalesced) and meet in LDS - the element-per-thread form read the mirror with a stride of n doubles (0.44 ms at 8192^2).  Any pair
// that fails `v == r || (finite && |v - r| <= tol)` raises the flag; the diagonal is checked against zero for the skew kind.
// `skew` bit 0: the skew kind; bit 1: ishermitian's rule for real data (same test, plus "a NaN on the diagonal fails").
constexpr int SYM_T = 32;
__global__ void __launch_bounds__(256) k_issymmetric(const double* __restrict__ a, u64 n, u64 tiles, int skew, double tol, int* __restrict__ bad) {
    __shared__ double up[SYM_T][SYM_T + 1], lo[SYM_T][SYM_T + 1];
    // blockIdx.x enumerates the pairs (ti <= tj) column by column of the tile grid
    const u64 p = blockIdx.x;
    u64 tj = (u64)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);  // tile column tj holds pairs tj (tj + 1) / 2 ... + tj
    while (tj * (tj + 1) / 2 > p) --tj;
    while ((tj + 1) * (tj + 2) / 2 <= p) ++tj;
    const u64 ti = p - tj * (tj + 1) / 2;
    (void)tiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads, four rows of the tile each
    for (int q = 0; q < 4; ++q) {
        const int c = ty + 8 * q;
        const u64 ru = ti * SYM_T + tx, cu = tj * SYM_T + c;  // upper tile element (ru, cu)
        up[c][tx] = (ru < n && cu < n) ? a[ru + cu * n] : 0.0;
        const u64 rl = tj * SYM_T + tx, cl = ti * SYM_T + c;  // mirror tile element (rl, cl)
        lo[c][tx] = (rl < n && cl < n) ? a[rl + cl * n] : 0.0;
    }
    __syncthreads();
    bool fail = false;
    for (int q = 0; q < 4; ++q) {
        const int c = ty + 8 * q;
        const u64 row = ti * SYM_T + tx, col = tj * SYM_T + c;
        if (row >= n || col >= n || row > col) continue;
        const double v = up[c][tx];
        if (row == col && !(skew & 1)) {  // the Hermitian kind also refuses a NaN on the diagonal (ishermitian.rs:462-465)
            fail |= (skew & 2) && isnan(v);
            continue;
        }
        const double m = lo[tx][c];  // a(col, row)
        const double r = row == col ? 0.0 : ((skew & 1) ? -m : m);
        bool ok = v == r;
        if (!ok && isfinite(v) && isfinite(r)) ok = fabs(v - r) <= tol;
        fail |= !ok;
    }
    if (fail) *bad = 1;
}

// bandwidth (bandwidth.rs:341-365): the largest row - col (lower) and col - row (upper) over the entries that are non-zero or NaN.  A
// grid-stride sweep in storage order; (row, col) advance with the stride by one add and one wrap instead of a 64-bit division per
// element.  At most one atomicMax pair per workgroup - atomics resolve at the memory side, no fence is needed for a max.
__global__ void __launch_bounds__(kB) k_bandwidth(const double* __restrict__ a, u64 rows, u64 total, unsigned* __restrict__ res) {
    const u64 stride = (u64)gridDim.x * kB;
    u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    u64 r = i % rows, col = i / rows;
    const u64 sr = stride % rows, sc = stride / rows;
    unsigned lo = 0, up = 0;
    constexpr int U = 8;  // eight loads in flight per thread: one at a time left the sweep latency-bound (0.45 ms at 8192^2)
    while (i < total) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = i + u * stride < total ? __builtin_nontemporal_load(a + i + u * stride) : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (v[u] != 0.0) {  // NaN != 0 is true: NaN counts (bandwidth.rs:354)
                if (r >= col) lo = max(lo, (unsigned)(r - col));
                else up = max(up, (unsigned)(col - r));
            }
            r += sr;
            col += sc;
            if (r >= rows) {
                r -= rows;
                ++col;
            }
        }
        i += U * stride;
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = max(lo, (unsigned)__shfl_xor((int)lo, o));
        up = max(up, (unsigned)__shfl_xor((int)up, o));
    }
    __shared__ unsigned part[2][kB / 64];
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = lo;
        part[1][threadIdx.x >> 6] = up;
    }
    __syncthreads();
    if (threadIdx.x < 2) {  // thread 0: lower, thread 1: upper
        unsigned m = 0;
        for (int w = 0; w < kB / 64; ++w) m = max(m, part[threadIdx.x][w]);
        // same-address atomics serialise (one per wave cost 0.33 ms at 8192^2): one per workgroup, and only when it can still raise
        // the running maximum (a stale read only lets a redundant atomic through - the maximum is monotonic)
        if (m > __hip_atomic_load(res + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(res + threadIdx.x, m);
    }
}

// peaks (builtins/array/creation/peaks.rs:511-550): the test surface, every product and sum in the CPU's order; three exponentials per point
// (the device's, within an ulp of the host libm's: tests state the bound).  GRID: x = -3 + 6 col / (n - 1), y likewise from the row.
__device__ __forceinline__ double peaks_at(double x, double y) {
    const double x2 = x * x, y2 = y * y, x3 = x2 * x, y5 = (y2 * y2) * y;
    const double a = 1.0 - x, yp = y + 1.0, xp = x + 1.0;
    const double t1 = 3.0 * (a * a) * exp(-x2 - yp * yp);
    const double t2 = 10.0 * (x / 5.0 - x3 - y5) * exp(-x2 - y2);
    const double t3 = 1.0 / 3.0 * exp(-(xp * xp) - y2);
    return t1 - t2 - t3;
}

template <bool GRID>
__global__ void __launch_bounds__(kB) k_peaks(const double* __restrict__ xs, const double* __restrict__ ys, u64 n, u64 total, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    double x, y;
    if (GRID) {
        const u64 row = o % n, col = o / n;
        x = n == 1 ? 3.0 : -3.0 + 6.0 * (double)col / (double)(n - 1);
        y = n == 1 ? 3.0 : -3.0 + 6.0 * (double)row / (double)(n - 1);
    } else {
        x = xs[o], y = ys[o];
    }
    out[o] = peaks_at(x, y);
}

// covariance_to_correlation (runmat-accelerate/src/simple_provider.rs:885-975): validate a covariance matrix the way the CPU does - finite or NaN
// entries, non-negative diagonal, symmetric to 1e-10 relative, |cov| within sqrt(var_i var_j) - and scale it by the standard deviations.
// flags[0]: a non-finite entry; flags[1]: a diagonal entry that is not >= 0; flags[2] / flags[3]: the smallest (col, row) order index of a pair
// that is not symmetric / exceeds the variance bound (the CPU stops at the first pair in that order, symmetry tested first).
__global__ void __launch_bounds__(kB) k_cov_validate(const double* __restrict__ m, u64 n, unsigned long long* __restrict__ flags) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= n * n) return;
    const u64 row = o % n, col = o / n;
    const double a = m[o];
    if (!isnan(a) && !isfinite(a)) flags[0] = 1;
    if (row == col && !(a >= 0.0)) flags[1] = 1;
    if (row < col) {
        const double b = m[col + row * n];
        if (isnan(a) && isnan(b)) return;
        bool sym_bad = isnan(a) || isnan(b);
        if (!sym_bad) {
            const double tol = 1.0e-10 * fmax(fmax(fabs(a), fabs(b)), 1.0);
            sym_bad = !(fabs(a - b) <= tol);
        }
        if (sym_bad) {
            atomicMin(flags + 2, (unsigned long long)o);
            return;
        }
        const double vr = m[row + row * n], vc = m[col + col * n];
        if (isnan(vr) || isnan(vc)) return;
        const double mc = sqrt(vr * vc);
        const double bt = 1.0e-10 * fmax(fmax(mc, fabs(a)), 1.0);
        if (!(fabs(a) <= mc + bt)) atomicMin(flags + 3, (unsigned long long)o);
    }
}

__global__ void __launch_bounds__(kB) k_cov_to_corr(const double* __restrict__ m, u64 n, double* __restrict__ corr, double* __restrict__ sigma) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= n * n) return;
    const u64 row = o % n, col = o / n;
    const double sr = sqrt(m[row + row * n]), sc = sqrt(m[col + col * n]);
    const double den = sr * sc;
    corr[o] = den == 0.0 ? NAN : m[o] / den;
    if (col == 0) sigma[row] = sr;
}

// corrcoef from the covariance matrix (corrcoef.rs:720-787, 895-926): r(i, j) = cov(i, j) / (sqrt(var_i) sqrt(var_j)), NaN unless both variances
// are finite and positive, values within 1e-12 outside [-1, 1] pulled onto the bound; the diagonal is exactly 1 where the deviation is positive.
__global__ void __launch_bounds__(kB) k_corr_from_cov(const double* __restrict__ cov, u64 n, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= n * n) return;
    const u64 i = o % n, j = o / n;
    const double vi = cov[i + i * n], vj = cov[j + j * n];
    double r = NAN;
    if (i == j) {
        double v = vi;
        if (v < 0.0 && v > -1.0e-12) v = 0.0;
        if (!isnan(v) && sqrt(v) > 0.0) r = 1.0;
    } else if (isfinite(vi) && isfinite(vj) && vi > 0.0 && vj > 0.0) {
        const double sx = sqrt(vi), sy = sqrt(vj);
        if (sx != 0.0 && sy != 0.0) {
            r = cov[o] / (sx * sy);
            if (r > 1.0 && r - 1.0 < 1.0e-12) r = 1.0;
            else if (r < -1.0 && -1.0 - r < 1.0e-12) r = -1.0;
        }
    }
    out[o] = r;
}

// trapezoid terms: t[k + 1] = 0.5 * w_k * (x[k] + x[k + 1]), t[0] = 0 along the dimension (simple_provider.rs:2534-2563); their running /
// total sums are the library's cumulative-scan and reduction kernels.  KIND: 0 unit, 1 scalar, 3 coordinate vector, 4 spacing tensor.
template <int KIND>
__global__ void __launch_bounds__(kB) k_trapz_terms(const double* __restrict__ x, u64 pre, u64 len, u64 total, double scalar, const double* __restrict__ sp,
                                                    double* __restrict__ t) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    const u64 k1 = (o / pre) % len;
    double v = 0.0;
    if (k1 > 0) {
        const u64 i0 = o - pre;
        double w = 1.0;
        if (KIND == 1) w = scalar;
        if (KIND == 3) w = sp[k1] - sp[k1 - 1];
        if (KIND == 4) w = sp[o] - sp[i0];
        v = 0.5 * w * (x[i0] + x[o]);
    }
    __builtin_nontemporal_store(v, t + o);
}

// norm: one sweep leaves per-workgroup partials - sum |x| (NaN iff the operand holds one), and what the order needs of max |x|, min |x|,
// the count of nonzeros, the sum of squares, sum |x|^p - and a one-workgroup kernel folds them and writes the [1, 1] result: no read-back.
// Two / Fro square the values as they are; when the largest magnitude turns out to lie beyond 2^+-500 the fold asks a second sweep for
// squares of x * 2^-e (the CPU's root_sum_of_squares rescales as it goes, norm.rs:381-411: same protection, another summation order) -
// that sweep is always enqueued and leaves at once when it is not wanted.
struct NormPartial {
    double sum_abs, max_abs, min_abs, nnz, sq_big, sq_mid, sq_small, sump;  // the squares in three magnitude classes (Blue's accumulators)
    int nan;
    int pad;
};
// |x| above 2^500 is squared after scaling by 2^-600, below 2^-500 after scaling by 2^600: no overflow, no total underflow, ONE sweep
#define NORM_BIG 0x1p+500
#define NORM_SMALL 0x1p-500
#define NORM_SBIG 0x1p-600
#define NORM_SSMALL 0x1p+600
__device__ __forceinline__ void norm_fold(NormPartial& m, const NormPartial& o) {
    m.sum_abs += o.sum_abs;
    m.max_abs = o.max_abs > m.max_abs ? o.max_abs : m.max_abs;
    m.min_abs = o.min_abs < m.min_abs ? o.min_abs : m.min_abs;
    m.nnz += o.nnz;
    m.sq_big += o.sq_big;
    m.sq_mid += o.sq_mid;
    m.sq_small += o.sq_small;
    m.sump += o.sump;
    m.nan |= o.nan;
}
__device__ __forceinline__ void norm_clear(NormPartial& a) {
    a.sum_abs = a.nnz = a.sq_big = a.sq_mid = a.sq_small = a.sump = 0.0;
    a.max_abs = 0.0;
    a.min_abs = __longlong_as_double(0x7ff0000000000000ll);
    a.nan = 0;
}
// ORDER (compile time): only the accumulators that order needs run per element - fp64 compares and selects are the cost of this sweep, not
// HBM (all seven accumulators in one loop: 33 instructions per element, 2.8 TB/s).  sum |x| is always kept: a NaN anywhere makes it NaN.
__device__ __forceinline__ NormPartial norm_shfl_down(const NormPartial& a, int d) {
    NormPartial o;
    o.sum_abs = __shfl_down(a.sum_abs, d);
    o.max_abs = __shfl_down(a.max_abs, d);
    o.min_abs = __shfl_down(a.min_abs, d);
    o.nnz = __shfl_down(a.nnz, d);
    o.sq_big = __shfl_down(a.sq_big, d);
    o.sq_mid = __shfl_down(a.sq_mid, d);
    o.sq_small = __shfl_down(a.sq_small, d);
    o.sump = __shfl_down(a.sump, d);
    o.nan = __shfl_down(a.nan, d);
    o.pad = 0;
    return o;
}
// a workgroup's partial in thread 0: shuffles inside the waves, sixteen records through LDS
__device__ __forceinline__ NormPartial norm_block_fold(NormPartial a, NormPartial* sh) {
    for (int d = 32; d > 0; d >>= 1) {
        const NormPartial o = norm_shfl_down(a, d);
        norm_fold(a, o);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wv] = a;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) norm_fold(a, sh[w]);
    __syncthreads();
    return a;
}
constexpr int NORM_THREADS = 1024;  // (the reductions' measurement: 1024-thread workgroups stream 8-20 % faster than 256)
template <int ORDER>
__global__ void __launch_bounds__(NORM_THREADS) k_norm_sweep(const double* __restrict__ x, u64 n, double p, NormPartial* __restrict__ parts,
                                                    const double* __restrict__ rescale_io) {
    __shared__ NormPartial sh[NORM_THREADS / 64];
    // ORDER 9 = the second sweep of a Two / Fro norm whose largest magnitude lies outside 2^+-500: squares of x * 2^-e.  It is always
    // launched and leaves at once when the first sweep (ORDER 2: plain squares) did not ask for it.
    const double rescale = ORDER == 9 ? *rescale_io : 1.0;
    if (ORDER == 9 && rescale == 0.0) return;
    // four independent accumulator sets, one per load of a trip (a single set chains eight dependent adds per trip: 3.2 TB/s)
    double sum_abs4[4] = {0.0, 0.0, 0.0, 0.0}, max_abs4[4] = {0.0, 0.0, 0.0, 0.0}, sq4[4] = {0.0, 0.0, 0.0, 0.0}, nnz4[4] = {0.0, 0.0, 0.0, 0.0},
           sump4[4] = {0.0, 0.0, 0.0, 0.0};
    const double inf_ = __longlong_as_double(0x7ff0000000000000ll);
    double min_abs4[4] = {inf_, inf_, inf_, inf_};
#define NORM_TAKE(raw, S)                                                      \
    {                                                                          \
        const double v_ = fabs(raw);                                           \
        sum_abs4[S] += v_;                                                     \
        if (ORDER == 2 || ORDER == 3) max_abs4[S] = fmax(v_, max_abs4[S]);     \
        if (ORDER == 4) min_abs4[S] = fmin(v_, min_abs4[S]);                   \
        if (ORDER == 5) nnz4[S] += v_ != 0.0 ? 1.0 : 0.0;                      \
        if (ORDER == 2) sq4[S] += v_ * v_;                                     \
        if (ORDER == 9) {                                                      \
            const double sv_ = v_ * rescale;                                   \
            sq4[S] += sv_ * sv_;                                               \
        }                                                                      \
        if (ORDER == 8) sump4[S] += pow(v_, p);                                \
    }
    // four 16-byte loads in flight per thread and trip (an aligned body), the ragged ends element by element
    typedef double v2d __attribute__((ext_vector_type(2)));
    const u64 head = (((uintptr_t)x & 15) && n) ? 1 : 0;  // x is 8-byte aligned: at most one element before the 16-byte boundary
    const u64 pairs = (n - head) / 2;
    const v2d* xp = reinterpret_cast<const v2d*>(x + head);
    // every workgroup owns ONE contiguous run of the operand
    const u64 per = (pairs + gridDim.x - 1) / gridDim.x;
    const u64 p0 = (u64)blockIdx.x * per, p1 = (p0 + per < pairs) ? p0 + per : pairs;
    u64 i = p0 + threadIdx.x;
    for (; i + 3 * NORM_THREADS < p1; i += 4 * NORM_THREADS) {
        const v2d q0 = __builtin_nontemporal_load(xp + i), q1 = __builtin_nontemporal_load(xp + i + NORM_THREADS);
        const v2d q2 = __builtin_nontemporal_load(xp + i + 2 * NORM_THREADS), q3 = __builtin_nontemporal_load(xp + i + 3 * NORM_THREADS);
        NORM_TAKE(q0.x, 0) NORM_TAKE(q1.x, 1) NORM_TAKE(q2.x, 2) NORM_TAKE(q3.x, 3) NORM_TAKE(q0.y, 0) NORM_TAKE(q1.y, 1) NORM_TAKE(q2.y, 2) NORM_TAKE(q3.y, 3)
    }
    for (; i < p1; i += NORM_THREADS) {
        const v2d q = __builtin_nontemporal_load(xp + i);
        NORM_TAKE(q.x, 0) NORM_TAKE(q.y, 1)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) NORM_TAKE(x[0], 0)
        if (head + 2 * pairs < n) NORM_TAKE(x[n - 1], 1)
    }
    const double sum_abs = (sum_abs4[0] + sum_abs4[1]) + (sum_abs4[2] + sum_abs4[3]);
    const double max_abs = fmax(fmax(max_abs4[0], max_abs4[1]), fmax(max_abs4[2], max_abs4[3]));
    const double min_abs = fmin(fmin(min_abs4[0], min_abs4[1]), fmin(min_abs4[2], min_abs4[3]));
    const double nnz = (nnz4[0] + nnz4[1]) + (nnz4[2] + nnz4[3]);
    const double sq_mid = (sq4[0] + sq4[1]) + (sq4[2] + sq4[3]), sq_big = 0.0, sq_small = 0.0;
    const double sump = (sump4[0] + sump4[1]) + (sump4[2] + sump4[3]);
#undef NORM_TAKE
    NormPartial a;
    a.sum_abs = sum_abs;
    a.max_abs = max_abs;
    a.min_abs = min_abs;
    a.nnz = nnz;
    a.sq_big = sq_big;
    a.sq_mid = sq_mid;
    a.sq_small = sq_small;
    a.sump = sump;
    a.nan = sum_abs != sum_abs ? 1 : 0;
    a.pad = 0;
    a = norm_block_fold(a, sh);
    if (threadIdx.x == 0) parts[blockIdx.x] = a;
}

// the partials of a sweep -> the norm.  A kernel of its own: folding them in the sweep's last workgroup needs a device-scope fence per
// workgroup, and on this part (eight XCDs, L2s that are not coherent with each other) each of those writes the XCD's dirty lines back -
// 77 us for an 80 MB vector against 18 us for the two-kernel reductions.
template <int ORDER>
__global__ void __launch_bounds__(NORM_THREADS) k_norm_final(const NormPartial* __restrict__ parts, unsigned nparts, double p, double* __restrict__ rescale_io,
                                                             double* __restrict__ out) {
    __shared__ NormPartial sh[NORM_THREADS / 64];
    const double rescale = ORDER == 9 ? *rescale_io : 1.0;
    if (ORDER == 9 && rescale == 0.0) return;
    NormPartial t;
    norm_clear(t);
    for (unsigned b = threadIdx.x; b < nparts; b += NORM_THREADS) norm_fold(t, parts[b]);
    t = norm_block_fold(t, sh);
    if (threadIdx.x == 0) {
        const NormPartial r = t;
        double v;
        if (r.nan) v = __longlong_as_double(0x7ff8000000000000ll);
        else if (ORDER == 1) v = r.sum_abs;
        else if (ORDER == 3) v = r.max_abs;
        else if (ORDER == 4) v = isinf(r.min_abs) ? 0.0 : r.min_abs;  // norm.rs:341-345
        else if (ORDER == 5) v = r.nnz;
        else if (ORDER == 8) v = pow(r.sump, 1.0 / p);
        else if (ORDER == 9) v = sqrt(r.sq_mid) / rescale;
        else if (isinf(r.max_abs) || r.max_abs == 0.0) v = r.max_abs;
        else if (r.max_abs > NORM_BIG || r.max_abs < NORM_SMALL) {
            // the plain squares overflowed or vanished: ask the second sweep for x * 2^-e, e the exponent of the largest magnitude
            const int e = (int)((unsigned)__double2hiint(r.max_abs) >> 20) - 1022;
            *rescale_io = ldexp(1.0, -e);
            v = 0.0;
        } else v = sqrt(r.sq_mid);
        *out = v;
    }
}

std::vector<size_t> matrix_shape(const std::vector<size_t>& s) {
    if (s.empty()) return {1, 1};
    if (s.size() == 1) return {s[0], 1};
    return s;
}

// ensure_diag_shape + is_vector_like (simple_provider.rs:2394-2412)
int vector_operand(const Buffer& b, const char* what) {
    const std::vector<size_t>& s = b.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] != 1) return fail(RMHIP_ERR_UNSUPPORTED, "%s: input must be 2-D", what);
    const size_t rows = s.empty() ? 1 : s[0], cols = s.size() < 2 ? 1 : s[1];
    if (!(rows == 1 || cols == 1 || s.size() <= 1)) return fail(RMHIP_ERR_UNSUPPORTED, "%s: input must be a vector", what);
    return RMHIP_OK;
}

}  // namespace
}  // namespace rmhip

int rmhip_diag_from_vector(rmhip_ctx* ctx, rmhip_buf vector, long long offset, long long rows_or_neg, long long cols_or_neg, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer vb, ob;
    RMHIP_TRY(c->get(vector, &vb));
    RMHIP_TRY(vector_operand(vb, "diag"));
    size_t rows, cols;
    if (rows_or_neg < 0 || cols_or_neg < 0) {  // diag_from_vector: square, len + |offset| (simple_provider.rs:2346-2355)
        const unsigned long long shift = offset < 0 ? (unsigned long long)(-offset) : (unsigned long long)offset;
        rows = cols = vb.numel + (size_t)shift;
    } else {
        rows = (size_t)rows_or_neg;
        cols = (size_t)cols_or_neg;
    }
    if (rows && cols > (size_t)-1 / 8 / rows) return fail(RMHIP_ERR_INVALID, "diag: result size exceeds limits");
    const size_t shape[2] = {rows, cols};
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (ob.numel) {
        hipLaunchKernelGGL(k_diag_from_vector, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, vb.data(), (u64)vb.numel, offset, (u64)rows, (u64)cols, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_kron(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb, ob;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    const size_t rank = std::max<size_t>(std::max(ab.shape.size(), bb.shape.size()), 1);
    if (rank > 8) return fail(RMHIP_ERR_UNSUPPORTED, "kron: more than 8 dimensions");
    KronDims d;
    d.rank = (int)rank;
    std::vector<size_t> oshape(rank);
    for (size_t k = 0; k < rank; ++k) {
        d.sa[k] = k < ab.shape.size() ? ab.shape[k] : 1;
        d.sb[k] = k < bb.shape.size() ? bb.shape[k] : 1;
        oshape[k] = (size_t)(d.sa[k] * d.sb[k]);
    }
    RMHIP_TRY(c->new_buffer(oshape.data(), rank, out, &ob));
    if (ob.numel) {
        if (ob.numel < (1ull << 32)) hipLaunchKernelGGL(k_kron<unsigned>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), bb.data(), d, (u64)ob.numel, ob.data());
        else hipLaunchKernelGGL(k_kron<u64>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), bb.data(), d, (u64)ob.numel, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_cross(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int dim_one_based_or_0, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb, ob;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    const std::vector<size_t> shape = matrix_shape(ab.shape);
    if (ab.numel != bb.numel || shape != matrix_shape(bb.shape)) return fail(RMHIP_ERR_SHAPE, "cross: inputs must be the same size");
    size_t dim = 0;  // zero-based
    if (dim_one_based_or_0 > 0) {
        if ((size_t)dim_one_based_or_0 > shape.size())
            return fail(RMHIP_ERR_INVALID, "cross: dimension %d exceeds the number of array dimensions (%zu)", dim_one_based_or_0, shape.size());
        dim = (size_t)dim_one_based_or_0 - 1;
        if (shape[dim] != 3) return fail(RMHIP_ERR_INVALID, "cross: dimension %d must have length 3", dim_one_based_or_0);
    } else if (dim_one_based_or_0 == 0) {
        while (dim < shape.size() && shape[dim] != 3) ++dim;
        if (dim == shape.size()) return fail(RMHIP_ERR_INVALID, "cross: inputs must have a dimension of length 3");
    } else {
        return fail(RMHIP_ERR_INVALID, "cross: dimension must be >= 1");
    }
    u64 pre = 1, post = 1;
    for (size_t k = 0; k < dim; ++k) pre *= shape[k];
    for (size_t k = dim + 1; k < shape.size(); ++k) post *= shape[k];
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), out, &ob));
    if (ob.numel) {
        hipLaunchKernelGGL(k_cross, dim3(grid_for(pre * post)), dim3(kB), 0, c->stream, ab.data(), bb.data(), pre, post, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_gradient_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, double spacing, rmhip_buf coordinates_or_0, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "gradient_dim: dim must be >= 0");
    Buffer ab, cb, ob;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = matrix_shape(ab.shape);
    u64 pre = 1, len = 1;
    for (size_t k = 0; k < shape.size() && k < (size_t)dim; ++k) pre *= shape[k];
    if ((size_t)dim < shape.size()) len = shape[dim];
    const double* coords = nullptr;
    if (coordinates_or_0) {
        RMHIP_TRY(c->get(coordinates_or_0, &cb));
        if (cb.numel != len) return fail(RMHIP_ERR_SHAPE, "gradient: coordinate vector length must match the dimension (%zu vs %llu)", cb.numel, len);
        coords = cb.data();
    }
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (len <= 1) return launch_fill(c, ob.data(), ob.numel, 0.0);  // gradient.rs:691-692: nothing to difference
    hipLaunchKernelGGL(k_gradient, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), pre, len, (u64)ob.numel, spacing, coords, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

static int symmetry_test(Context* c, const char* who, rmhip_buf a, int mode, double tolerance, int* result) {
    if (!result) return fail(RMHIP_ERR_INVALID, "null result");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t>& s = ab.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] != 1) return fail(RMHIP_ERR_INVALID, "%s: inputs must be 2-D matrices or vectors", who);
    const size_t rows = s.empty() ? 1 : s[0], cols = s.size() < 2 ? 1 : s[1];
    if (rows != cols) {
        *result = 0;
        return RMHIP_OK;
    }
    if (rows == 0) {
        *result = 1;
        return RMHIP_OK;
    }
    std::shared_ptr<Allocation> flag;
    RMHIP_TRY(c->alloc_device(1, &flag));
    RMHIP_HIP_CHECK(hipMemsetAsync(flag->ptr, 0, sizeof(double), c->stream));
    const u64 tiles = (rows + SYM_T - 1) / SYM_T, pairs = tiles * (tiles + 1) / 2;
    if (pairs > 0x7fffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "%s: %zu rows", who, rows);
    hipLaunchKernelGGL(k_issymmetric, dim3((unsigned)pairs), dim3(256), 0, c->stream, ab.data(), (u64)rows, tiles, mode, tolerance, (int*)flag->ptr);
    c->tel.kernel_launches++;
    int bad = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&bad, flag->ptr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *result = bad ? 0 : 1;
    return RMHIP_OK;
}

int rmhip_issymmetric(rmhip_ctx* ctx, rmhip_buf a, int skew, double tolerance, int* result) {
    CTX_OR_FAIL(ctx);
    return symmetry_test(c, "issymmetric", a, skew ? 1 : 0, tolerance, result);
}

int rmhip_ishermitian(rmhip_ctx* ctx, rmhip_buf a, int skew, double tolerance, int* result) {
    CTX_OR_FAIL(ctx);
    return symmetry_test(c, "ishermitian", a, (skew ? 1 : 0) | 2, tolerance, result);
}

int rmhip_bandwidth(rmhip_ctx* ctx, rmhip_buf a, unsigned* lower, unsigned* upper) {
    CTX_OR_FAIL(ctx);
    if (!lower || !upper) return fail(RMHIP_ERR_INVALID, "null result");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t>& s = ab.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] > 1) return fail(RMHIP_ERR_INVALID, "bandwidth: invalid input: input must be a 2-D matrix");
    // bandwidth.rs:303-318: a rank-1 shape is a ROW here
    const u64 rows = s.empty() ? 1 : (s.size() == 1 ? 1 : s[0]), cols = s.empty() ? 1 : (s.size() == 1 ? s[0] : s[1]);
    *lower = *upper = 0;
    if (rows == 0 || cols == 0) return RMHIP_OK;
    if (rows > 0xffffffffull || cols > 0xffffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "bandwidth: %zu x %zu", (size_t)rows, (size_t)cols);
    std::shared_ptr<Allocation> res;
    RMHIP_TRY(c->alloc_device(1, &res));
    RMHIP_HIP_CHECK(hipMemsetAsync(res->ptr, 0, sizeof(double), c->stream));
    const u64 total = rows * cols;
    const unsigned grid = (unsigned)std::min<u64>((total + kB - 1) / kB, 2048);
    hipLaunchKernelGGL(k_bandwidth, dim3(grid), dim3(kB), 0, c->stream, ab.data(), rows, total, (unsigned*)res->ptr);
    c->tel.kernel_launches++;
    unsigned host[2] = {0, 0};
    RMHIP_HIP_CHECK(hipMemcpyAsync(host, res->ptr, sizeof(host), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *lower = host[0];
    *upper = host[1];
    return RMHIP_OK;
}

int rmhip_peaks(rmhip_ctx* ctx, size_t n, rmhip_buf x_or_0, rmhip_buf y_or_0, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer xb, yb, ob;
    if (x_or_0 || y_or_0) {  // peaks_xy: same-shape coordinate tensors
        if (!x_or_0 || !y_or_0) return fail(RMHIP_ERR_INVALID, "peaks_xy: both coordinate tensors are required");
        RMHIP_TRY(c->get(x_or_0, &xb));
        RMHIP_TRY(c->get(y_or_0, &yb));
        if (xb.shape != yb.shape) return fail(RMHIP_ERR_SHAPE, "peaks: X and Y must be the same size");
        RMHIP_TRY(c->new_buffer(xb.shape.data(), xb.shape.size(), out, &ob));
        if (ob.numel == 0) return RMHIP_OK;
        hipLaunchKernelGGL(k_peaks<false>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, xb.data(), yb.data(), (u64)0, (u64)ob.numel, ob.data());
    } else {
        const size_t shape[2] = {n, n};
        RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
        if (ob.numel == 0) return RMHIP_OK;
        hipLaunchKernelGGL(k_peaks<true>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, (const double*)nullptr, (const double*)nullptr, (u64)n, (u64)ob.numel, ob.data());
    }
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_covariance_to_correlation(rmhip_ctx* ctx, rmhip_buf matrix, rmhip_buf* correlation, rmhip_buf* sigma) {
    CTX_OR_FAIL(ctx);
    if (!correlation || !sigma) return fail(RMHIP_ERR_INVALID, "null output");
    *correlation = *sigma = 0;
    Buffer mb;
    RMHIP_TRY(c->get(matrix, &mb));
    if (mb.shape.size() > 2) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance matrix must be two-dimensional");
    const u64 rows = mb.shape.empty() ? 1 : mb.shape[0], cols = mb.shape.size() < 2 ? 1 : mb.shape[1];  // simple_provider.rs:885-894
    if (rows != cols) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance matrix must be square");
    const u64 n = rows;
    if (n > 0) {
        std::shared_ptr<Allocation> fl;
        RMHIP_TRY(c->alloc_device(4, &fl));
        unsigned long long init[4] = {0, 0, ~0ull, ~0ull}, host[4];
        RMHIP_HIP_CHECK(hipMemcpyAsync(fl->ptr, init, sizeof init, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_cov_validate, dim3(grid_for(n * n)), dim3(kB), 0, c->stream, mb.data(), n, (unsigned long long*)fl->ptr);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipMemcpyAsync(host, fl->ptr, sizeof host, hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        // the CPU's order of checks (simple_provider.rs:908-952)
        if (host[0]) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance matrix must contain finite values or NaN");
        if (host[1]) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance matrix diagonal entries must be nonnegative");
        if (host[2] != ~0ull && host[2] <= host[3]) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance matrix must be symmetric");
        if (host[3] != ~0ull) return fail(RMHIP_ERR_INVALID, "covariance_to_correlation: covariance magnitude exceeds variance bounds");
    }
    const size_t cshape[2] = {(size_t)n, (size_t)n}, sshape[2] = {(size_t)n, 1};
    Buffer cb, sb;
    RMHIP_TRY(c->new_buffer(cshape, 2, correlation, &cb));
    int rc = c->new_buffer(sshape, 2, sigma, &sb);
    if (rc == RMHIP_OK && n > 0) {
        hipLaunchKernelGGL(k_cov_to_corr, dim3(grid_for(n * n)), dim3(kB), 0, c->stream, mb.data(), n, cb.data(), sb.data());
        c->tel.kernel_launches++;
        if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "covariance_to_correlation: launch failed");
    }
    if (rc != RMHIP_OK) {
        rmhip_free(ctx, *correlation);
        if (*sigma) rmhip_free(ctx, *sigma);
        *correlation = *sigma = 0;
    }
    return rc;
}

int rmhip_corrcoef(rmhip_ctx* ctx, rmhip_buf matrix, int biased, int rows_mode, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (rows_mode != 0) return fail(RMHIP_ERR_UNSUPPORTED, "corrcoef: 'complete' / 'pairwise' row selection uses the CPU path");
    Buffer mb;
    RMHIP_TRY(c->get_raw(matrix, &mb));  // shape only
    if (mb.shape.size() > 2) return fail(RMHIP_ERR_INVALID, "corrcoef: inputs must be 2-D matrices or vectors");
    rmhip_buf cov = 0;
    RMHIP_TRY(rmhip_covariance(ctx, matrix, biased, &cov));  // [cols, cols]; all NaN when the denominator is not positive (corrcoef.rs:726-735)
    Buffer cb, ob;
    int rc = c->get(cov, &cb);
    if (rc == RMHIP_OK) rc = c->new_buffer(cb.shape.data(), cb.shape.size(), out, &ob);
    if (rc == RMHIP_OK && ob.numel) {
        const u64 n = cb.shape.empty() ? 1 : cb.shape[0];
        hipLaunchKernelGGL(k_corr_from_cov, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, cb.data(), n, ob.data());
        c->tel.kernel_launches++;
        if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "corrcoef: launch failed");
    }
    rmhip_free(ctx, cov);
    return rc;
}

int rmhip_trapz_dim(rmhip_ctx* ctx, rmhip_buf a, int dim, int cumulative, int spacing_kind, double scalar, rmhip_buf spacing_or_0, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "trapezoid: dim must be >= 0");
    Buffer ab, sb;
    RMHIP_TRY(c->get(a, &ab));
    std::vector<size_t> shape = matrix_shape(ab.shape);
    while (shape.size() <= (size_t)dim) shape.push_back(1);  // simple_provider.rs:2497-2499
    u64 pre = 1;
    for (int k = 0; k < dim; ++k) pre *= shape[k];
    const u64 len = shape[dim];
    const double* sp = nullptr;
    int kind = spacing_kind;
    if (kind == 2) {  // ScalarHandle: the first element of a resident tensor
        RMHIP_TRY(c->get(spacing_or_0, &sb));
        if (sb.numel == 0) return fail(RMHIP_ERR_INVALID, "trapezoid: scalar spacing is empty");
        RMHIP_HIP_CHECK(hipMemcpyAsync(&scalar, sb.data(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        kind = 1;
    } else if (kind == 3 || kind == 4) {
        RMHIP_TRY(c->get(spacing_or_0, &sb));
        if (kind == 3 && sb.numel < len) return fail(RMHIP_ERR_SHAPE, "trapezoid: spacing vector is shorter than integration dimension");
        if (kind == 4 && sb.numel < ab.numel) return fail(RMHIP_ERR_SHAPE, "trapezoid: spacing tensor is smaller than input");
        sp = sb.data();
    } else if (kind != 0 && kind != 1) {
        return fail(RMHIP_ERR_INVALID, "trapezoid: spacing kind %d", spacing_kind);
    }
    std::vector<size_t> oshape = shape;
    if (!cumulative) oshape[dim] = 1;
    if (ab.numel == 0 || len <= 1) {  // nothing to integrate: zeros of the output's shape (simple_provider.rs:2544-2546, 2576-2577)
        Buffer ob;
        RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), out, &ob));
        return ob.numel ? launch_fill(c, ob.data(), ob.numel, 0.0) : RMHIP_OK;
    }
    rmhip_buf tid = 0;
    Buffer tb;
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), &tid, &tb));
    const unsigned grid = grid_for(tb.numel);
    switch (kind) {
        case 0: hipLaunchKernelGGL(k_trapz_terms<0>, dim3(grid), dim3(kB), 0, c->stream, ab.data(), pre, len, (u64)tb.numel, scalar, sp, tb.data()); break;
        case 1: hipLaunchKernelGGL(k_trapz_terms<1>, dim3(grid), dim3(kB), 0, c->stream, ab.data(), pre, len, (u64)tb.numel, scalar, sp, tb.data()); break;
        case 3: hipLaunchKernelGGL(k_trapz_terms<3>, dim3(grid), dim3(kB), 0, c->stream, ab.data(), pre, len, (u64)tb.numel, scalar, sp, tb.data()); break;
        default: hipLaunchKernelGGL(k_trapz_terms<4>, dim3(grid), dim3(kB), 0, c->stream, ab.data(), pre, len, (u64)tb.numel, scalar, sp, tb.data()); break;
    }
    c->tel.kernel_launches++;
    int rc = hipGetLastError() == hipSuccess ? RMHIP_OK : fail(RMHIP_ERR_HIP, "trapezoid: launch failed");
    if (!rc) rc = cumulative ? rmhip_cumulative(ctx, 0, tid, dim, 0, 0, out) : rmhip_reduce(ctx, RMHIP_RSUM, tid, dim, 0, out);
    rmhip_free(ctx, tid);
    return rc;
}

// order: 1 One, 2 Two, 3 Inf, 4 NegInf, 5 Zero, 6 Fro, 7 Nuc, 8 P(p)  (ProviderNormOrder, lib.rs:745-754)
int rmhip_norm(rmhip_ctx* ctx, rmhip_buf a, int order, double p, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (order < 1 || order > 8) return fail(RMHIP_ERR_INVALID, "norm: order %d", order);
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    // classify_tensor (norm.rs:299-318)
    const std::vector<size_t>& s = ab.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] > 1) return fail(RMHIP_ERR_INVALID, "norm: input must be a vector or 2-D matrix.");
    const size_t rows = s.empty() ? 0 : s[0], cols = s.size() < 2 ? 1 : s[1];
    const bool matrix = !(s.size() <= 1 || rows <= 1 || cols <= 1);
    if (matrix && (order == 2 || order == 7)) {
        // spectral / nuclear norm (norm.rs:531-541, 560-567): the largest / the sum of the singular values, from the one-sided Jacobi
        // decomposition of svdsolve.hip (min(rows, cols) <= 4096, finite data; the CPU uses nalgebra's SVD: parity by tolerance)
        std::vector<double> sv;
        RMHIP_TRY(svd_values_host(c, "norm", ab.data(), rows, cols, &sv));
        double r = 0.0;
        if (order == 2) {
            for (double v : sv) r = v > r ? v : r;
        } else {
            std::sort(sv.begin(), sv.end(), [](double x, double y) { return x > y; });  // (nalgebra hands them over in decreasing order)
            for (double v : sv) r += v;
        }
        const size_t one1[2] = {1, 1};
        Buffer rb;
        RMHIP_TRY(c->new_buffer(one1, 2, out, &rb));
        return launch_fill(c, rb.data(), 1, r);
    }
    if (matrix && (order == 4 || order == 5 || order == 8)) return fail(RMHIP_ERR_INVALID, "norm: order not defined for matrices");  // norm.rs:441-451
    if (!matrix && order == 7) return fail(RMHIP_ERR_INVALID, "norm: nuclear norm is only defined for matrices.");
    if (!matrix && order == 8 && !(std::isfinite(p) && p >= 1.0)) return fail(RMHIP_ERR_INVALID, "norm: vector norm order %g must satisfy p >= 1 (or use 0, Inf, or -Inf).", p);
    const size_t one[2] = {1, 1};
    double value = 0.0;
    const u64 n = ab.numel;
    if (n > 0 && matrix && (order == 1 || order == 3)) {
        // max column / row sum of |a| (norm.rs:497-529): |a|, sum along one dimension, max of the sums - the library's kernels
        // (rmhip_reduce in include mode: a NaN sum makes the maximum NaN, as norm.rs:419-421 wants) - the [1, 1] maximum IS the result
        rmhip_buf absb = 0, sums = 0;
        int rc = rmhip_unary(ctx, RMHIP_ABS, a, &absb);
        if (!rc) rc = rmhip_reduce(ctx, RMHIP_RSUM, absb, order == 1 ? 0 : 1, 0, &sums);
        if (!rc) rc = rmhip_reduce(ctx, RMHIP_RMAX, sums, -1, 0, out);
        if (absb) rmhip_free(ctx, absb);
        if (sums) rmhip_free(ctx, sums);
        return rc;
    } else if (n > 0) {
        // one sweep, the value formed on the device by the last workgroup: no read-back, no synchronisation
        const unsigned grid = (unsigned)std::max<u64>(1, std::min<u64>((n + 8191) / 8192, (u64)c->num_cus * 8));
        std::shared_ptr<Allocation> ws;
        RMHIP_TRY(c->alloc_device((size_t)grid * (sizeof(NormPartial) / 8) + 2, &ws));
        NormPartial* parts = (NormPartial*)ws->ptr;
        double* rescale = (double*)(parts + grid);
        RMHIP_HIP_CHECK(hipMemsetAsync(rescale, 0, sizeof(double), c->stream));
        Buffer ob;
        RMHIP_TRY(c->new_buffer(one, 2, out, &ob));
#define RMHIP_NORM_LAUNCH(O)                                                                                                             \
    hipLaunchKernelGGL(k_norm_sweep<O>, dim3(grid), dim3(NORM_THREADS), 0, c->stream, ab.data(), n, p, parts, (const double*)rescale); \
    hipLaunchKernelGGL(k_norm_final<O>, dim3(1), dim3(NORM_THREADS), 0, c->stream, (const NormPartial*)parts, grid, p, rescale, ob.data()); \
    c->tel.kernel_launches += 2
        switch (order) {
            case 1: RMHIP_NORM_LAUNCH(1); break;
            case 3: RMHIP_NORM_LAUNCH(3); break;
            case 4: RMHIP_NORM_LAUNCH(4); break;
            case 5: RMHIP_NORM_LAUNCH(5); break;
            case 8: RMHIP_NORM_LAUNCH(8); break;
            default:  // Two (vector) and Fro: plain squares, and the rescaled sweep that only runs when they were not enough
                RMHIP_NORM_LAUNCH(2);
                RMHIP_NORM_LAUNCH(9);
                break;
        }
#undef RMHIP_NORM_LAUNCH
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    Buffer ob;
    RMHIP_TRY(c->new_buffer(one, 2, out, &ob));
    const int rc = launch_fill(c, ob.data(), 1, value);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}
