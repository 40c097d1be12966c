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
        if (row >= n || col >= n || row > col || (row == col && !skew)) continue;
        const double v = up[c][tx];
        const double m = lo[tx][c];  // a(col, row)
        const double r = row == col ? 0.0 : (skew ? -m : m);
        bool ok = v == r;
        if (!ok && isfinite(v) && isfinite(r)) ok = fabs(v - r) <= tol;
        fail |= !ok;
    }
    if (fail) *bad = 1;
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

// norm: per-workgroup partials of one sweep - sum |x|, max |x|, min |x|, count of nonzeros, NaN seen, sum (|x| scale)^2, sum |x|^p -
// combined by the last workgroup to finish (ticket), so every vector norm and the Frobenius norm is ONE launch after the sweep that
// found the scale.  The squares are scaled by an exact power of two taken from max |x| (the CPU's root_sum_of_squares rescales as it
// goes, norm.rs:381-411: same protection against overflow / underflow, different summation order).
struct NormPartial {
    double sum_abs, max_abs, min_abs, nnz, sumsq, sump;
    int nan;
    int pad;
};
__global__ void __launch_bounds__(256) k_norm_sweep(const double* __restrict__ x, u64 n, double scale, double p, int want_p, NormPartial* __restrict__ parts,
                                                    unsigned* __restrict__ ticket, NormPartial* __restrict__ out) {
    __shared__ NormPartial sh[256];
    __shared__ int last;
    NormPartial a;
    a.sum_abs = 0.0;
    a.max_abs = 0.0;
    a.min_abs = __longlong_as_double(0x7ff0000000000000ll);
    a.nnz = 0.0;
    a.sumsq = 0.0;
    a.sump = 0.0;
    a.nan = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const double v = fabs(__builtin_nontemporal_load(x + i));
        if (v != v) a.nan = 1;
        a.sum_abs += v;
        a.max_abs = v > a.max_abs ? v : a.max_abs;
        a.min_abs = v < a.min_abs ? v : a.min_abs;
        a.nnz += v != 0.0 ? 1.0 : 0.0;
        const double sv = v * scale;
        a.sumsq += sv * sv;
        if (want_p) a.sump += pow(v, p);
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            NormPartial& m = sh[threadIdx.x];
            const NormPartial& o = sh[threadIdx.x + d];
            m.sum_abs += o.sum_abs;
            m.max_abs = o.max_abs > m.max_abs ? o.max_abs : m.max_abs;
            m.min_abs = o.min_abs < m.min_abs ? o.min_abs : m.min_abs;
            m.nnz += o.nnz;
            m.sumsq += o.sumsq;
            m.sump += o.sump;
            m.nan |= o.nan;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        parts[blockIdx.x] = sh[0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    NormPartial t = sh[0];
    t.sum_abs = t.nnz = t.sumsq = t.sump = 0.0;
    t.max_abs = 0.0;
    t.min_abs = __longlong_as_double(0x7ff0000000000000ll);
    t.nan = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) {
        const NormPartial o = parts[b];
        t.sum_abs += o.sum_abs;
        t.max_abs = o.max_abs > t.max_abs ? o.max_abs : t.max_abs;
        t.min_abs = o.min_abs < t.min_abs ? o.min_abs : t.min_abs;
        t.nnz += o.nnz;
        t.sumsq += o.sumsq;
        t.sump += o.sump;
        t.nan |= o.nan;
    }
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            NormPartial& m = sh[threadIdx.x];
            const NormPartial& o = sh[threadIdx.x + d];
            m.sum_abs += o.sum_abs;
            m.max_abs = o.max_abs > m.max_abs ? o.max_abs : m.max_abs;
            m.min_abs = o.min_abs < m.min_abs ? o.min_abs : m.min_abs;
            m.nnz += o.nnz;
            m.sumsq += o.sumsq;
            m.sump += o.sump;
            m.nan |= o.nan;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *out = sh[0];
        *ticket = 0;
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

int rmhip_issymmetric(rmhip_ctx* ctx, rmhip_buf a, int skew, double tolerance, int* result) {
    CTX_OR_FAIL(ctx);
    if (!result) return fail(RMHIP_ERR_INVALID, "null result");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t>& s = ab.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] != 1) return fail(RMHIP_ERR_INVALID, "issymmetric: inputs must be 2-D matrices or vectors");
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
    if (pairs > 0x7fffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "issymmetric: %zu rows", rows);
    hipLaunchKernelGGL(k_issymmetric, dim3((unsigned)pairs), dim3(256), 0, c->stream, ab.data(), (u64)rows, tiles, skew ? 1 : 0, tolerance, (int*)flag->ptr);
    c->tel.kernel_launches++;
    int bad = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&bad, flag->ptr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    *result = bad ? 0 : 1;
    return RMHIP_OK;
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
    if (matrix && (order == 2 || order == 7)) return fail(RMHIP_ERR_UNSUPPORTED, "norm: the spectral / nuclear norm of a matrix needs its singular values");
    if (matrix && (order == 4 || order == 5 || order == 8)) return fail(RMHIP_ERR_INVALID, "norm: order not defined for matrices");  // norm.rs:441-451
    if (!matrix && order == 7) return fail(RMHIP_ERR_INVALID, "norm: nuclear norm is only defined for matrices.");
    if (!matrix && order == 8 && !(std::isfinite(p) && p >= 1.0)) return fail(RMHIP_ERR_INVALID, "norm: vector norm order %g must satisfy p >= 1 (or use 0, Inf, or -Inf).", p);
    const size_t one[2] = {1, 1};
    double value = 0.0;
    const u64 n = ab.numel;
    if (n > 0 && matrix && (order == 1 || order == 3)) {
        // max column / row sum of |a| (norm.rs:497-529): |a|, sum along one dimension, max of the sums - the library's kernels
        rmhip_buf absb = 0, sums = 0, mx = 0;
        int rc = rmhip_unary(ctx, RMHIP_ABS, a, &absb);
        if (!rc) rc = rmhip_reduce(ctx, RMHIP_RSUM, absb, order == 1 ? 0 : 1, 0, &sums);
        if (!rc) rc = rmhip_reduce(ctx, RMHIP_RMAX, sums, -1, 0, &mx);
        if (!rc) rc = rmhip_read_scalar(ctx, mx, 0, &value);
        if (!rc) {  // any NaN in the matrix is a NaN sum: the norm is NaN (norm.rs:419-421) whatever the maximum made of it
            rmhip_buf tot = 0;
            double t = 0.0;
            rc = rmhip_reduce(ctx, RMHIP_RSUM, sums, -1, 0, &tot);
            if (!rc) rc = rmhip_read_scalar(ctx, tot, 0, &t);
            if (tot) rmhip_free(ctx, tot);
            if (!rc && t != t) value = t;
        }
        if (absb) rmhip_free(ctx, absb);
        if (sums) rmhip_free(ctx, sums);
        if (mx) rmhip_free(ctx, mx);
        if (rc) return rc;
    } else if (n > 0) {
        const unsigned grid = (unsigned)std::min<u64>((n + 1023) / 1024, (u64)c->num_cus * 8);
        std::shared_ptr<Allocation> ws;
        RMHIP_TRY(c->alloc_device((size_t)(grid + 1) * (sizeof(NormPartial) / 8) + 1, &ws));
        NormPartial* parts = (NormPartial*)ws->ptr;
        NormPartial* total = parts + grid;
        unsigned* ticket = (unsigned*)(total + 1);
        RMHIP_HIP_CHECK(hipMemsetAsync(ticket, 0, sizeof(unsigned), c->stream));
        NormPartial h;
        auto sweep = [&](double scale, int want_p) -> int {
            hipLaunchKernelGGL(k_norm_sweep, dim3(grid), dim3(256), 0, c->stream, ab.data(), n, scale, p, want_p, parts, ticket, total);
            c->tel.kernel_launches++;
            RMHIP_HIP_CHECK(hipMemcpyAsync(&h, total, sizeof h, hipMemcpyDeviceToHost, c->stream));
            RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
            return RMHIP_OK;
        };
        RMHIP_TRY(sweep(1.0, order == 8));
        if (h.nan) {
            value = std::numeric_limits<double>::quiet_NaN();
        } else if (order == 1) {
            value = h.sum_abs;
        } else if (order == 3) {
            value = h.max_abs;
        } else if (order == 4) {
            value = std::isinf(h.min_abs) ? 0.0 : h.min_abs;  // norm.rs:341-345
        } else if (order == 5) {
            value = h.nnz;
        } else if (order == 8) {
            value = std::pow(h.sump, 1.0 / p);
        } else {  // Two (vector) / Fro
            if (std::isinf(h.max_abs)) value = h.max_abs;
            else if (h.max_abs == 0.0) value = 0.0;
            else {
                int e = 0;
                (void)std::frexp(h.max_abs, &e);
                const double scale = std::ldexp(1.0, -e);  // max |x| * scale in [0.5, 1): the squares neither overflow nor all underflow
                if (e > 500 || e < -500) {
                    RMHIP_TRY(sweep(scale, 0));
                    value = std::sqrt(h.sumsq) * std::ldexp(1.0, e);
                } else {
                    value = std::sqrt(h.sumsq);  // the first sweep's plain squares
                }
            }
        }
    }
    Buffer ob;
    RMHIP_TRY(c->new_buffer(one, 2, out, &ob));
    const int rc = launch_fill(c, ob.data(), 1, value);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}
