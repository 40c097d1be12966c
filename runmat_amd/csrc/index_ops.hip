// index_ops.hip -- subscript / grid / slice-write hooks and the last per-element unary forms: integer or copy work, bit-exact.
//   ndgrid                    crates/runmat-accelerate-api/src/lib.rs:1567-1569   (simple_provider.rs:2784-2855)
//   sub2ind / ind2sub         lib.rs:3084-3112      (simple_provider.rs:8340-8420, 2268-2291; builtins/array/indexing/ind2sub.rs:289-353)
//   scatter_column / _row     lib.rs:3064-3082      (the callers: runmat-vm/src/indexing/write_slice.rs:680-709)
//   pow2_scale                lib.rs:2325-2331      (simple_provider.rs:5822-5850)
//   round_digits              lib.rs:2197-2204      (simple_provider.rs:5359-5420; decimals mode)
//   unary_real / imag / conj / angle, logical_isreal   lib.rs:2217-2240, 2055   (simple_provider.rs:5482-5640, 4786) on real storage
#include <algorithm>
#include <cmath>
#include <cstring>

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
constexpr int kMaxDims = 16;
inline unsigned grid_for(u64 n) { return (unsigned)((n + kB - 1) / kB); }

__global__ void __launch_bounds__(kB) k_ndgrid(const double* __restrict__ axis, u64 stride, u64 extent, u64 total, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    __builtin_nontemporal_store(axis[(o / stride) % extent], out + o);
}

struct SubArgs {
    int rank;
    const double* in[kMaxDims];
    u64 dim[kMaxDims], stride[kMaxDims];
    unsigned char scalar[kMaxDims];
};
// coerce_sub2ind_value (simple_provider.rs:2268-2291) per subscript; the FIRST failure in the CPU's (element, dimension) order wins
__global__ void __launch_bounds__(kB) k_sub2ind(SubArgs a, u64 len, double* __restrict__ out, u64* __restrict__ first_bad) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= len) return;
    u64 off = 0;
    for (int d = 0; d < a.rank; ++d) {
        const double raw = a.in[d][a.scalar[d] ? 0 : i];
        const double r = round(raw);
        const bool bad = !isfinite(raw) || fabs(r - raw) > 2.220446049250313e-16 || r < 1.0 || r > (double)a.dim[d];
        if (bad) {
            atomicMin(first_bad, i * kMaxDims + (u64)d);
            return;
        }
        off += ((u64)r - 1) * a.stride[d];
    }
    out[i] = (double)(off + 1);
}

struct IndArgs {
    int rank;
    u64 dim[kMaxDims], stride[kMaxDims];
    double* out[kMaxDims];
};
// coerce_linear_index + compute_subscripts (ind2sub.rs:289-353)
__global__ void __launch_bounds__(kB) k_ind2sub(const double* __restrict__ idx, u64 len, u64 total, IndArgs a, u64* __restrict__ first_bad) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= len) return;
    const double raw = idx[i];
    const double r = round(raw);
    if (!isfinite(raw) || fabs(r - raw) > 2.220446049250313e-16 || r < 1.0 || r > (double)total) {
        atomicMin(first_bad, i);
        return;
    }
    const u64 z = (u64)r - 1;
    for (int d = 0; d < a.rank; ++d) a.out[d][i] = (double)((z / a.stride[d]) % a.dim[d] + 1);
}

// out = matrix with one column (COL) or one row replaced by `values`
template <bool COL>
__global__ void __launch_bounds__(kB) k_scatter_line(const double* __restrict__ m, const double* __restrict__ v, u64 rows, u64 total, u64 which, double* __restrict__ out) {
    const u64 o = (u64)blockIdx.x * kB + threadIdx.x;
    if (o >= total) return;
    const u64 r = o % rows, col = o / rows;
    const bool hit = COL ? col == which : r == which;
    __builtin_nontemporal_store(hit ? v[COL ? r : col] : m[o], out + o);
}

// m * 2^e: an integral exponent is the exact power (what exp2 returns for it), anything else goes through exp2 (within an ulp of libm's)
__global__ void __launch_bounds__(kB) k_pow2_scale(const double* __restrict__ m, const double* __restrict__ e, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const double ev = e[i];
    double p;
    if (ev == trunc(ev) && fabs(ev) < 4096.0) {
        const int k = (int)ev;
        if (k > 1023) p = __longlong_as_double(0x7ff0000000000000ll);
        else if (k >= -1022) p = __longlong_as_double((long long)(k + 1023) << 52);
        else if (k >= -1074) p = __longlong_as_double(1ll << (k + 1074));
        else p = 0.0;
    } else {
        p = exp2(ev);
    }
    __builtin_nontemporal_store(m[i] * p, out + i);
}

__global__ void __launch_bounds__(kB) k_round_decimals(const double* __restrict__ x, u64 n, int digits, double factor, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = v;
    if (isfinite(v)) {
        if (digits == 0) r = round(v);
        else if (isfinite(factor) && factor != 0.0) r = round(v * factor) / factor;
    }
    __builtin_nontemporal_store(r, out + i);
}

// KIND 0: angle of a real = atan2(+0, x): +0 for x > 0 and +0, pi for x < 0 and -0, NaN for NaN; 1: zeros (imag of a real)
template <int KIND>
__global__ void __launch_bounds__(kB) k_real_parts(const double* __restrict__ x, u64 n, double* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    double r = 0.0;
    if (KIND == 0) {
        const double v = x[i];
        r = v != v ? v : (signbit(v) ? 0x1.921fb54442d18p+1 : 0.0);
    }
    __builtin_nontemporal_store(r, out + i);
}

// Rust's f64::powi = compiler-rt __powidf2: square-and-multiply, reciprocal at the end for a negative exponent
double powi10(int b) {
    const bool recip = b < 0;
    double a = 10.0, r = 1.0;
    long long e = b;
    if (e < 0) e = -e;
    while (true) {
        if (e & 1) r *= a;
        e /= 2;
        if (e == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

u64 product(const size_t* s, size_t n) {
    u64 p = 1;
    for (size_t i = 0; i < n; ++i) p *= s[i];
    return p;
}

}  // namespace
}  // namespace rmhip

int rmhip_ndgrid(rmhip_ctx* ctx, const rmhip_buf* axes, size_t n_axes, const size_t* output_shape, size_t rank, size_t output_count, rmhip_buf* outputs) {
    CTX_OR_FAIL(ctx);
    if (!axes || !output_shape || !outputs) return fail(RMHIP_ERR_INVALID, "ndgrid: null argument");
    if (output_count == 0) return fail(RMHIP_ERR_INVALID, "ndgrid: missing outputs");
    if (output_count > n_axes) return fail(RMHIP_ERR_INVALID, "ndgrid: too many outputs for axes");
    if (rank == 0) return fail(RMHIP_ERR_INVALID, "ndgrid: missing shape");
    const u64 total = product(output_shape, rank);
    for (size_t d = 0; d < output_count; ++d) outputs[d] = 0;
    int rc = RMHIP_OK;
    u64 stride = 1;
    for (size_t d = 0; d < output_count && rc == RMHIP_OK; ++d) {
        Buffer ab, ob;
        rc = c->get(axes[d], &ab);
        const u64 extent = d < rank ? output_shape[d] : 1;
        if (!rc && ab.numel != extent) rc = fail(RMHIP_ERR_SHAPE, "ndgrid: axis %zu length %zu does not match output extent %llu", d + 1, ab.numel, extent);
        if (!rc) rc = c->new_buffer(output_shape, rank, &outputs[d], &ob);
        if (!rc && total > 0) {
            hipLaunchKernelGGL(k_ndgrid, dim3(grid_for(total)), dim3(kB), 0, c->stream, ab.data(), stride, extent, total, ob.data());
            c->tel.kernel_launches++;
        }
        stride *= extent;
    }
    if (rc == RMHIP_OK && hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "ndgrid: launch failed");
    if (rc)
        for (size_t d = 0; d < output_count; ++d)
            if (outputs[d]) rmhip_free(ctx, outputs[d]);
    return rc;
}

int rmhip_meshgrid(rmhip_ctx* ctx, const double* x, size_t nx, const double* y, size_t ny, const double* z_or_null, size_t nz, rmhip_buf* outputs) {
    CTX_OR_FAIL(ctx);
    if (!outputs || (nx && !x) || (ny && !y)) return fail(RMHIP_ERR_INVALID, "meshgrid: null argument");
    // ops/constructors.rs:230-308: X(iy, ix, iz) = x[ix], Y = y[iy], Z = z[iz] on [ny, nx] or [ny, nx, nz] - the N-D grid of the axes
    // (y, x, z) with the first two outputs exchanged; a one-point Z axis keeps the shape two-dimensional and still yields Z
    const size_t n_out = z_or_null ? 3 : 2;
    if (!z_or_null) nz = 1;
    const size_t shape[3] = {ny, nx, nz};
    const size_t rank = nz == 1 ? 2 : 3;
    rmhip_buf axes[3] = {0, 0, 0}, grids[3] = {0, 0, 0};
    const double* src[3] = {y, x, z_or_null};
    int rc = RMHIP_OK;
    for (size_t d = 0; d < n_out && rc == RMHIP_OK; ++d) {
        const size_t ashape[2] = {shape[d], 1};
        rc = rmhip_upload(ctx, src[d], ashape, 2, &axes[d]);
    }
    if (rc == RMHIP_OK) rc = rmhip_ndgrid(ctx, axes, n_out, shape, rank, n_out, grids);
    for (size_t d = 0; d < n_out; ++d)
        if (axes[d]) rmhip_free(ctx, axes[d]);
    if (rc != RMHIP_OK) return rc;
    outputs[0] = grids[1], outputs[1] = grids[0];
    if (n_out == 3) outputs[2] = grids[2];
    return RMHIP_OK;
}

int rmhip_sub2ind(rmhip_ctx* ctx, const size_t* dims, const size_t* strides, const rmhip_buf* inputs, const unsigned char* scalar_mask, size_t rank, size_t len,
                  const size_t* output_shape, size_t out_rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!dims || !strides || !inputs || !scalar_mask || !out || (out_rank && !output_shape)) return fail(RMHIP_ERR_INVALID, "sub2ind: null argument");
    if (rank == 0 || rank > (size_t)kMaxDims) return fail(RMHIP_ERR_UNSUPPORTED, "sub2ind: %zu dimensions", rank);
    if (product(output_shape, out_rank) != len) return fail(RMHIP_ERR_INVALID, "sub2ind: output shape does not match subscript sizes");
    SubArgs a;
    a.rank = (int)rank;
    std::vector<Buffer> bufs(rank);
    for (size_t d = 0; d < rank; ++d) {
        RMHIP_TRY(c->get(inputs[d], &bufs[d]));
        if (bufs[d].numel < (scalar_mask[d] ? (len ? 1 : 0) : len)) return fail(RMHIP_ERR_SHAPE, "sub2ind: subscript %zu has %zu elements, %zu needed", d + 1, bufs[d].numel, len);
        a.in[d] = bufs[d].data();
        a.dim[d] = dims[d];
        a.stride[d] = strides[d];
        a.scalar[d] = scalar_mask[d] ? 1 : 0;
    }
    Buffer ob;
    RMHIP_TRY(c->new_buffer(output_shape, out_rank, out, &ob));
    if (len == 0) return RMHIP_OK;
    std::shared_ptr<Allocation> flag;
    int rc = c->alloc_device(1, &flag);
    u64 first = ~0ull;
    if (!rc && hipMemcpyAsync(flag->ptr, &first, sizeof first, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(RMHIP_ERR_HIP, "sub2ind: copy failed");
    if (!rc) {
        hipLaunchKernelGGL(k_sub2ind, dim3(grid_for(len)), dim3(kB), 0, c->stream, a, (u64)len, ob.data(), (u64*)flag->ptr);
        c->tel.kernel_launches++;
        if (hipMemcpyAsync(&first, flag->ptr, sizeof first, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
            rc = fail(RMHIP_ERR_HIP, "sub2ind: read-back failed");
    }
    if (!rc && first != ~0ull) {  // the CPU's message for the first offending subscript (simple_provider.rs:2268-2291)
        const u64 i = first / kMaxDims;
        const int d = (int)(first % kMaxDims);
        double raw = 0.0;
        (void)hipMemcpy(&raw, a.in[d] + (a.scalar[d] ? 0 : i), sizeof raw, hipMemcpyDeviceToHost);
        if (!std::isfinite(raw)) rc = fail(RMHIP_ERR_INVALID, "sub2ind: subscript in dimension %d must be finite", d + 1);
        else if (std::fabs(std::round(raw) - raw) > 2.220446049250313e-16) rc = fail(RMHIP_ERR_INVALID, "sub2ind: subscript in dimension %d must be an integer", d + 1);
        else rc = fail(RMHIP_ERR_INVALID, "sub2ind: subscript %lld exceeds dimension %d (size %zu)", (long long)std::round(raw), d + 1, dims[d]);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_ind2sub(rmhip_ctx* ctx, const size_t* dims, const size_t* strides, size_t rank, rmhip_buf indices, size_t total, size_t len, const size_t* output_shape,
                  size_t out_rank, rmhip_buf* outputs) {
    CTX_OR_FAIL(ctx);
    if (!dims || !strides || !outputs || (out_rank && !output_shape)) return fail(RMHIP_ERR_INVALID, "ind2sub: null argument");
    if (rank == 0 || rank > (size_t)kMaxDims) return fail(RMHIP_ERR_UNSUPPORTED, "ind2sub: %zu dimensions", rank);
    Buffer ib;
    RMHIP_TRY(c->get(indices, &ib));
    if (product(output_shape, out_rank) != len) return fail(RMHIP_ERR_INVALID, "ind2sub: output shape does not match index tensor");
    if (ib.numel != len) return fail(RMHIP_ERR_INVALID, "ind2sub: index tensor length does not match provided shape");
    IndArgs a;
    a.rank = (int)rank;
    for (size_t d = 0; d < rank; ++d) outputs[d] = 0;
    int rc = RMHIP_OK;
    for (size_t d = 0; d < rank && rc == RMHIP_OK; ++d) {
        Buffer ob;
        rc = c->new_buffer(output_shape, out_rank, &outputs[d], &ob);
        a.out[d] = rc ? nullptr : ob.data();
        a.dim[d] = dims[d] ? dims[d] : 1;
        a.stride[d] = strides[d] ? strides[d] : 1;
    }
    u64 first = ~0ull;
    if (!rc && len > 0) {
        std::shared_ptr<Allocation> flag;
        rc = c->alloc_device(1, &flag);
        if (!rc && hipMemcpyAsync(flag->ptr, &first, sizeof first, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(RMHIP_ERR_HIP, "ind2sub: copy failed");
        if (!rc) {
            hipLaunchKernelGGL(k_ind2sub, dim3(grid_for(len)), dim3(kB), 0, c->stream, ib.data(), (u64)len, (u64)total, a, (u64*)flag->ptr);
            c->tel.kernel_launches++;
            if (hipMemcpyAsync(&first, flag->ptr, sizeof first, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
                rc = fail(RMHIP_ERR_HIP, "ind2sub: read-back failed");
        }
        if (!rc && first != ~0ull) {  // coerce_linear_index's wording (ind2sub.rs:326-353): the builtin passes a provider error on
            double raw = 0.0;
            (void)hipMemcpy(&raw, ib.data() + first, sizeof raw, hipMemcpyDeviceToHost);
            const double r = std::round(raw);
            if (!std::isfinite(raw) || std::fabs(r - raw) > 2.220446049250313e-16 || r < 1.0) rc = fail(RMHIP_ERR_INVALID, "Linear indices must be positive integers.");
            else rc = fail(RMHIP_ERR_INVALID, "Index exceeds number of array elements. Index must not exceed %zu.", total);
        }
    }
    if (rc)
        for (size_t d = 0; d < rank; ++d)
            if (outputs[d]) rmhip_free(ctx, outputs[d]);
    return rc;
}

int rmhip_scatter_line(rmhip_ctx* ctx, rmhip_buf matrix, int is_column, size_t index, rmhip_buf values, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb, vb, ob;
    RMHIP_TRY(c->get(matrix, &mb));
    RMHIP_TRY(c->get(values, &vb));
    const std::vector<size_t>& s = mb.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] != 1) return fail(RMHIP_ERR_UNSUPPORTED, "scatter_%s: matrix must be 2-D", is_column ? "column" : "row");
    const size_t rows = s.empty() ? 1 : s[0], cols = s.size() < 2 ? 1 : s[1];
    const size_t need = is_column ? rows : cols, limit = is_column ? cols : rows;
    if (index >= limit) return fail(RMHIP_ERR_INVALID, "scatter_%s: index %zu out of range (%zu)", is_column ? "column" : "row", index, limit);
    if (vb.numel != need) return fail(RMHIP_ERR_SHAPE, "scatter_%s: %zu values for %zu elements", is_column ? "column" : "row", vb.numel, need);
    RMHIP_TRY(c->new_buffer(s.data(), s.size(), out, &ob));
    if (ob.numel) {
        if (is_column) hipLaunchKernelGGL(k_scatter_line<true>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, mb.data(), vb.data(), (u64)rows, (u64)ob.numel, (u64)index, ob.data());
        else hipLaunchKernelGGL(k_scatter_line<false>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, mb.data(), vb.data(), (u64)rows, (u64)ob.numel, (u64)index, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_pow2_scale(rmhip_ctx* ctx, rmhip_buf mantissa, rmhip_buf exponent, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb, eb, ob;
    RMHIP_TRY(c->get(mantissa, &mb));
    RMHIP_TRY(c->get(exponent, &eb));
    if (mb.shape != eb.shape) return fail(RMHIP_ERR_SHAPE, "shape mismatch");  // simple_provider.rs:5835-5837
    RMHIP_TRY(c->new_buffer(mb.shape.data(), mb.shape.size(), out, &ob));
    if (ob.numel) {
        hipLaunchKernelGGL(k_pow2_scale, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, mb.data(), eb.data(), (u64)ob.numel, ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_round_digits(rmhip_ctx* ctx, rmhip_buf a, int digits, int significant, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (significant) return fail(RMHIP_ERR_UNSUPPORTED, "round_digits: significant-digit rounding takes floor(log10|x|) of the host's libm per element: not served");
    Buffer ab, ob;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), out, &ob));
    if (ob.numel) {
        hipLaunchKernelGGL(k_round_decimals, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), (u64)ob.numel, digits, powi10(digits), ob.data());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int rmhip_real_part(rmhip_ctx* ctx, int part, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (part < 0 || part > 3) return fail(RMHIP_ERR_INVALID, "real_part: part must be 0 (real), 1 (imag), 2 (conj) or 3 (angle)");
    Buffer ab, ob;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (part == 0 || part == 2) {  // real(x) = conj(x) = x on real storage
        RMHIP_HIP_CHECK(hipMemcpyAsync(ob.data(), ab.data(), ob.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return RMHIP_OK;
    }
    if (part == 1) hipLaunchKernelGGL(k_real_parts<1>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), (u64)ob.numel, ob.data());
    else hipLaunchKernelGGL(k_real_parts<0>, dim3(grid_for(ob.numel)), dim3(kB), 0, c->stream, ab.data(), (u64)ob.numel, ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_isreal(rmhip_ctx* ctx, rmhip_buf a, int* result) {
    CTX_OR_FAIL(ctx);
    if (!result) return fail(RMHIP_ERR_INVALID, "null result");
    Buffer ab;
    RMHIP_TRY(c->lookup(a, &ab));  // (an unknown handle is an error: simple_provider.rs:4786-4795)
    *result = ab.cplx ? 0 : 1;     // complex-interleaved storage only comes out of the transforms / complex constructors (fft.hip)
    return RMHIP_OK;
}
