// signal_ops.hip -- convolutions and window generators of the signal builtins.
//   conv1d            crates/runmat-accelerate-api/src/lib.rs:2535-2542   (builtins/math/signal/conv.rs:481-517; simple_provider.rs:1780-1842, 6015-6064)
//   conv2d            lib.rs:2543-2550    (builtins/math/signal/conv2.rs:595-640; simple_provider.rs:6065-6154)
//   hann_window / hamming_window / blackman_window   lib.rs:1797-1807   (simple_provider.rs:95-120, 6453-6472)
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
    for (u64 i = lo; i <= hi; ++i) {
        const double p = a[i] * bb[n - i];
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
    const u64 R = r0 + o % rows, Cc = c0 + o / rows;
    const u64 ac_lo = Cc >= bc_n - 1 ? Cc - (bc_n - 1) : 0, ac_hi = Cc < ac_n - 1 ? Cc : ac_n - 1;
    const u64 ar_lo = R >= br_n - 1 ? R - (br_n - 1) : 0, ar_hi = R < ar_n - 1 ? R : ar_n - 1;
    double acc = 0.0;
    for (u64 ac = ac_lo; ac <= ac_hi; ++ac) {
        const double* acol = a + ac * ar_n;
        const double* bcol = bb + (bc_n - 1 - (Cc - ac)) * br_n;
        for (u64 ar = ar_lo; ar <= ar_hi; ++ar) {
            const double p = acol[ar] * bcol[br_n - 1 - (R - ar)];
            acc = acc + p;
        }
    }
    out[o] = acc;
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
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));
    if (len == 0) return RMHIP_OK;
    if (len > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "conv1d: %llu outputs", len);
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
    if (ob.numel > 0x7fffffffull * kB) return fail(RMHIP_ERR_UNSUPPORTED, "conv2d: %zu outputs", ob.numel);
    const int in_lds = bb.numel <= LDS_TAPS;
    hipLaunchKernelGGL(k_conv2d, dim3(grid_for(ob.numel)), dim3(kB), in_lds ? bb.numel * sizeof(double) : 0, c->stream, ab.data(), ar_n, ac_n, bb.data(), br_n, bc_n, r0, c0,
                       rows, cols, in_lds, ob.data());
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
