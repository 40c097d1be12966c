// special.hip -- provider hooks behind RunMat's special fusion patterns:
//   image_normalize   crates/runmat-accelerate-api/src/lib.rs:2407-2413, descriptor :3563-3577
//                     CPU semantics crates/runmat-accelerate/src/simple_provider.rs:7893-7993
//                     (== cpu_image_normalize, crates/runmat-accelerate/tests/image_normalize.rs:7-66)
// The tensor is [batch, height, width] column-major, i.e. the BATCH index is the fastest one and a
// plane's elements lie `batch` doubles apart.  Per batch element: mean over the plane, two-pass variance
// (sum of (x - mean)^2 / plane), sigma = sqrt(var + eps), inv = sigma > 0 ? 1/sigma : 0,
// y = (x - mean) * inv [* gain] [+ bias] [max 0] [powf gamma].
//
// HBM-bound: TWO reads and one write per element (24 B).  The CPU computes a two-pass variance (mean, then the squared
// deviations: two reads of the plane); here the plane is read once for both statistics: every thread takes its
// elements in register-resident chunks, forms each chunk's mean and its squared deviations about THAT mean (an exact
// two-pass inside the chunk), and merges (count, mean, M2) triples with the pairwise update of Chan, Golub & LeVeque -
// per thread, per block (LDS) and across blocks, each in a fixed order (no atomics: run-to-run deterministic).  The
// result has the two-pass algorithm's accuracy (no cancellation: every M2 term is a sum of squares of small
// deviations) and differs from the CPU's bits only in the last place or two (tests: 1e-12 relative).  Every thread
// walks the linear index with a stride that is a multiple of `batch`, so its batch element is fixed (no per-element
// modulo) and loads stay coalesced.
#include <cstdlib>
#include "common.h"
#include "skel_common.h"

namespace rmhip {

static constexpr int IN_MAX_BATCH = 256;  // above: the statistics by the strided moments reduction (reduce2.hip), flat apply threads
static constexpr int IN_BLOCK = 1024;  // streaming passes: the largest multiple of `batch` <= 1024 threads per block

// T = storage type (float on a precision-32 provider); sums and statistics are f64 either way.
// VEC = elements per access (a 16-byte vector when the element count allows it: a thread's VEC consecutive elements are VEC
// consecutive batch indices - modulo the extent - and every stride, a multiple of the extent, keeps them fixed).
template <class T, int VEC>
struct VecT {
    typedef T type __attribute__((ext_vector_type(VEC)));
};
template <class T>
struct VecT<T, 1> {
    typedef T type;
};
template <class T, int VEC>
__device__ __forceinline__ void load_vec(const T* __restrict__ x, size_t vi, double (&out)[VEC]) {
    // streamed once per pass: non-temporal, as in the fused elementwise kernels (worth 5 % on loads, 10 % on stores there)
    if constexpr (VEC == 1) {
        out[0] = (double)__builtin_nontemporal_load(x + vi);
    } else {
        const typename VecT<T, VEC>::type v = __builtin_nontemporal_load((const typename VecT<T, VEC>::type*)x + vi);
#pragma unroll
        for (int l = 0; l < VEC; ++l) out[l] = (double)v[l];
    }
}

// ---- one-pass plane statistics ----------------------------------------------------------------------------------------
struct Mom {  // count, mean, sum of squared deviations about the mean
    double n, mean, m2;
};
__device__ __forceinline__ Mom mom_merge(const Mom a, const Mom b) {  // Chan et al.: exact in exact arithmetic, order-dependent in rounding
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    const double n = a.n + b.n, d = b.mean - a.mean, r = 1.0 / n;
    return Mom{n, a.mean + d * (b.n * r), a.m2 + b.m2 + d * d * (a.n * b.n * r)};
}
// partial[block][b] = (count, mean, M2) of this block's share of plane b
template <class T, int VEC>
__global__ void __launch_bounds__(IN_BLOCK) k_plane_moments(const T* __restrict__ x, size_t total, int batch, Mom* __restrict__ partial) {
    __shared__ Mom s[IN_BLOCK * VEC > 2048 ? 2048 : IN_BLOCK * VEC];
    const int t = threadIdx.x;
    const size_t nvec = total / VEC;  // total % VEC == 0 (the host's choice of VEC)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    constexpr int CH = 8;  // vectors per chunk: CH * VEC values live in registers while the chunk's two passes run
    Mom acc[VEC];
#pragma unroll
    for (int l = 0; l < VEC; ++l) acc[l] = Mom{0.0, 0.0, 0.0};
    size_t i = (size_t)blockIdx.x * blockDim.x + t;
    for (; i + (CH - 1) * stride < nvec; i += CH * stride) {
        double v[CH][VEC];
#pragma unroll
        for (int u = 0; u < CH; ++u) load_vec<T, VEC>(x, i + u * stride, v[u]);
        const double r = 1.0 / (acc[0].n + (double)CH);  // every lane of the thread has seen the same number of values
#pragma unroll
        for (int l = 0; l < VEC; ++l) {
            double sum = 0.0;
#pragma unroll
            for (int u = 0; u < CH; ++u) sum += v[u][l];
            const double cm = sum * (1.0 / CH);
            double m2 = 0.0;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const double d = v[u][l] - cm;
                m2 += d * d;
            }
            const double d = cm - acc[l].mean;
            acc[l].m2 += m2 + d * d * (acc[l].n * (double)CH * r);
            acc[l].mean += d * ((double)CH * r);
            acc[l].n += (double)CH;
        }
    }
    for (; i < nvec; i += stride) {  // ragged tail: one value at a time (Welford)
        double v[VEC];
        load_vec<T, VEC>(x, i, v);
#pragma unroll
        for (int l = 0; l < VEC; ++l) {
            acc[l].n += 1.0;
            const double d = v[l] - acc[l].mean;
            acc[l].mean += d / acc[l].n;
            acc[l].m2 += d * (v[l] - acc[l].mean);
        }
    }
    // fold the VEC * blockDim.x triples (entry e belongs to batch element e % batch) in two levels, each in a fixed order
    const int ngroups = VEC * (int)blockDim.x / batch, m = (int)blockDim.x / batch;
    const int g1 = ngroups < 32 ? (ngroups < m ? ngroups : m) : (m < 32 ? m : 32);
    __shared__ Mom s2[IN_BLOCK];
    // the triples go through LDS in two halves when VEC * blockDim.x exceeds the staging array (f32: four per thread)
    Mom mine2 = Mom{0.0, 0.0, 0.0};
    constexpr int CAP = IN_BLOCK * VEC > 2048 ? 2048 : IN_BLOCK * VEC;
    const int entries = VEC * (int)blockDim.x;
    for (int base = 0; base < entries; base += CAP) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < VEC; ++l) {
            const int e = VEC * t + l - base;
            if (e >= 0 && e < CAP) s[e] = acc[l];
        }
        __syncthreads();
        if (t < batch * g1) {
            const int bb = t % batch, g = t / batch;
            for (int j = g; j < ngroups; j += g1) {
                const int e = j * batch + bb - base;
                if (e >= 0 && e < CAP) mine2 = mom_merge(mine2, s[e]);
            }
        }
    }
    if (t < batch * g1) s2[t] = mine2;
    __syncthreads();
    if (t < batch) {
        Mom a = Mom{0.0, 0.0, 0.0};
        for (int g = 0; g < g1; ++g) a = mom_merge(a, s2[g * batch + t]);
        partial[(size_t)blockIdx.x * batch + t] = a;
    }
}
// stats[b] = mean, stats[batch + b] = 1 / sqrt(M2 / plane + eps) (0 when sigma is not positive) from the block triples: one
// 256-thread block per plane b - thread t folds blocks t, t + 256, ... in that order, the 256 results merge in a fixed shuffle /
// LDS tree.  (Round 2: ONE block for all planes, every thread a serial chain of 32 merges and then 16 threads a serial chain of
// 64 more - 22 us, 3 % of a 16 x 2160 x 3840 call, for a few hundred KiB of partials.)
static constexpr int FIN_BLOCK = 256;
__device__ __forceinline__ Mom mom_shfl_down(const Mom a, int off) {
    return Mom{__shfl_down(a.n, off, 64), __shfl_down(a.mean, off, 64), __shfl_down(a.m2, off, 64)};
}
__global__ void __launch_bounds__(FIN_BLOCK) k_plane_moments_final(const Mom* __restrict__ partial, int nblocks, int batch, double plane,
                                                                   double epsilon, double* __restrict__ stats) {
    __shared__ Mom s[FIN_BLOCK / 64];
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63;
    Mom a = Mom{0.0, 0.0, 0.0};
    int k = t;
    for (; k + 3 * FIN_BLOCK < nblocks; k += 4 * FIN_BLOCK) {  // four independent loads in flight per trip
        const Mom v0 = partial[(size_t)k * batch + b], v1 = partial[(size_t)(k + FIN_BLOCK) * batch + b],
                  v2 = partial[(size_t)(k + 2 * FIN_BLOCK) * batch + b], v3 = partial[(size_t)(k + 3 * FIN_BLOCK) * batch + b];
        a = mom_merge(mom_merge(mom_merge(mom_merge(a, v0), v1), v2), v3);
    }
    for (; k < nblocks; k += FIN_BLOCK) a = mom_merge(a, partial[(size_t)k * batch + b]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Mom o = mom_shfl_down(a, off);
        if ((lane & (2 * off - 1)) == 0) a = mom_merge(a, o);
    }
    if (lane == 0) s[t >> 6] = a;
    __syncthreads();
    if (t != 0) return;
    Mom tot = s[0];
    for (int w = 1; w < FIN_BLOCK / 64; ++w) tot = mom_merge(tot, s[w]);
    stats[b] = tot.mean;
    const double variance = tot.m2 / plane;
    const double sigma = sqrt(variance + epsilon);
    stats[batch + b] = sigma > 0.0 ? 1.0 / sigma : 0.0;
}

// one pixel: y = (x - mean) * inv [* gain] [+ bias] [max 0] [^ gamma]
__device__ __forceinline__ double imgnorm_value(double x, double mu, double inv, int has_gain, double gain, int has_bias, double bias,
                                                int clamp_zero, int has_gamma, double gamma) {
    double w = (x - mu) * inv;
    if (has_gain) w *= gain;
    if (has_bias) w += bias;
    if (clamp_zero) w = fmax(w, 0.0);  // f64::max: a NaN operand loses
    // gamma step: for a positive base one fused log -> multiply -> exp (rm_pow_pos, skel_common.h: ~55 instructions, the
    // library pow is ~200 and made this pass VALU-bound); relative error <= (0.45 |g ln w| + 1.5) 2^-53, |g ln w| <= 32,
    // beyond that (pixels below e^(-32 / g)) and for 0, +Inf, NaN and negative bases (no clamp requested) the library pow.
    // (+0 - half of a clamped image - is answered in place: pow(+0, g > 0) = +0)
    if (has_gamma) {
        if (gamma > 0.0 && w > 0.0) w = rm_pow_pos(w, gamma);
        else if (!(gamma > 0.0 && __double_as_longlong(w) == 0ll)) w = rm_pow_cold(w, gamma);
    }
    return w;
}

template <class T, int VEC>
__global__ void __launch_bounds__(IN_BLOCK) k_imgnorm_apply(const T* __restrict__ x, T* __restrict__ y, size_t total,
                                                       int batch, const double* __restrict__ stats, int has_gain, double gain,
                                                       int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    const int b = (VEC * (int)threadIdx.x) % batch;
    double mu[VEC], inv[VEC];
#pragma unroll
    for (int l = 0; l < VEC; ++l) {
        const int bl = (b + l) % batch;  // VEC need not divide an odd batch extent: the thread's elements wrap around it
        mu[l] = stats[bl];
        inv[l] = stats[batch + bl];
    }
    const size_t nvec = total / VEC;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // (fetching four vectors per trip before touching any of them - the gamma step is ~60 instructions per element and the pass runs
    // at 4.4 TB/s against 5.9 without it - measured WORSE: 0.645 -> 0.678 ms with gamma, 0.556 -> 0.613 without)
    constexpr int APPLY_U = 1;
    auto one = [&](double (&v)[VEC], size_t i) {
#pragma unroll
        for (int l = 0; l < VEC; ++l) v[l] = imgnorm_value(v[l], mu[l], inv[l], has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
        if constexpr (VEC == 1) {
            __builtin_nontemporal_store((T)v[0], y + i);
        } else {
            typename VecT<T, VEC>::type r;
#pragma unroll
            for (int l = 0; l < VEC; ++l) r[l] = (T)v[l];
            __builtin_nontemporal_store(r, (typename VecT<T, VEC>::type*)y + i);
        }
    };
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (APPLY_U - 1) * stride < nvec; i += APPLY_U * stride) {
        double v[APPLY_U][VEC];
#pragma unroll
        for (int u = 0; u < APPLY_U; ++u) load_vec<T, VEC>(x, i + u * stride, v[u]);
#pragma unroll
        for (int u = 0; u < APPLY_U; ++u) one(v[u], i + u * stride);
    }
    for (; i < nvec; i += stride) {
        double v[VEC];
        load_vec<T, VEC>(x, i, v);
        one(v, i);
    }
}

// more planes than a block has threads for (batch > IN_MAX_BATCH): flat threads, the plane of an element from its index
template <int VEC>
__global__ void __launch_bounds__(256) k_imgnorm_apply_flat(const double* __restrict__ x, double* __restrict__ y, size_t total, unsigned batch,
                                                            const double* __restrict__ stats, int has_gain, double gain, int has_bias, double bias,
                                                            int clamp_zero, int has_gamma, double gamma) {
    const size_t nvec = total / VEC, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        double v[VEC];
        load_vec<double, VEC>(x, i, v);
        const unsigned b = (unsigned)((i * VEC) % batch);  // VEC divides batch: b + l stays below it
#pragma unroll
        for (int l = 0; l < VEC; ++l) v[l] = imgnorm_value(v[l], stats[b + l], stats[batch + b + l], has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
        if constexpr (VEC == 1) {
            __builtin_nontemporal_store(v[0], y + i);
        } else {
            typename VecT<double, VEC>::type r;
#pragma unroll
            for (int l = 0; l < VEC; ++l) r[l] = v[l];
            __builtin_nontemporal_store(r, (typename VecT<double, VEC>::type*)y + i);
        }
    }
}
static int image_normalize_many(Context* c, const double* x, double* y, size_t batch, size_t plane, double epsilon, int has_gain, double gain,
                                int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    if (batch > 0xffffffffULL) return fail(RMHIP_ERR_UNSUPPORTED, "image_normalize: batch %zu not supported by provider", batch);
    std::shared_ptr<Allocation> st;
    RMHIP_TRY(c->alloc_device(2 * batch, &st));
    RMHIP_TRY(launch_plane_stats(c, x, batch, plane, epsilon, st->ptr));
    const size_t total = batch * plane;
    const bool v2 = batch % 2 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    const size_t nvec = v2 ? total / 2 : total;
    const size_t want = (nvec + 511) / 512, cap = (size_t)c->num_cus * 32;
    const unsigned grid = (unsigned)(want < cap ? (want < 1 ? 1 : want) : cap);
    if (v2)
        hipLaunchKernelGGL(k_imgnorm_apply_flat<2>, dim3(grid), dim3(256), 0, c->stream, x, y, total, (unsigned)batch, (const double*)st->ptr, has_gain, gain,
                           has_bias, bias, clamp_zero, has_gamma, gamma);
    else
        hipLaunchKernelGGL(k_imgnorm_apply_flat<1>, dim3(grid), dim3(256), 0, c->stream, x, y, total, (unsigned)batch, (const double*)st->ptr, has_gain, gain,
                           has_bias, bias, clamp_zero, has_gamma, gamma);
    c->tel.kernel_launches += 1;
    RMHIP_HIP_CHECK(hipGetLastError());
    // (the statistics go back to the pool on return: reuse is stream-ordered, rmhip_core.cpp release_device)
    return RMHIP_OK;
}

template <class T, int VEC>
static int image_normalize_vec(Context* c, const T* x, T* y, size_t batch, size_t height, size_t width, double epsilon,
                               int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    const size_t plane = height * width, total = batch * plane;
    const unsigned threads = (unsigned)((IN_BLOCK / batch) * batch);  // a multiple of batch
    size_t want = (total / VEC + (size_t)threads * 2 - 1) / ((size_t)threads * 2);  // >= 2 vectors per thread, grid-stride above the cap
    const size_t cap = (size_t)c->num_cus * 8;
    if (want < 1) want = 1;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    RMHIP_TRY(c->ensure_scratch(sizeof(Mom) * (size_t)grid * batch + sizeof(double) * 2 * batch));
    Mom* partial = reinterpret_cast<Mom*>(c->scratch);
    double* stats = reinterpret_cast<double*>(partial + (size_t)grid * batch);
    hipLaunchKernelGGL((k_plane_moments<T, VEC>), dim3(grid), dim3(threads), 0, c->stream, x, total, (int)batch, partial);
    hipLaunchKernelGGL(k_plane_moments_final, dim3((unsigned)batch), dim3(FIN_BLOCK), 0, c->stream, partial, (int)grid, (int)batch, (double)plane,
                       epsilon, stats);
    // the apply pass with the gamma step runs 512-thread blocks: with the cold pow out of line (skel_common.h: rm_pow_cold) it needs 51
    // registers instead of 84 and three such blocks share a CU - 0.645 -> 0.633 ms at 16 x 2160 x 3840; without gamma the 1024-thread
    // blocks stay (0.551 against 0.567 / 0.587 ms with 512 / 256).  RMHIP_IMG_APPLY_BLOCK overrides (A/B).
    static const int apply_env = std::getenv("RMHIP_IMG_APPLY_BLOCK") ? std::atoi(std::getenv("RMHIP_IMG_APPLY_BLOCK")) : 0;
    const int apply_block = apply_env > 0 ? apply_env : (has_gamma ? 512 : IN_BLOCK);
    const unsigned athreads = (unsigned)(((size_t)(apply_block < 64 ? 64 : (apply_block > IN_BLOCK ? IN_BLOCK : apply_block)) / batch) * batch);
    const unsigned at = athreads ? athreads : threads;
    size_t awant = (total / VEC + (size_t)at * 2 - 1) / ((size_t)at * 2);
    const size_t acap = (size_t)c->num_cus * 8 * (threads / at ? threads / at : 1);
    const unsigned agrid = (unsigned)(awant < acap ? (awant < 1 ? 1 : awant) : acap);
    hipLaunchKernelGGL((k_imgnorm_apply<T, VEC>), dim3(agrid), dim3(at), 0, c->stream, x, y, total, (int)batch, stats, has_gain, gain,
                       has_bias, bias, clamp_zero, has_gamma, gamma);
    c->tel.kernel_launches += 3;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
template <class T>
static int image_normalize_any(Context* c, const T* x, T* y, size_t batch, size_t height, size_t width, double epsilon,
                               int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    if (batch * height * width == 0) return RMHIP_OK;
    if (batch > (size_t)IN_MAX_BATCH) {
        if constexpr (sizeof(T) == 8)
            return image_normalize_many(c, x, y, batch, height * width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
        else
            return fail(RMHIP_ERR_UNSUPPORTED, "image_normalize: batch %zu > %d takes the f64 kernels", batch, IN_MAX_BATCH);  // rmhip_ops.cpp widens first
    }
    // 16-byte accesses whenever the element count and the alignment allow them: a vector's elements are consecutive batch indices
    // (wrapping around an extent the width does not divide) and every stride is a multiple of the extent, so they stay fixed
    const bool a16 = ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    const size_t total = batch * height * width;
    constexpr int WIDE = 16 / (int)sizeof(T);
    if (a16 && total % WIDE == 0)
        return image_normalize_vec<T, WIDE>(c, x, y, batch, height, width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
    if constexpr (sizeof(T) == 4) {
        if (a16 && total % 2 == 0)
            return image_normalize_vec<T, 2>(c, x, y, batch, height, width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
    }
    return image_normalize_vec<T, 1>(c, x, y, batch, height, width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
}
int image_normalize_device(Context* c, const double* x, double* y, size_t batch, size_t height, size_t width, double epsilon,
                           int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    return image_normalize_any(c, x, y, batch, height, width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
}
int image_normalize_device_f32(Context* c, const float* x, float* y, size_t batch, size_t height, size_t width, double epsilon,
                               int has_gain, double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    return image_normalize_any(c, x, y, batch, height, width, epsilon, has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma);
}

// ---- diag_extract (lib.rs:1625-1632; simple_provider.rs:3281-3312, index rule :2386-2392) ----------------
__global__ void __launch_bounds__(256) k_diag_extract(const double* __restrict__ a, size_t rows, long long offset, size_t len,
                                                      double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const size_t r = offset >= 0 ? i : i + (size_t)(-offset), c = offset >= 0 ? i + (size_t)offset : i;
    out[i] = a[r + c * rows];
}

int diag_extract_device(Context* c, const double* a, size_t rows, long long offset, size_t len, double* out) {
    if (len == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_diag_extract, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, a, rows, offset, len, out);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// sanitize_covariance (cov.rs:1218-1227): a finite diagonal entry in (-1e-12, 0) is rounding noise -> 0
__global__ void __launch_bounds__(256) k_cov_sanitize_diag(double* __restrict__ cm, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = cm[i + i * n];
    if (v == v && fabs(v) != __builtin_inf() && v < 0.0 && v > -1.0e-12) cm[i + i * n] = 0.0;
}

int cov_sanitize_diag_device(Context* c, double* cm, size_t n) {
    if (n == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_cov_sanitize_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, cm, n);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

}  // namespace rmhip
