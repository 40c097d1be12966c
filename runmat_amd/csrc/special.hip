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
#include "pow_tab.h"

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
    // fold the VEC * blockDim.x triples (entry e belongs to batch element e % batch): g1 = blockDim.x / batch threads per batch
    // element each merge their share of its ngroups entries in order, then a binary tree over the g1 results (fixed order; the first
    // version - 32 threads per element in chains of up to 64 merges, then one thread over the 32 - took longer than the streaming
    // part of a single 4K plane: 79 us)
    const int ngroups = VEC * (int)blockDim.x / batch, g1 = (int)blockDim.x / batch;
    __shared__ Mom s2[IN_BLOCK];
    // the triples go through LDS in two halves when VEC * blockDim.x exceeds the staging array (f32: four per thread)
    Mom mine2 = Mom{0.0, 0.0, 0.0};
    constexpr int CAP = IN_BLOCK * VEC > 2048 ? 2048 : IN_BLOCK * VEC;
    const int entries = VEC * (int)blockDim.x;
    const int bb = t % batch, g = t / batch;  // t < batch * g1 == blockDim.x
    for (int base = 0; base < entries; base += CAP) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < VEC; ++l) {
            const int e = VEC * t + l - base;
            if (e >= 0 && e < CAP) s[e] = acc[l];
        }
        __syncthreads();
        for (int j = g; j < ngroups; j += g1) {
            const int e = j * batch + bb - base;
            if (e >= 0 && e < CAP) mine2 = mom_merge(mine2, s[e]);
        }
    }
    s2[t] = mine2;
    __syncthreads();
    int h = 1;
    while (h < g1) h <<= 1;
    int cnt = g1;
    for (h >>= 1; h >= 1; h >>= 1) {
        if (g < h && g + h < cnt) s2[t] = mom_merge(s2[t], s2[t + h * batch]);
        cnt = h < cnt ? h : cnt;
        __syncthreads();
    }
    if (t < batch) partial[(size_t)blockIdx.x * batch + t] = s2[t];
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
template <bool TAB>
__device__ __forceinline__ double imgnorm_value(double x, double mu, double inv, int has_gain, double gain, int has_bias, double bias,
                                                int clamp_zero, int has_gamma, double gamma, const PowTables& tb) {
    double w = (x - mu) * inv;
    if (has_gain) w *= gain;
    if (has_bias) w += bias;
    if (clamp_zero) w = fmax(w, 0.0);  // f64::max: a NaN operand loses
    // gamma step: for a positive base table log -> multiply -> table exp (rm_pow_tab above; the library pow is ~200 instructions and
    // made this pass VALU-bound); for 0, +Inf, NaN, subnormal and negative bases (no clamp requested) and beyond |g ln w| > 700 the
    // library pow.  (+0 - half of a clamped image - is answered in place: pow(+0, g > 0) = +0)
    if (has_gamma) {
        if (gamma > 0.0 && w > 0.0) w = TAB ? rm_pow_tab(w, gamma, tb) : rm_pow_pos(w, gamma);
        else if (!(gamma > 0.0 && __double_as_longlong(w) == 0ll)) w = rm_pow_cold(w, gamma);
    }
    return w;
}

template <class T, int VEC, bool TAB = true>
__global__ void __launch_bounds__(IN_BLOCK) k_imgnorm_apply(const T* __restrict__ x, T* __restrict__ y, size_t total,
                                                       int batch, const double* __restrict__ stats, int has_gain, double gain,
                                                       int has_bias, double bias, int clamp_zero, int has_gamma, double gamma) {
    __shared__ __attribute__((aligned(16))) double s_pow[kPowLdsDoubles];
    PowTables tb{};
    if (TAB && has_gamma && gamma > 0.0) tb = pow_stage_tables(s_pow, threadIdx.x, blockDim.x);  // uniform branch: the barrier inside is safe
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
        for (int l = 0; l < VEC; ++l) v[l] = imgnorm_value<TAB>(v[l], mu[l], inv[l], has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma, tb);
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
        for (int l = 0; l < VEC; ++l)
            v[l] = imgnorm_value<false>(v[l], stats[b + l], stats[batch + b + l], has_gain, gain, has_bias, bias, clamp_zero, has_gamma, gamma, PowTables{});
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
    size_t want = (total / VEC + (size_t)threads * 8 - 1) / ((size_t)threads * 8);  // >= 8 vectors (one register chunk) per thread, grid-stride above the cap
    const size_t cap = (size_t)c->num_cus * 8;
    if (want < 1) want = 1;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    RMHIP_TRY(c->ensure_scratch(sizeof(Mom) * (size_t)grid * batch + sizeof(double) * 2 * batch));
    Mom* partial = reinterpret_cast<Mom*>(c->scratch);
    double* stats = reinterpret_cast<double*>(partial + (size_t)grid * batch);
    hipLaunchKernelGGL((k_plane_moments<T, VEC>), dim3(grid), dim3(threads), 0, c->stream, x, total, (int)batch, partial);
    hipLaunchKernelGGL(k_plane_moments_final, dim3((unsigned)batch), dim3(FIN_BLOCK), 0, c->stream, partial, (int)grid, (int)batch, (double)plane,
                       epsilon, stats);
    // the apply pass with the gamma step runs 256-thread blocks (round 4, table pow: 0.617 / 0.632 / 0.651 ms at 256 / 512 / 1024 threads,
    // 16 x 2160 x 3840; round 3's rm_pow_pos on the same box 0.721 / 0.744 / 0.757); without gamma the 1024-thread blocks stay (0.551
    // against 0.567 / 0.587 ms with 512 / 256).  RMHIP_IMG_APPLY_BLOCK overrides (A/B).
    static const int apply_env = std::getenv("RMHIP_IMG_APPLY_BLOCK") ? std::atoi(std::getenv("RMHIP_IMG_APPLY_BLOCK")) : 0;
    const int apply_block = apply_env > 0 ? apply_env : (has_gamma ? 256 : IN_BLOCK);
    const unsigned athreads = (unsigned)(((size_t)(apply_block < 64 ? 64 : (apply_block > IN_BLOCK ? IN_BLOCK : apply_block)) / batch) * batch);
    const unsigned at = athreads ? athreads : threads;
    size_t awant = (total / VEC + (size_t)at * 2 - 1) / ((size_t)at * 2);
    const size_t acap = (size_t)c->num_cus * 8 * (threads / at ? threads / at : 1);
    const unsigned agrid = (unsigned)(awant < acap ? (awant < 1 ? 1 : awant) : acap);
    static const bool pow_tab = !(std::getenv("RMHIP_IMG_POW") && std::getenv("RMHIP_IMG_POW")[0] == 'p');  // RMHIP_IMG_POW=pos: round 3's rm_pow_pos (A/B)
    if (pow_tab)
        hipLaunchKernelGGL((k_imgnorm_apply<T, VEC, true>), dim3(agrid), dim3(at), 0, c->stream, x, y, total, (int)batch, stats, has_gain, gain,
                           has_bias, bias, clamp_zero, has_gamma, gamma);
    else
        hipLaunchKernelGGL((k_imgnorm_apply<T, VEC, false>), dim3(agrid), dim3(at), 0, c->stream, x, y, total, (int)batch, stats, has_gain, gain,
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

// ---- tall-skinny (centred) Gram matrix: G = (X - 1 mu)' (X - 1 mu), X rows x cols column-major, cols <= 40 ---------------------------
// `cov` of many samples of a few variables (and `syrk` of such a matrix, mu = null).  The MFMA route runs a split-k product whose
// 256-wide tiles hold 8-32 useful columns and needs the centred copy first: 688 us for 2^20 x 8, 245 us for 2^18 x 32.  Here the
// matrix is read in place, coalesced (consecutive threads = consecutive rows of each column), the means subtracted on the way; a
// thread keeps the 8 x 8 products of one pair of eight-column blocks in registers (grid.y = the block pairs of the lower triangle,
// 1 / 3 / 6 / 10 / 15 for 8 / 16 / 24 / 32 / 40 columns: the second to tenth read of a row chunk comes from L2), a workgroup folds its threads
// in a fixed shuffle / LDS tree and the chunks are summed in order by a second kernel: deterministic, no atomics.  VALU fp64 - at 64
// fused multiply-adds per 16 loaded values the pass stays bound by the reads.
static constexpr int GS_BLOCK = 256;
// wave sum on the DPP network, result in lane 63 (a __shfl_down tree is ds_bpermute based and 64 of them per wave cost more than
// the streaming pass they follow); fixed order
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double gs_dpp(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double gs_wave_sum63(double v) {
    v += gs_dpp<0x111, 0xf>(v);  // row_shr:1
    v += gs_dpp<0x112, 0xf>(v);  // row_shr:2
    v += gs_dpp<0x114, 0xf>(v);  // row_shr:4
    v += gs_dpp<0x118, 0xf>(v);  // row_shr:8 -> lane 15 of every row holds the row sum
    v += gs_dpp<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += gs_dpp<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}
__device__ __forceinline__ void gs_pair(int p, int* jb, int* kb) {
    int j = 0;
    while ((j + 1) * (j + 2) / 2 <= p) ++j;
    *jb = j;
    *kb = p - j * (j + 1) / 2;
}
// Columns cols1 .. cols - 1 come from a second matrix x2 of the same height (least squares: the Gram matrix of [A | b] holds A'A and
// A'b, one pass over both).
template <class T>
__global__ void __launch_bounds__(GS_BLOCK) k_gram_skinny(const T* __restrict__ x, const T* __restrict__ x2, int cols1, size_t rows, int cols,
                                                          const double* __restrict__ mu, double* __restrict__ partial) {
    __shared__ double lds[GS_BLOCK / 64][64];
    int jb, kb;
    gs_pair((int)blockIdx.y, &jb, &kb);
    const int j0 = jb * 8, k0 = kb * 8;
    const bool diag = jb == kb;
    double mj[8], mk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        mj[u] = (mu && j0 + u < cols) ? mu[j0 + u] : 0.0;
        mk[u] = (mu && k0 + u < cols) ? mu[k0 + u] : 0.0;
    }
    double acc[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 8; ++v) acc[u][v] = 0.0;
    size_t chunk = (rows + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + GS_BLOCK - 1) / GS_BLOCK * GS_BLOCK;
    const size_t begin = (size_t)blockIdx.x * chunk;
    size_t end = begin + chunk;
    if (end > rows) end = rows;
    // columns past the last one are loaded from the last one and zeroed afterwards: a load under a condition is a branch whose
    // merge point waits for the value, which serialised the sixteen loads of an iteration (2^20 x 8: 30 us of kernel for 67 MB)
    const T* cj[8];
    const T* ck[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int jc = j0 + u < cols ? j0 + u : cols - 1, kc = k0 + u < cols ? k0 + u : cols - 1;
        cj[u] = jc < cols1 ? x + (size_t)jc * rows : x2 + (size_t)(jc - cols1) * rows;
        ck[u] = kc < cols1 ? x + (size_t)kc * rows : x2 + (size_t)(kc - cols1) * rows;
    }
    auto load_row = [&](size_t r, double (&a)[8], double (&b)[8]) {
        double ra[8], rb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = (double)cj[u][r];
        if (!diag) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rb[u] = (double)ck[u][r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = j0 + u < cols ? ra[u] - mj[u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = diag ? a[u] : (k0 + u < cols ? rb[u] - mk[u] : 0.0);
    };
    auto fold_row = [&](const double (&a)[8], const double (&b)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[u][v] = __builtin_fma(a[u], b[v], acc[u][v]);
    };
    size_t r = begin + threadIdx.x;
    for (; r + GS_BLOCK < end; r += 2 * GS_BLOCK) {  // two rows' loads in flight before the first product
        double a0[8], b0[8], a1[8], b1[8];
        load_row(r, a0, b0);
        load_row(r + GS_BLOCK, a1, b1);
        fold_row(a0, b0);
        fold_row(a1, b1);
    }
    if (r < end) {
        double a0[8], b0[8];
        load_row(r, a0, b0);
        fold_row(a0, b0);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const double s = gs_wave_sum63(acc[u][v]);
            if (lane == 63) lds[wave][u * 8 + v] = s;
        }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        partial[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 + t] = ((lds[0][t] + lds[1][t]) + lds[2][t]) + lds[3][t];
    }
}
// G (cols x cols, both triangles) from the chunk partials: 16 groups of 64 threads each sum every 16th chunk in order, the group sums
// are added in group order (one thread summing 256 partials one load after the other took longer than the streaming pass)
static constexpr int GS_FGROUPS = 16;
// The covariance form divides by the denominator and applies sanitize_covariance's diagonal rule (cov.rs:1218-1227) on the way out.
__global__ void __launch_bounds__(64 * GS_FGROUPS) k_gram_skinny_final(const double* __restrict__ partial, int nchunks, int npairs, int cols,
                                                                        double denom, int sanitize, double* __restrict__ g) {
    __shared__ double lds[GS_FGROUPS][64];
    int jb, kb;
    gs_pair((int)blockIdx.x, &jb, &kb);
    const int t = threadIdx.x & 63, q = threadIdx.x >> 6, j = jb * 8 + (t >> 3), k = kb * 8 + (t & 7);
    double part = 0.0;
    for (int cchunk = q; cchunk < nchunks; cchunk += GS_FGROUPS) part += partial[((size_t)cchunk * npairs + blockIdx.x) * 64 + t];
    lds[q][t] = part;
    __syncthreads();
    if (q != 0) return;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < GS_FGROUPS; ++w) s += lds[w][t];
    if (j >= cols || k >= cols) return;
    if (jb == kb && k > j) return;  // the diagonal blocks hold both triangles: keep the lower one, mirror it
    s = s / denom;
    if (sanitize && j == k && s == s && fabs(s) != __builtin_inf() && s < 0.0 && s > -1.0e-12) s = 0.0;
    g[(size_t)j + (size_t)k * cols] = s;
    g[(size_t)k + (size_t)j * cols] = s;
}
bool gram_skinny_applies(size_t rows, size_t cols) { return cols >= 1 && cols <= 40 && rows >= 4096 && rows >= 64 * cols; }
template <class T>
static int gram_skinny_any(Context* c, const T* x, size_t rows, size_t cols, const double* mu, double denom, bool sanitize, double* g, const T* x2,
                           size_t cols2) {
    const size_t cols1 = cols;
    cols += x2 ? cols2 : 0;  // g is (cols1 + cols2)^2
    const int nb = (int)((cols + 7) / 8), npairs = nb * (nb + 1) / 2;
    // the loop is a load -> 64 fma chain per row with nothing else in flight: two workgroups per CU (the register budget's
    // occupancy) with two rows' loads ahead of the products; one per CU without the unrolling ran 62 us for 2^20 x 8 (67 MB)
    size_t nchunks = (rows + (size_t)GS_BLOCK * 8 - 1) / ((size_t)GS_BLOCK * 8);  // >= 8 rows per thread
    static const int gs_bpc = std::getenv("RMHIP_GRAM_BPC") ? std::atoi(std::getenv("RMHIP_GRAM_BPC")) : 2;  // dev knob (A/B): workgroups per CU
    const size_t cap = (size_t)c->num_cus * (size_t)(gs_bpc > 0 ? gs_bpc : 2) / (size_t)npairs;
    if (nchunks > cap) nchunks = cap ? cap : 1;
    if (nchunks < 1) nchunks = 1;
    RMHIP_TRY(c->ensure_scratch(sizeof(double) * nchunks * (size_t)npairs * 64));
    double* partial = c->scratch;
    hipLaunchKernelGGL(k_gram_skinny<T>, dim3((unsigned)nchunks, (unsigned)npairs), dim3(GS_BLOCK), 0, c->stream, x, x2 ? x2 : x, (int)cols1, rows, (int)cols, mu,
                       partial);
    hipLaunchKernelGGL(k_gram_skinny_final, dim3((unsigned)npairs), dim3(64 * GS_FGROUPS), 0, c->stream, (const double*)partial, (int)nchunks, npairs, (int)cols, denom,
                       sanitize ? 1 : 0, g);
    c->tel.kernel_launches += 2;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int gram_skinny_device(Context* c, const double* x, size_t rows, size_t cols, const double* mu, double denom, bool sanitize, double* g,
                       const double* x2, size_t cols2) {
    return gram_skinny_any(c, x, rows, cols, mu, denom, sanitize, g, x2, cols2);
}
int gram_skinny_device_f32(Context* c, const float* x, size_t rows, size_t cols, const double* mu, double denom, bool sanitize, double* g) {
    return gram_skinny_any<float>(c, x, rows, cols, mu, denom, sanitize, g, nullptr, 0);
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
