// rng.hip -- `random_uniform` / `random_normal` on the CPU-parity stream
// (crates/runmat-accelerate-api/src/lib.rs:1713-1728, 1772).
//
// The reference CPU generator (crates/runmat-runtime/src/builtins/common/random.rs) is a serial
// 64-bit LCG  s <- s*6364136223846793005 + 1 (:7-13, :271-278), uniform = (s >> 11) * 2^-53, and
// Box-Muller on consecutive pairs (u1, u2) -> (r cos t, r sin t) emitted consecutively
// (:279-288, :530-543).  An LCG admits O(log n) skip-ahead (the reference's own advance_state,
// :238-256), so the SAME stream is generated in parallel here: thread g starts at step 2g
// (pairs) or g (uniforms) via skip-ahead and then jumps by the grid stride with a precomputed
// (multiplier, increment) pair.  The integer stream is bit-exact with the CPU; only the libm
// calls (log, sqrt, cos, sin) differ by rounding.  The reference's wgpu path uses a different
// generator (Philox, backend/wgpu/shaders/creation.rs:707-794) and therefore a different stream
// from its own CPU path; parity here is with the CPU.
#include "common.h"
#include "skel_common.h"
#include "rng_tables.h"

namespace rmhip {

#include "skel_rng.h"

void lcg_jump_host(unsigned long long delta, unsigned long long* mult, unsigned long long* plus) { lcg_jump(delta, mult, plus); }

uint64_t lcg_advance(uint64_t state, uint64_t delta) {
    unsigned long long m, p;
    lcg_jump(delta, &m, &p);
    return m * state + p;
}

// T = storage type: double, or float on a precision-32 provider (the f64 stream rounded on store -- what the CPU's
// rand/randn(..., 'single') produce).
// A thread follows the states it EMITS, not the ones before them: x_i -> x_(i + stride) is the same affine jump whatever the
// starting point, so an element costs one 64-bit multiply (three quarter-rate 32-bit ones) instead of two.
// KIND 0: u; 1: a + (b - a) u with d = b - a rounded on the host as the CPU rounds it (random.rs:514-528); 3: lower + min(floor(u span),
// span - 1) in integer arithmetic, converted once (simple_provider.rs:3707-3714)
struct UniformXform {
    double a, d;           // KIND 1
    long long lower;       // KIND 3
    unsigned long long span;
};
template <class T, int KIND>
__global__ void __launch_bounds__(256) k_rng_uniform(unsigned long long state, T* __restrict__ out, size_t n,
                                                     unsigned long long jm, unsigned long long jp, UniformXform xf) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= n) return;
    unsigned long long x = lcg_skip2(state, 256ULL * blockIdx.x, threadIdx.x + 1ULL);  // the state element g is drawn from
    for (size_t i = g; i < n; i += stride) {
        const double u = lcg_uniform(x);
        double v = u;
        if (KIND == 1) v = xf.a + xf.d * u;
        if (KIND == 3) {
            unsigned long long off = (unsigned long long)__builtin_floor(u * (double)xf.span);
            if (off >= xf.span) off = xf.span - 1;
            v = (double)(xf.lower + (long long)off);
        }
        out[i] = (T)v;
        x = jm * x + jp;  // jump `stride` steps
    }
}

typedef float v2s __attribute__((ext_vector_type(2)));
template <class T>
struct PairOf;
template <>
struct PairOf<double> {
    typedef v2d type;
};
template <>
struct PairOf<float> {
    typedef v2s type;
};

// `random_exponential` (random.rs:290-300): -mu ln(max(u, MIN_POSITIVE)), the table logarithm of the Box-Muller radius (<= 1 ulp)
template <class T>
__global__ void __launch_bounds__(256) k_rng_exponential(unsigned long long state, T* __restrict__ out, size_t n, unsigned long long jm,
                                                         unsigned long long jp, double mu) {
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= n) return;
    unsigned long long x = lcg_skip2(state, 256ULL * blockIdx.x, threadIdx.x + 1ULL);
    for (size_t i = g; i < n; i += stride) {
        double ln_u;
        (void)bm_radius(x, tb, &ln_u);
        out[i] = (T)(-mu * ln_u);
        x = jm * x + jp;
    }
}

// SCALED: mu + sigma z (generate_normal_scaled, random.rs:302-320)
// Affine maps of the first 256 odd step counts: thread t of a block draws u1 of its first pair 2 t + 1 steps past the block's start
// state - one 64-bit multiply-add from this table instead of a skip-ahead loop per thread (computed by the compiler).
struct LcgSkipTable {
    unsigned long long mp[512];
};
static constexpr LcgSkipTable make_skip_table() {
    LcgSkipTable t{};
    for (int i = 0; i < 256; ++i) {
        unsigned long long delta = 2ULL * i + 1ULL, cur_mult = 6364136223846793005ULL, cur_plus = 1ULL, acc_mult = 1ULL, acc_plus = 0ULL;
        while (delta > 0) {
            if (delta & 1ULL) {
                acc_mult = acc_mult * cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = cur_plus * (cur_mult + 1ULL);
            cur_mult = cur_mult * cur_mult;
            delta >>= 1;
        }
        t.mp[2 * i] = acc_mult;
        t.mp[2 * i + 1] = acc_plus;
    }
    return t;
}
static __device__ const LcgSkipTable kPairSkip = make_skip_table();

// Round 6: WHICH pairs a block generates.  A grid-stride loop over the whole tensor wrote 4.7 TB/s whatever the arithmetic cost (the
// step without its stores takes 130 us per 1e8 samples, with them 169); one contiguous chunk of kPairsPerThread * 256 pairs per block,
// walked front to back in 4 KiB rows, writes 5.2 TB/s (153 us; scripts/micro/rng_patterns.hip: 8 / 16 / 32 / 64 pairs per thread
// 162 / 153 / 156 / 154 us, per-wave chunks and a doubling skip-ahead per thread slower) - the write-only pattern that also serves
// k_fill / k_linspace best (profiles/r04_write_patterns.txt).
static constexpr int kPairsPerThread = 16;
template <class T, bool SCALED = false>
__global__ void __launch_bounds__(256) k_rng_normal(unsigned long long state, T* __restrict__ out, size_t n,
                                                    unsigned long long jm, unsigned long long jp, double mu = 0.0, double sigma = 1.0) {
    typedef typename PairOf<T>::type P;
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t npairs = (n + 1) / 2;
    const size_t full = n / 2;  // whole pairs; an odd length adds z0 of one more pair (random.rs:536-540)
    const size_t base = (size_t)blockIdx.x * (256 * kPairsPerThread);
    const size_t lim = base + 256 * kPairsPerThread < full ? base + 256 * kPairsPerThread : full;
    // the state u1 of this thread's first pair is drawn from: 2 base steps to the block's chunk (uniform: scalar unit), 2 t + 1 more by table
    unsigned long long mb, pb;
    lcg_jump(2ULL * base, &mb, &pb);
    unsigned long long x1 = kPairSkip.mp[2 * threadIdx.x] * (mb * state + pb) + kPairSkip.mp[2 * threadIdx.x + 1];
    const bool aligned = (((uintptr_t)out) & (2 * sizeof(T) - 1)) == 0;
    size_t i = base + threadIdx.x;
    const int iters = i < lim ? (int)((lim - i + 255) / 256) : 0;  // (a 32-bit trip count: no 64-bit compare per pair)
#pragma unroll 1
    for (int it = 0; it < iters; ++it, i += 256) {
        const unsigned long long x2 = lcg_step(x1);
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(x2, tb, &sn, &cs);
        double z0 = radius * cs, z1 = radius * sn;
        if (SCALED) {
            z0 = mu + sigma * z0;
            z1 = mu + sigma * z1;
        }
        if (aligned) *(P*)(out + 2 * i) = P{(T)z0, (T)z1};
        else {
            out[2 * i] = (T)z0;
            out[2 * i + 1] = (T)z1;
        }
        x1 = jm * x1 + jp;  // the next row of the chunk: 256 pairs = 512 steps on
    }
    if (i == full && i < npairs && i < base + 256 * kPairsPerThread) {  // n odd: z0 of the last pair
        const unsigned long long x2 = lcg_step(x1);
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(x2, tb, &sn, &cs);
        double z0 = radius * cs;
        if (SCALED) z0 = mu + sigma * z0;
        out[2 * i] = (T)z0;
    }
}

// `stochastic_evolution` (lib.rs:1759-1769; CPU: builtins/stats/random/stochastic_evolution.rs:10-30):
//   for step in 0..steps { z = generate_normal(len); value *= exp(drift + scale * z) }
// Every step draws len normals (whole pairs) from the shared stream, so the pair that feeds elements
// (2i, 2i+1) in step t starts at stream position 2*npairs*t + 2i: the thread skips ahead to 2i once and
// then jumps by 2*npairs per step.  The state stays in registers across all steps -- 16 B of HBM
// traffic per element for the whole time loop instead of (32*steps) B for the materialised plan.
template <class T>
__global__ void __launch_bounds__(256) k_stochastic_evolution(unsigned long long state, const T* __restrict__ in,
                                                              T* __restrict__ out, size_t n, double drift, double scale,
                                                              unsigned steps, unsigned long long gm, unsigned long long gp,
                                                              unsigned long long sm, unsigned long long sp) {
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t npairs = (n + 1) / 2;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= npairs) return;
    unsigned long long s0 = lcg_skip2(state, 512ULL * blockIdx.x, 2ULL * threadIdx.x + 1ULL);  // the state u1 of pair g, step 0, is drawn from
    typedef typename PairOf<T>::type P;
    const bool aligned = ((((uintptr_t)in) | ((uintptr_t)out)) & (2 * sizeof(T) - 1)) == 0;
    for (size_t i = g; i < npairs; i += stride) {
        const bool two = 2 * i + 1 < n;
        double v0, v1 = 0.0;
        if (two && aligned) {
            const P v = *(const P*)(in + 2 * i);
            v0 = (double)v.x;
            v1 = (double)v.y;
        } else {
            v0 = (double)in[2 * i];
            if (two) v1 = (double)in[2 * i + 1];
        }
        unsigned long long s = s0;
        for (unsigned t = 0; t < steps; ++t) {
            const double radius = bm_radius(s, tb);  // as in k_rng_normal: same stream
            double sn, cs;
            bm_sincos(lcg_step(s), tb, &sn, &cs);
            const double t0 = scale * (radius * cs), t1 = scale * (radius * sn);
            v0 = v0 * exp(drift + t0);
            v1 = v1 * exp(drift + t1);
            s = sm * s + sp;  // same pair, next step: 2*npairs draws later
        }
        if (two && aligned) *(P*)(out + 2 * i) = P{(T)v0, (T)v1};
        else {
            out[2 * i] = (T)v0;
            if (two) out[2 * i + 1] = (T)v1;
        }
        s0 = gm * s0 + gp;  // pair i + stride
    }
}

static unsigned rng_grid(const Context* c, size_t work) {
    size_t want = (work + 255) / 256;
    const size_t cap = (size_t)c->num_cus * 8;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

template <class T, int KIND>
static int rng_uniform_any(Context* c, uint64_t state, T* out, size_t n, const UniformXform& xf) {
    if (n == 0) return RMHIP_OK;
    const unsigned grid = rng_grid(c, n);
    unsigned long long jm, jp;
    lcg_jump((unsigned long long)grid * 256ULL, &jm, &jp);
    hipLaunchKernelGGL((k_rng_uniform<T, KIND>), dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, out, n, jm, jp, xf);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_rng_uniform(Context* c, uint64_t state, double* out, size_t n) { return rng_uniform_any<double, 0>(c, state, out, n, UniformXform{}); }
int launch_rng_uniform_f32(Context* c, uint64_t state, float* out, size_t n) { return rng_uniform_any<float, 0>(c, state, out, n, UniformXform{}); }
// a + (b - a) u: the difference is rounded here, once, as the CPU does
int launch_rng_unifrnd(Context* c, uint64_t state, double a, double b, double* out64, float* out32, size_t n) {
    UniformXform xf{};
    xf.a = a;
    xf.d = b - a;
    return out32 ? rng_uniform_any<float, 1>(c, state, out32, n, xf) : rng_uniform_any<double, 1>(c, state, out64, n, xf);
}
int launch_rng_integer_range(Context* c, uint64_t state, long long lower, unsigned long long span, double* out64, float* out32, size_t n) {
    UniformXform xf{};
    xf.lower = lower;
    xf.span = span;
    return out32 ? rng_uniform_any<float, 3>(c, state, out32, n, xf) : rng_uniform_any<double, 3>(c, state, out64, n, xf);
}
template <class T>
static int rng_exponential_any(Context* c, uint64_t state, double mu, T* out, size_t n) {
    if (n == 0) return RMHIP_OK;
    const unsigned grid = rng_grid(c, n);
    unsigned long long jm, jp;
    lcg_jump((unsigned long long)grid * 256ULL, &jm, &jp);
    hipLaunchKernelGGL(k_rng_exponential<T>, dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, out, n, jm, jp, mu);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_rng_exponential(Context* c, uint64_t state, double mu, double* out64, float* out32, size_t n) {
    return out32 ? rng_exponential_any(c, state, mu, out32, n) : rng_exponential_any(c, state, mu, out64, n);
}

template <class T, bool SCALED>
static int rng_normal_any(Context* c, uint64_t state, T* out, size_t n, double mu = 0.0, double sigma = 1.0) {
    if (n == 0) return RMHIP_OK;
    const size_t npairs = (n + 1) / 2, per_block = (size_t)256 * kPairsPerThread;
    if ((npairs + per_block - 1) / per_block > 0x7fffffffULL) return fail(RMHIP_ERR_UNSUPPORTED, "random_normal: tensor too large");
    const unsigned grid = (unsigned)((npairs + per_block - 1) / per_block);  // one contiguous chunk per block
    unsigned long long jm, jp;
    lcg_jump(512ULL, &jm, &jp);  // one 256-pair row
    hipLaunchKernelGGL((k_rng_normal<T, SCALED>), dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, out, n, jm, jp, mu, sigma);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_rng_normal(Context* c, uint64_t state, double* out, size_t n) { return rng_normal_any<double, false>(c, state, out, n); }
int launch_rng_normal_f32(Context* c, uint64_t state, float* out, size_t n) { return rng_normal_any<float, false>(c, state, out, n); }
int launch_rng_normrnd(Context* c, uint64_t state, double mu, double sigma, double* out64, float* out32, size_t n) {
    return out32 ? rng_normal_any<float, true>(c, state, out32, n, mu, sigma) : rng_normal_any<double, true>(c, state, out64, n, mu, sigma);
}

template <class T>
static int stochastic_evolution_any(Context* c, uint64_t state, const T* in, T* out, size_t n, double drift, double scale,
                                    unsigned steps, uint64_t draws_per_step) {
    if (n == 0) return RMHIP_OK;
    const size_t npairs = (n + 1) / 2;
    const unsigned grid = rng_grid(c, npairs);
    unsigned long long gm, gp, sm, sp;
    lcg_jump(2ULL * grid * 256ULL, &gm, &gp);
    lcg_jump(draws_per_step ? (unsigned long long)draws_per_step : 2ULL * (unsigned long long)npairs, &sm, &sp);
    hipLaunchKernelGGL(k_stochastic_evolution<T>, dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, in, out, n,
                       drift, scale, steps, gm, gp, sm, sp);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_stochastic_evolution(Context* c, uint64_t state, const double* in, double* out, size_t n, double drift,
                                double scale, unsigned steps, uint64_t draws_per_step) {
    return stochastic_evolution_any(c, state, in, out, n, drift, scale, steps, draws_per_step);
}
int launch_stochastic_evolution_f32(Context* c, uint64_t state, const float* in, float* out, size_t n, double drift,
                                    double scale, unsigned steps, uint64_t draws_per_step) {
    return stochastic_evolution_any(c, state, in, out, n, drift, scale, steps, draws_per_step);
}

}  // namespace rmhip
