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

namespace rmhip {

static constexpr unsigned long long kMult = 6364136223846793005ULL;
static constexpr unsigned long long kInc = 1ULL;

// (mult, plus) such that advancing `delta` steps is s -> mult*s + plus   (random.rs:238-256)
__host__ __device__ static inline void lcg_jump(unsigned long long delta, unsigned long long* mult,
                                               unsigned long long* plus) {
    unsigned long long cur_mult = kMult, cur_plus = kInc, acc_mult = 1ULL, acc_plus = 0ULL;
    while (delta > 0) {
        if (delta & 1ULL) {
            acc_mult = acc_mult * cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = cur_plus * (cur_mult + 1ULL);
        cur_mult = cur_mult * cur_mult;
        delta >>= 1;
    }
    *mult = acc_mult;
    *plus = acc_plus;
}

uint64_t lcg_advance(uint64_t state, uint64_t delta) {
    unsigned long long m, p;
    lcg_jump(delta, &m, &p);
    return m * state + p;
}

__device__ __forceinline__ double lcg_next_uniform(unsigned long long& s) {
    s = s * kMult + kInc;
    return (double)(s >> 11) * (1.0 / 9007199254740992.0);
}

// T = storage type: double, or float on a precision-32 provider (the f64 stream rounded on store -- what the CPU's
// rand/randn(..., 'single') produce)
template <class T>
__global__ void __launch_bounds__(256) k_rng_uniform(unsigned long long state, T* __restrict__ out, size_t n,
                                                     unsigned long long jm, unsigned long long jp) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= n) return;
    unsigned long long m, p;
    lcg_jump(g, &m, &p);
    unsigned long long s = m * state + p;  // state before element g
    for (size_t i = g; i < n; i += stride) {
        unsigned long long t = s;
        out[i] = (T)lcg_next_uniform(t);
        s = jm * s + jp;  // jump `stride` steps
    }
}

typedef double v2d __attribute__((ext_vector_type(2)));
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

template <class T>
__global__ void __launch_bounds__(256) k_rng_normal(unsigned long long state, T* __restrict__ out, size_t n,
                                                    unsigned long long jm, unsigned long long jp) {
    typedef typename PairOf<T>::type P;
    const size_t npairs = (n + 1) / 2;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= npairs) return;
    unsigned long long m, p;
    lcg_jump(2 * g, &m, &p);
    unsigned long long s = m * state + p;  // state before pair g
    const bool aligned = (((uintptr_t)out) & (2 * sizeof(T) - 1)) == 0;
    for (size_t i = g; i < npairs; i += stride) {
        unsigned long long t = s;
        double u1 = lcg_next_uniform(t);
        if (u1 <= 0.0) u1 = 2.2250738585072014e-308;  // f64::MIN_POSITIVE (random.rs:13,281-283)
        const double u2 = lcg_next_uniform(t);
        const double radius = sqrt(-2.0 * rm_log_pos(u1));  // skel_common.h: < 1 ulp, 40 instead of 102 VALU instructions
        // cos / sin of 2*pi*u2 through sin(pi t), cos(pi t) at t = 2*u2 (rm_sincospi2, skel_common.h): the argument reduction is
        // exact and most of the kernel's VALU work disappears (the kernel is VALU-bound: ~250 fp64 instructions per pair with
        // the library's log / cos / sin, ~145 now).  The CPU evaluates cos(fl(2*pi*u2))
        // (random.rs:284-287); the two differ by the rounding of the angle, <= 4.5e-16 * radius in the result - the size
        // of the libm differences the stream tolerance (8e-14 absolute) already covers.
        double sn, cs;
        rm_sincospi2(2.0 * u2, &sn, &cs);
        const double z0 = radius * cs, z1 = radius * sn;
        if (2 * i + 1 < n) {
            if (aligned) *(P*)(out + 2 * i) = P{(T)z0, (T)z1};
            else {
                out[2 * i] = (T)z0;
                out[2 * i + 1] = (T)z1;
            }
        } else {
            out[2 * i] = (T)z0;  // odd length: z1 of the last pair is dropped (random.rs:536-540)
        }
        s = jm * s + jp;  // jump 2*stride steps
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
    const size_t npairs = (n + 1) / 2;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (g >= npairs) return;
    unsigned long long m, p;
    lcg_jump(2 * g, &m, &p);
    unsigned long long s0 = m * state + p;  // state before pair g of step 0
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
            unsigned long long u = s;
            double u1 = lcg_next_uniform(u);
            if (u1 <= 0.0) u1 = 2.2250738585072014e-308;
            const double u2 = lcg_next_uniform(u);
            const double radius = sqrt(-2.0 * rm_log_pos(u1));  // skel_common.h: < 1 ulp, 40 instead of 102 VALU instructions
            double sn, cs;
            rm_sincospi2(2.0 * u2, &sn, &cs);  // as in k_rng_normal: same stream, exact argument reduction
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

template <class T>
static int rng_uniform_any(Context* c, uint64_t state, T* out, size_t n) {
    if (n == 0) return RMHIP_OK;
    const unsigned grid = rng_grid(c, n);
    unsigned long long jm, jp;
    lcg_jump((unsigned long long)grid * 256ULL, &jm, &jp);
    hipLaunchKernelGGL(k_rng_uniform<T>, dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, out, n, jm, jp);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_rng_uniform(Context* c, uint64_t state, double* out, size_t n) { return rng_uniform_any(c, state, out, n); }
int launch_rng_uniform_f32(Context* c, uint64_t state, float* out, size_t n) { return rng_uniform_any(c, state, out, n); }

template <class T>
static int rng_normal_any(Context* c, uint64_t state, T* out, size_t n) {
    if (n == 0) return RMHIP_OK;
    const unsigned grid = rng_grid(c, (n + 1) / 2);
    unsigned long long jm, jp;
    lcg_jump(2ULL * grid * 256ULL, &jm, &jp);
    hipLaunchKernelGGL(k_rng_normal<T>, dim3(grid), dim3(256), 0, c->stream, (unsigned long long)state, out, n, jm, jp);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_rng_normal(Context* c, uint64_t state, double* out, size_t n) { return rng_normal_any(c, state, out, n); }
int launch_rng_normal_f32(Context* c, uint64_t state, float* out, size_t n) { return rng_normal_any(c, state, out, n); }

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
