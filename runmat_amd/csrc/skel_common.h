// skel_common.h -- device helpers shared by the ahead-of-time kernels (*.hip) and the hipRTC
// generated fused kernels (this file is also embedded as a string, see Makefile `*_str.inc`).
// Scalar semantics follow the reference CPU builtins, not WGSL builtins:
//   sign      crates/runmat-runtime/src/builtins/math/elementwise/sign.rs:236-246
//   max / min crates/runmat-runtime/src/builtins/math/reduction/max.rs:2323-2344,1715-1728
//             / min.rs:1519-1531  (Include-NaN: any NaN operand => NaN; -0 < +0)
//   round     Rust f64::round == C round(): half away from zero (rounding/round.rs:305)
// No `#include` and no `#pragma once` here: hipRTC sees this text inline.
#ifndef RMHIP_SKEL_COMMON
#define RMHIP_SKEL_COMMON

typedef unsigned long long rm_u64;

__device__ __forceinline__ double rm_nan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ bool rm_isnan(double x) { return x != x; }
__device__ __forceinline__ bool rm_isinf(double x) {
    return __builtin_fabs(x) == __builtin_inf();
}
__device__ __forceinline__ bool rm_isfinite(double x) {
    return __builtin_fabs(x) < __builtin_inf();
}
__device__ __forceinline__ double rm_f32(double x) { return (double)(float)x; }
__device__ __forceinline__ double rm_sign(double x) {
    return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : (x == 0.0 ? 0.0 : x));
}
__device__ __forceinline__ double rm_max(double a, double b) {
    if (a != a || b != b) return rm_nan();
    if (b > a) return b;
    if (b < a) return a;
    if (b == 0.0 && a == 0.0) return (!__builtin_signbit(b) && __builtin_signbit(a)) ? b : a;
    return a;
}
__device__ __forceinline__ double rm_min(double a, double b) {
    if (a != a || b != b) return rm_nan();
    if (b < a) return b;
    if (b > a) return a;
    if (b == 0.0 && a == 0.0) return (__builtin_signbit(b) && !__builtin_signbit(a)) ? b : a;
    return a;
}
__device__ __forceinline__ double rm_heaviside(double v) {
    if (v != v) return v;
    return v > 0.0 ? 1.0 : (v == 0.0 ? 0.5 : 0.0);
}
// mod / rem: the select chains of crates/runmat-accelerate/src/fusion.rs:2954-2970
__device__ __forceinline__ double rm_mod(double l, double r) {
    if (rm_isinf(r) && rm_isfinite(l)) return (l == 0.0 || rm_sign(l) == rm_sign(r)) ? l : r;
    return l - r * floor(l / r);
}
__device__ __forceinline__ double rm_rem(double l, double r) {
    if (rm_isinf(r) && rm_isfinite(l)) return l;
    return l - r * trunc(l / r);
}

// sinc: crates/runmat-runtime/src/builtins/math/.../sinc.rs:302-311
__device__ __forceinline__ double rm_sinc(double v) {
    if (v == 0.0) return 1.0;
    if (rm_isfinite(v) && v == trunc(v)) return 0.0;
    const double scaled = 3.14159265358979323846 * v;
    return sin(scaled) / scaled;
}

// sin / cos for kernels whose results are stored as f32 (RM_RESULT_F32, set by the generator for a precision-32
// provider).  The library sin spends ~60 fp64 instructions per call on a double-double argument reduction and on
// polynomial tails that only matter for the last bits of an f64 result; with four of them per lane the f32 fused kernel
// was VALU- and power-bound (169 -> 245 us per 8192^2 dispatch as the clock dropped).  Here: one-step Cody-Waite with
// fma (n*PIO2_HI is exact inside the fma and the difference is a multiple of 2^-52 below 1, so r is exact; PIO2_LO adds
// the next 53 bits), then the degree-13 / degree-14 minimax polynomials on [-pi/4, pi/4] (fdlibm's published
// coefficients) without correction tails: ~30 instructions, error < 1.5 ulp of f64, i.e. 2^-28 of an f32 ulp - the
// stored value is the correctly rounded f32 except when the exact result lies within that distance of a rounding
// boundary.  |x| >= 2^20, Inf and NaN take the library path.
// Horner step d = a*b + c with the coefficient as the (single allowed) scalar operand of v_fma_f64.  Written out because
// the compiler otherwise keeps every coefficient in a VGPR pair and copies it into the accumulator of a v_fmac_f64
// first (one v_mov_b64 per step, also in the library's own sin): a third of the VALU work of the function.
#define RM_FMA_SC(d, a, b, c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c))
__device__ __forceinline__ double rm_sincos_r32(double x, int shift) {
    const double ax = __builtin_fabs(x);  // odd symmetry is applied on the bits at the end: keeps sin(-0) = -0
    if (!(ax < 1048576.0)) return shift ? cos(x) : sin(x);
    const double n = __builtin_rint(ax * 0x1.45f306dc9c883p-1);
    double r = __builtin_fma(n, -0x1.921fb54442d18p+0, ax);
    r = __builtin_fma(n, -0x1.1a62633145c07p-54, r);
    const unsigned q = (unsigned)(int)n + (unsigned)shift;
    const double z = r * r;
    double ps = __builtin_fma(z, 0x1.5d93a5acfd57cp-33, -0x1.ae5e68a2b9cebp-26);
    RM_FMA_SC(ps, z, ps, 0x1.71de357b1fe7dp-19);
    RM_FMA_SC(ps, z, ps, -0x1.a01a019c161d5p-13);
    RM_FMA_SC(ps, z, ps, 0x1.111111110f8a6p-7);
    RM_FMA_SC(ps, z, ps, -0x1.5555555555549p-3);
    const double sn = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -0x1.8fae9be8838d4p-37, 0x1.1ee9ebdb4b1c4p-29);
    RM_FMA_SC(pc, z, pc, -0x1.27e4f809c52adp-22);
    RM_FMA_SC(pc, z, pc, 0x1.a01a019cb1590p-16);
    RM_FMA_SC(pc, z, pc, -0x1.6c16c16c15177p-10);
    RM_FMA_SC(pc, z, pc, 0x1.555555555554cp-5);
    pc = __builtin_fma(z, pc, -0.5);
    const double cs = __builtin_fma(z, pc, 1.0);
    const double v = (q & 1u) ? cs : sn;
    unsigned flip = (q & 2u) << 30;
    if (!shift) flip ^= (unsigned)((rm_u64)__double_as_longlong(x) >> 32) & 0x80000000u;
    return __longlong_as_double(__double_as_longlong(v) ^ (long long)((rm_u64)flip << 32));
}

// Natural logarithm for the kernels that are VALU-bound on it (Box-Muller radius of randn / stochastic_evolution, the gamma
// step of image_normalize): the library log is 102 VALU instructions on gfx950, this one ~40 - frexp, one division, the
// two interleaved Horner chains of fdlibm's log (its published coefficients Lg1..Lg7 and the ln2 split).  Error < 1 ulp
// (0.86 measured against logl on 2e8 arguments: uniforms in (0,1) and random positive doubles including denormals).
// Zero, negative, infinite and NaN arguments take the library path.
__device__ __forceinline__ double rm_log_pos(double x) {
    if (!(x > 0.0 && x < __builtin_inf())) return log(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0x1.6a09e667f3bcdp-1) {  // sqrt(1/2): keep 1 + f in [sqrt(1/2), sqrt(2))
        m = m * 2.0;
        e -= 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    double t1 = __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3);  // Lg6, Lg4
    RM_FMA_SC(t1, w, t1, 0x1.999999997fa04p-2);                                 // Lg2
    t1 = w * t1;
    double t2 = __builtin_fma(w, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3);  // Lg7, Lg5
    RM_FMA_SC(t2, w, t2, 0x1.2492494229359p-2);                                 // Lg3
    RM_FMA_SC(t2, w, t2, 0x1.5555555555593p-1);                                 // Lg1
    t2 = z * t2;
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 0x1.62e42fee00000p-1 - ((hfsq - (s * (hfsq + R) + dk * 0x1.a39ef35793c76p-33)) - f);  // ln2_hi, ln2_lo
}


// x^g for the gamma step of image_normalize (x > 0 finite; anything else, and |g ln x| > 32, takes the library pow).  One pass:
// the logarithm above kept as a two-piece value L + l (the sum it forms, renormalised - two more additions), y = g (L + l) with
// the product's rounding error recovered by an fma, then exp(y) = 2^n exp(r), n = rint(y / ln 2), r = y - n ln2_hi - n ln2_lo + the
// low part, exp(r) as the degree-13 Taylor polynomial on |r| <= 0.35 (truncation 4e-18).  ~55 VALU instructions against ~80 for
// this log followed by the library exp (whose special-case ladders the range check here replaces) and ~200 for the library pow.
// Error: (0.45 |g ln x| + 1.5) units of 2^-53 relative - the logarithm's own sub-ulp error is what the exponent multiplies;
// measured max 14.9 units (1.7e-15) over 3e7 (x, g) with |g ln x| <= 32 against powl, 2-3 units on image data (x in (0, 4], g in
// [0.2, 3.2]).  The CPU's powf is < 1 ulp; tests/test_gpu_parity.py states 4e-15.
// pow as the fused kernels and the per-op kernel call it.  An exponent of exactly 2 - `x.^2`, the common case by far - is the exact
// product rounded once (what fdlibm-derived libms return for it, and within half an ulp of any other); everything else is the
// library's.  In the 14-op chain of benchmarks/elementwise-math the library pow was a third of the kernel's instructions.
__device__ __forceinline__ double rm_pow(double x, double y) {
    if (y == 2.0) return x * x;
    return pow(x, y);
}
// the library pow, out of line: it is the cold path of rm_pow_pos (non-positive / non-finite bases, |g ln x| > 32) and inlining its
// special-case ladders into the caller costs the hot path ~20 registers
__device__ __noinline__ double rm_pow_cold(double x, double g) { return pow(x, g); }
__device__ __forceinline__ double rm_pow_pos(double x, double g) {
    if (!(x > 0.0 && x < __builtin_inf())) return rm_pow_cold(x, g);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0x1.6a09e667f3bcdp-1) {  // sqrt(1/2): keep 1 + f in [sqrt(1/2), sqrt(2))
        m = m * 2.0;
        e -= 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    double t1 = __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3);  // Lg6, Lg4
    RM_FMA_SC(t1, w, t1, 0x1.999999997fa04p-2);                                 // Lg2
    t1 = w * t1;
    double t2 = __builtin_fma(w, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3);  // Lg7, Lg5
    RM_FMA_SC(t2, w, t2, 0x1.2492494229359p-2);                                 // Lg3
    RM_FMA_SC(t2, w, t2, 0x1.5555555555593p-1);                                 // Lg1
    t2 = z * t2;
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double A = dk * 0x1.62e42fee00000p-1;  // exact: ln2_hi has 21 trailing zero bits
    const double hi = A + f;                     // |A| >= 0.69 > |f| or A == 0: the error of this sum is f - (hi - A)
    const double lo = (f - (hi - A)) - (hfsq - (s * (hfsq + R) + dk * 0x1.a39ef35793c76p-33));
    const double L = hi + lo;
    const double l = lo - (L - hi);
    const double yh = g * L;
    if (!(__builtin_fabs(yh) <= 32.0)) return rm_pow_cold(x, g);
    const double yl = __builtin_fma(g, L, -yh) + g * l;
    const double n = __builtin_rint(yh * 0x1.71547652b82fep+0);
    double r = __builtin_fma(n, -0x1.62e42fee00000p-1, yh);
    r = __builtin_fma(n, -0x1.a39ef35793c76p-33, r);
    r += yl;
    double p = __builtin_fma(r, 0x1.6124613a86d09p-33, 0x1.1eed8eff8d898p-29);  // 1/13!, 1/12!
    RM_FMA_SC(p, p, r, 0x1.ae64567f544e4p-26);                                   // 1/11!
    RM_FMA_SC(p, p, r, 0x1.27e4fb7789f5cp-22);
    RM_FMA_SC(p, p, r, 0x1.71de3a556c734p-19);
    RM_FMA_SC(p, p, r, 0x1.a01a01a01a01ap-16);
    RM_FMA_SC(p, p, r, 0x1.a01a01a01a01ap-13);
    RM_FMA_SC(p, p, r, 0x1.6c16c16c16c17p-10);
    RM_FMA_SC(p, p, r, 0x1.1111111111111p-7);
    RM_FMA_SC(p, p, r, 0x1.5555555555555p-5);
    RM_FMA_SC(p, p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);  // |n| <= 47
}

// sin(pi t), cos(pi t) for the Box-Muller angle (t = 2 u in [0, 2); any |t| < 2^51 works): quarter-turn reduction in t itself
// (exact: n = rint(2t), r = t - n/2 by one fma, |r| <= 1/4), x = pi r with a two-term pi, then the same two minimax
// polynomials as rm_sincos_r32 and a rotation by n quarter turns.  ~30 VALU instructions against the library sincospi's 75;
// absolute error <= 1.4e-16 (measured against sinl / cosl on 1e8 uniforms).
__device__ __forceinline__ void rm_sincospi2(double t, double* sp, double* cp) {
    const double n = __builtin_rint(2.0 * t);
    const double r = __builtin_fma(n, -0.5, t);
    const double x = __builtin_fma(r, 0x1.921fb54442d18p+1, r * 0x1.1a62633145c07p-53);
    const double z = x * x;
    double ps = __builtin_fma(z, 0x1.5d93a5acfd57cp-33, -0x1.ae5e68a2b9cebp-26);
    RM_FMA_SC(ps, z, ps, 0x1.71de357b1fe7dp-19);
    RM_FMA_SC(ps, z, ps, -0x1.a01a019c161d5p-13);
    RM_FMA_SC(ps, z, ps, 0x1.111111110f8a6p-7);
    RM_FMA_SC(ps, z, ps, -0x1.5555555555549p-3);
    const double sn = __builtin_fma(x * z, ps, x);
    double pc = __builtin_fma(z, -0x1.8fae9be8838d4p-37, 0x1.1ee9ebdb4b1c4p-29);
    RM_FMA_SC(pc, z, pc, -0x1.27e4f809c52adp-22);
    RM_FMA_SC(pc, z, pc, 0x1.a01a019cb1590p-16);
    RM_FMA_SC(pc, z, pc, -0x1.6c16c16c15177p-10);
    RM_FMA_SC(pc, z, pc, 0x1.555555555554cp-5);
    pc = __builtin_fma(z, pc, -0.5);
    const double cs = __builtin_fma(z, pc, 1.0);
    const int q = (int)n & 3;
    const double s = (q & 1) ? cs : sn, c = (q & 1) ? sn : cs;
    *sp = (q >= 2) ? -s : s;
    *cp = (q == 1 || q == 2) ? -c : c;
}

#ifdef RM_RESULT_F32
__device__ __forceinline__ double rm_sin(double x) { return rm_sincos_r32(x, 0); }
__device__ __forceinline__ double rm_cos(double x) { return rm_sincos_r32(x, 1); }
#else
__device__ __forceinline__ double rm_sin(double x) { return sin(x); }
__device__ __forceinline__ double rm_cos(double x) { return cos(x); }
#endif

struct rm_d2 {
    double x, y;
} __attribute__((aligned(16)));

#endif  // RMHIP_SKEL_COMMON
