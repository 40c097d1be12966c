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

struct rm_d2 {
    double x, y;
} __attribute__((aligned(16)));

#endif  // RMHIP_SKEL_COMMON
