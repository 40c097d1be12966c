// pow_tab.h -- pow(w, gamma) for a positive base on the device: the gamma step of image_normalize (special.hip).
// Included inside `namespace rmhip` users and by scripts/micro/pow_accuracy.hip (accuracy against 80-bit powl).
#pragma once
#include "skel_common.h"
#include "pow_tables.h"

namespace rmhip {

// ---- pow(w, gamma) for a positive base: table logarithm -> multiply -> table exponential --------------------------------------
// The gamma step is what makes the apply pass VALU-bound (fp64: 16 lanes per clock and SIMD, profiles/r04_valu_instruction_rates.txt).
// Round 3's rm_pow_pos (skel_common.h) spends ~66 instructions per pixel, 11 of them an IEEE division (f / (2 + f)) and 13 an exp
// polynomial; this form needs ~45 and no division, with three small tables in LDS (scripts/gen_pow_tables.py, 5.1 KiB per workgroup):
//   ln w:  w = 2^e m, m in [1, 2); j = round((m - 1) 128), r = m inv_j - 1 (ONE fma: the 106-bit product minus 1 rounds at 2^-62),
//          ln m = lnc_j + log1p(r) with lnc_j = -ln(inv_j) taken from the STORED inv_j and kept as hi + lo, hi a multiple of 2^-42 so
//          that T = e ln2_hi + lnc_hi is exact; (s, err) = Fast2Sum(T, r) (|T| >= |r| or T = 0: the table spacing guarantees it);
//          lo = err + e ln2_lo + lnc_lo + r^2 q(r), q the degree-5 tail of log1p.  Above sqrt(2) the table holds ln(c_j / 2) and e
//          counts one more, so values near 1 from below do not cancel against -ln 2.
//   g ln w as yh + yl (product error recovered by one fma), refused beyond |yh| > 700 (library pow: overflow / subnormal results).
//   exp:   k = round(yh 128 / ln 2), r = yh - k ln2/128 (hi exact, lo, + yl), |r| <= 0.0027: e^r - 1 by a degree-5 polynomial,
//          2^(k/128) = 2^(k >> 7) T[k & 127] (1 + tail), result = ldexp(T + T (tail + p), k >> 7).
// Error against 80-bit powl over 4e6 bases in (0, 4), 2^-40 .. 2^2 and around 1, gamma in {0.45, 1.8, 2.2, 2.4}
// (scripts/micro/pow_accuracy.hip): <= 1.0 x 2^-53 relative - tighter than rm_pow_pos's (0.45 |g ln w| + 1.5) 2^-53.
typedef double pv2 __attribute__((ext_vector_type(2)));
struct PowTables {
    const pv2* lg;       // [129] (inv_j, lnc_hi_j)
    const double* lglo;  // [129] lnc_lo_j
    const pv2* ex;       // [128] (T_k, tail_k)
};
static constexpr int kPowLdsDoubles = 2 * 129 + 130 + 2 * 128;  // lglo padded to an even count: `ex` stays 16-byte aligned
__device__ __forceinline__ PowTables pow_stage_tables(double* lds, int tid, int nthreads) {
    for (int i = tid; i < 2 * 129; i += nthreads) lds[i] = (&kPowLog[0][0])[i];
    for (int i = tid; i < 129; i += nthreads) lds[2 * 129 + i] = kPowLogLo[i];
    for (int i = tid; i < 2 * 128; i += nthreads) lds[2 * 129 + 130 + i] = (&kPowExp[0][0])[i];
    __syncthreads();
    PowTables t;
    t.lg = reinterpret_cast<const pv2*>(lds);
    t.lglo = lds + 2 * 129;
    t.ex = reinterpret_cast<const pv2*>(lds + 2 * 129 + 130);
    return t;
}
__device__ __forceinline__ double rm_pow_tab(double x, double g, const PowTables& tb) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned hi = (unsigned)(bits >> 32);
    if (hi - 0x00100000u >= 0x7fe00000u) return rm_pow_cold(x, g);  // zero / subnormal / Inf / NaN / negative
    const unsigned mh = hi & 0xfffffu;
    const unsigned jr = mh + 0x1000u;  // round to the nearest 1/128 (may carry into bit 20: j = 128)
    const unsigned j = jr >> 13;
    const int e = (int)(hi >> 20) - 1023 + (j >= (unsigned)RM_POW_LOG_SPLIT ? 1 : 0);
    const double m = __longlong_as_double((long long)(((unsigned long long)(mh | 0x3ff00000u) << 32) | (bits & 0xffffffffull)));
    const pv2 te = tb.lg[j];
    const double lnc_lo = tb.lglo[j];
    const double r = __builtin_fma(m, te.x, -1.0);
    const double dk = (double)e;
    const double T = __builtin_fma(dk, 0x1.62e42fee00000p-1, te.y);  // exact: multiples of 2^-42 below 2^10
    const double s = T + r;
    const double err = (T - s) + r;
    double q = __builtin_fma(r, 0x1.2492492492492p-3, -0x1.5555555555555p-3);  // 1/7, -1/6
    q = __builtin_fma(r, q, 0x1.999999999999ap-3);                             // 1/5
    q = __builtin_fma(r, q, -0.25);
    q = __builtin_fma(r, q, 0x1.5555555555555p-2);                             // 1/3
    q = __builtin_fma(r, q, -0.5);
    double lo = __builtin_fma(dk, 0x1.a39ef35793c76p-33, lnc_lo);
    lo = __builtin_fma(r * r, q, lo) + err;
    const double yh = g * s;
    if (!(__builtin_fabs(yh) <= 700.0)) return rm_pow_cold(x, g);
    const double yl = __builtin_fma(g, s, -yh) + g * lo;
    const double kd = __builtin_rint(yh * 0x1.71547652b82fep+7);  // 128 / ln 2
    double rr = __builtin_fma(kd, -0x1.62e42fee00000p-8, yh);     // exact
    rr = __builtin_fma(kd, -0x1.a39ef35793c76p-40, rr);
    rr += yl;
    double p = __builtin_fma(rr, 0x1.1111111111111p-7, 0x1.5555555555555p-5);  // 1/120, 1/24
    p = __builtin_fma(rr, p, 0x1.5555555555555p-3);                            // 1/6
    p = __builtin_fma(rr, p, 0.5);
    p = __builtin_fma(rr * rr, p, rr);
    const int k = (int)kd;
    const pv2 ex = tb.ex[k & 127];
    const double res = __builtin_fma(ex.x, ex.y + p, ex.x);
    return __builtin_amdgcn_ldexp(res, k >> 7);
}

}  // namespace rmhip
