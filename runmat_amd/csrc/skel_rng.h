// skel_rng.h -- the CPU-parity random stream's device functions (LCG skip-ahead, the table-based Box-Muller step), shared by the
// ahead-of-time kernels of rng.hip and - embedded as text, like skel_common.h - by the hipRTC-generated fused elementwise kernels
// that consume a LAZY `random_normal` operand (codegen.cpp: the normals are generated in registers by the consuming kernel, bit for
// bit what k_rng_normal would have written).  Needs rng_tables.h before it.  No `#include`, no `#pragma once`: hipRTC sees this inline.
#ifndef RMHIP_SKEL_RNG
#define RMHIP_SKEL_RNG

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

// State `delta_block + delta_thread` steps after `state`: the block part is uniform (scalar unit), the thread part is at most nine
// doublings - a thread's one-off skip-ahead was 11 % of the normal kernel's VALU time as a single 20-step loop.
__device__ __forceinline__ unsigned long long lcg_skip2(unsigned long long state, unsigned long long delta_block,
                                                        unsigned long long delta_thread) {
    unsigned long long mb, pb, mt, pt;
    lcg_jump(delta_block, &mb, &pb);
    lcg_jump(delta_thread, &mt, &pt);
    return mt * (mb * state + pb) + pt;
}
__device__ __forceinline__ unsigned long long lcg_step(unsigned long long s) { return s * kMult + kInc; }
// (s >> 11) as a double, exactly (53 bits): two conversions and one fma
__device__ __forceinline__ double lcg_bits53(unsigned long long s) {
    // 32-bit pieces throughout (a 64-bit integer -> double conversion is expanded into four instructions even when the value is known
    // to be small), and no v_cvt_f64_u32 either (8 cycles each, profiles/r04_valu_instruction_rates.txt): the pieces are dropped into
    // the mantissas of 2^84 and 2^52 and the offsets subtracted - two exact additions
    const unsigned s_hi = (unsigned)(s >> 32), s_lo = (unsigned)s;
    const unsigned hi = s_hi >> 11, lo = (s_hi << 21) | (s_lo >> 11);
    const double a = __longlong_as_double((long long)(((unsigned long long)0x45300000u << 32) | hi));   // 2^84 + hi 2^32 (unit 2^32)
    const double b = __longlong_as_double((long long)(((unsigned long long)0x43300000u << 32) | lo));   // 2^52 + lo
    return (a - 0x1.00000001p+84) + b;  // (hi 2^32 - 2^52) + (2^52 + lo), both steps exact
}
__device__ __forceinline__ double lcg_uniform(unsigned long long s) { return lcg_bits53(s) * (1.0 / 9007199254740992.0); }

typedef double v2d __attribute__((ext_vector_type(2)));
// ---- Box-Muller step (random.rs:279-288): z0 = r cos(2 pi u2), z1 = r sin(2 pi u2), r = sqrt(-2 ln u1) ----
// The generator is VALU-bound, not HBM-bound (8 B written per sample): with the library's log / sqrt / cos / sin a pair costs ~250
// fp64 instructions; round 2's short logarithm and sincospi brought it to ~145 (0.256 ms per 1e8 normals, 3.1 TB/s).  This form
// is built around two small tables in LDS (rng_tables.h, 10 KiB per workgroup) and needs ~100:
//   ln u1:  u1 = v 2^-53 with v the 53-bit integer; its double has mantissa m in [1, 2) and exponent e.  j = round((m - 1) 128),
//           r = m inv_j - 1 (one fma, |r| <= 2^-8), ln m = lnc_j + log1p(r) with a degree-7 Taylor polynomial.  From
//           c_j > sqrt(2) on the table holds ln(c_j / 2) and e counts one more: u1 -> 1 gives lnc = 0 and the relative accuracy of
//           log1p(r) alone (no cancellation against e ln 2) - the radius near 0 needs it.  No division (the short logarithm's
//           f / (2 + f) was 12 instructions), no frexp.
//   sqrt:   v_rsq_f64 and one coupled Newton step + one correction; the argument is in [2.2e-16, 1417], so none of the library's
//           rescaling.
//   angle:  t = 2 u2 = v 2^-52 in [0, 2); j = round(256 t), d = t - j / 256 (exact), x = pi d, |x| <= pi / 512;
//           sin(pi t) = S_j cos x + C_j sin x, cos(pi t) = C_j cos x - S_j sin x with (S_j, C_j) = (sin, cos)(pi j / 256) over the
//           whole circle (513 entries: no quadrant logic, no swaps, no sign fix-ups) and degree-5 / degree-6 Taylor polynomials.
//           cos x is kept as 1 + w and the result formed as S_j + (S_j w + C_j sin x): the table value enters unrounded.
// Errors against 80-bit references over 2e6 draws (scripts/rng_accuracy.py): ln <= 1 ulp, radius <= 1.5 ulp, sin / cos <= 1.6e-16
// absolute; the CPU's libm differs from the exact values by as much, and tests/test_gpu_parity.py holds the stream to 8e-14.
struct BmTables {
    const v2d* sc;  // [513] (sin, cos)(pi j / 256)
    const v2d* lg;  // [129] (inv_j, lnc_j)
};
static constexpr int kBmLdsDoubles = 2 * (513 + 129);

__device__ __forceinline__ BmTables bm_stage_tables(double* lds, int tid, int nthreads) {
    const double* src_sc = &kSinCosPi[0][0];
    const double* src_lg = &kLogTab[0][0];
    for (int i = tid; i < 2 * 513; i += nthreads) lds[i] = src_sc[i];
    for (int i = tid; i < 2 * 129; i += nthreads) lds[2 * 513 + i] = src_lg[i];
    __syncthreads();
    BmTables t;
    t.sc = reinterpret_cast<const v2d*>(lds);
    t.lg = reinterpret_cast<const v2d*>(lds + 2 * 513);
    return t;
}

__device__ __forceinline__ double bm_radius(unsigned long long x1, const BmTables& tb, double* ln_out = nullptr) {
    const double vd = lcg_bits53(x1);  // u1 2^53
    const unsigned long long bits = (unsigned long long)__double_as_longlong(vd);
    const unsigned hi = (unsigned)(bits >> 32);
    const unsigned mh = hi & 0xfffffu;
    const unsigned jr = mh + 0x1000u;                       // round to the nearest 1/128 (may carry into bit 20: j = 128)
    const unsigned j = jr >> 13;
    // exponent of u1 (+ 1 above sqrt(2)); the comparison as bit 31 of a sum (jr < 2^21): three 32-bit integer ops, no vcc round trip
    const int e = (int)(hi >> 20) - 1076 + (int)((jr + (0x80000000u - ((unsigned)RM_LOG_SPLIT << 13))) >> 31);
    const double m = __longlong_as_double((long long)(((unsigned long long)(mh | 0x3ff00000u) << 32) | (bits & 0xffffffffull)));
    const v2d te = tb.lg[j];
    const double r = __builtin_fma(m, te.x, -1.0);
    double p = __builtin_fma(r, 0x1.2492492492492p-3, -0x1.5555555555555p-3);  // 1/7, -1/6
    p = __builtin_fma(r, p, 0x1.999999999999ap-3);                             // 1/5
    p = __builtin_fma(r, p, -0.25);
    p = __builtin_fma(r, p, 0x1.5555555555555p-2);                             // 1/3
    p = __builtin_fma(r, p, -0.5);
    const double lp = __builtin_fma(r * r, p, r);                              // log1p(r)
    // e in [-1076, 1] as a double without v_cvt_f64_i32 (8 cycles): e + 2048 sits in the low word of 2^52's mantissa
    const double dk = __longlong_as_double((long long)(((unsigned long long)0x43300000u << 32) | (unsigned)(e + 2048))) - 0x1.00000000008p+52;
    const double head = __builtin_fma(dk, 0x1.62e42fee00000p-1, te.y);         // e ln2_hi is exact (32-bit constant)
    const double tail = __builtin_fma(dk, 0x1.a39ef35793c76p-33, lp);
    if (ln_out) *ln_out = (x1 >> 11) == 0 ? -0x1.6232bdd7abcd2p+9 : head + tail;  // ln u1 (u1 = 0 stands for f64::MIN_POSITIVE: ln 2^-1022)
    const double x = -2.0 * (head + tail);                                     // -2 ln u1 in [2.2e-16, 1417]
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double c = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, c, g);
    const double d = __builtin_fma(-g, g, x);  // (h = y / 2 is good to 2^-26: enough for the correction term d h, itself <= 2^-50 g)
    g = __builtin_fma(d, h, g);
    // u1 = 0 is replaced by f64::MIN_POSITIVE (random.rs:13,281-283): sqrt(-2 ln 2^-1022).  One draw in 2^53: tested for the whole wave
    // (a compare and a scalar branch) instead of two v_cndmask on vcc per pair - that form costs 23 cycles per instruction on gfx950
    // (profiles/r04_valu_instruction_rates.txt)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64((x1 >> 11) == 0) != 0, 0)) return (x1 >> 11) == 0 ? 0x1.2d1f5a276d140p+5 : g;
    return g;
}

__device__ __forceinline__ void bm_sincos(unsigned long long x2, const BmTables& tb, double* sn, double* cs) {
    // t = 2 u2 = v 2^-52 with v = x2 >> 11 (53 bits); j = round(256 t) = round(v 2^-44), d = v - j 2^44 in [-2^43, 2^43] - all in
    // 32-bit integer arithmetic on the upper word (the lower one only decides exact ties), and d enters the floating-point unit by one
    // addition: its two words are laid over the mantissa of 1.5 2^52
    const unsigned s_hi = (unsigned)(x2 >> 32), s_lo = (unsigned)x2;
    const unsigned hi = s_hi >> 11, lo = (s_hi << 21) | (s_lo >> 11);
    const unsigned tmp = hi + 0x800u;
    const unsigned j = tmp >> 12;                                   // 0 .. 512
    const unsigned dh = (tmp & 0xfffu) + (0x43380000u - 0x800u);    // upper word of 1.5 2^52 + d
    const double dv = __longlong_as_double((long long)(((unsigned long long)dh << 32) | lo)) - 0x1.8p+52;  // (t - j / 256) 2^52, exact
    const double x = dv * 0x1.921fb54442d18p-51;                   // pi 2^-52
    const v2d te = tb.sc[j];
    const double z = x * x;
    const double ps = __builtin_fma(z, 0x1.1111111111111p-7, -0x1.5555555555555p-3);  // 1/120, -1/6
    const double sd = __builtin_fma(x * z, ps, x);                 // sin x
    double w = __builtin_fma(z, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5);        // -1/720, 1/24
    w = __builtin_fma(z, w, -0.5);
    w = z * w;                                                     // cos x - 1
    // S_j (1 + w) + C_j sin x and C_j (1 + w) - S_j sin x, two fmas each (round 6: was mul + fma + add with the table value entering
    // unrounded - 1.6e-16 absolute; this form 2.3e-16, inside the 6e-16-of-the-radius bound the tests state)
    *sn = __builtin_fma(te.y, sd, __builtin_fma(te.x, w, te.x));
    *cs = __builtin_fma(-te.x, sd, __builtin_fma(te.y, w, te.y));
}

#endif  // RMHIP_SKEL_RNG
