/*
 * oracle.c -- CPU restatement of the reference (runmat-org/runmat v0.6.1) dense-array hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so; the product path (runmat_amd/, librmhip.so)
 * never links, imports or calls it.
 *
 * Every function restates, loop for loop, the reference's single-threaded CPU ("semantic
 * baseline") implementation and cites the reference file:line it follows (paths relative to the
 * reference root). Column-major storage, f64 everywhere (crates/runmat-builtins/src/lib.rs:73-118).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - matmul / elementwise / broadcast / sum / mean / LCG-uniform / LU: pinned against the
 *     reference's own KATs and sequence definitions (tests/test_oracle_kats.py, tests/golden/).
 *   - transcendental maps: the reference calls Rust f64::{sin,cos,exp,ln,tanh,powf,...} which
 *     lower to the platform libm; no golden bits exist in the reference => stated ulp tolerance.
 *   - mldivide: the reference calls nalgebra 0.32.6 SVD::solve (third-party, absent from
 *     /root/reference); restated here with a one-sided Jacobi SVD pseudo-inverse solve; the
 *     reference pins it only by residual norms (mldivide.rs:662-696) => bit-level "parity
 *     unpinned", residual/forward-error parity only.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math, no OpenMP).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * matmul  -- crates/runmat-runtime/src/builtins/common/linalg.rs:6-32 (matmul_real)
 *            (identical loop in crates/runmat-accelerate/src/simple_provider.rs:7698-7741)
 * C[i + j*rows] = sum_k a[i + k*rows] * b[k + j*brows], k ascending, separate mul and add.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_matmul(const double* a, size_t arows, size_t acols, const double* b, size_t brows,
                       size_t bcols, double* out) {
    if (acols != brows) return 1; /* "Inner matrix dimensions must agree" linalg.rs:7-15 */
    for (size_t j = 0; j < bcols; ++j) {
        for (size_t i = 0; i < arows; ++i) {
            double sum = 0.0;
            for (size_t k = 0; k < acols; ++k) {
                sum += a[i + k * arows] * b[k + j * brows];
            }
            out[i + j * arows] = sum;
        }
    }
    return 0;
}

/* matmul_epilogue -- crates/runmat-accelerate/src/simple_provider.rs:7743-7846 (the reference
 * provider's statement of `MatmulEpilogue`, lib.rs:3498-3560): plain matmul, then per element
 * v = v*alpha + beta; row scale; col scale; clamp_min (f64::max); clamp_max (f64::min); powf; diag copy.
 * NULL pointers / has_* == 0 mean "absent". */
ORC_API int orc_matmul_epilogue(const double* a, size_t arows, size_t acols, const double* b, size_t brows,
                                size_t bcols, double alpha, double beta, const double* row_scale, int row_div,
                                const double* col_scale, int col_div, int has_min, double clamp_min, int has_max,
                                double clamp_max, int has_pow, double pow_exp, double* diag, double* out) {
    int rc = orc_matmul(a, arows, acols, b, brows, bcols, out);
    if (rc) return rc;
    for (size_t j = 0; j < bcols; ++j) {
        for (size_t i = 0; i < arows; ++i) {
            double v = out[i + j * arows] * alpha + beta;
            if (row_scale) v = row_div ? v / row_scale[i] : v * row_scale[i];
            if (col_scale) v = col_div ? v / col_scale[j] : v * col_scale[j];
            if (has_min) v = fmax(v, clamp_min);
            if (has_max) v = fmin(v, clamp_max);
            if (has_pow) v = pow(v, pow_exp);
            if (diag && i == j) diag[i] = v;
            out[i + j * arows] = v;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Unary maps -- e.g. crates/runmat-runtime/src/builtins/math/trigonometry/sin.rs:245-255
 * (`tensor.data.iter().map(|&v| v.sin()).collect()`): one libm call per element.
 * Op codes are the oracle's own (shared with tests via oracle.py); they follow the unary op
 * vocabulary of crates/runmat-accelerate/src/fusion.rs:2932-3026.
 * ---------------------------------------------------------------------------------------- */
enum {
    ORC_SIN = 0, ORC_COS, ORC_TAN, ORC_ASIN, ORC_ACOS, ORC_ATAN, ORC_SINH, ORC_COSH, ORC_TANH,
    ORC_ASINH, ORC_ACOSH, ORC_ATANH, ORC_EXP, ORC_EXPM1, ORC_LOG, ORC_LOG2, ORC_LOG10, ORC_LOG1P,
    ORC_SQRT, ORC_ABS, ORC_SIGN, ORC_FLOOR, ORC_CEIL, ORC_ROUND, ORC_FIX, ORC_NEG, ORC_EXP2,
    ORC_HEAVISIDE, ORC_ISNAN, ORC_ISINF, ORC_ISFINITE, ORC_UPLUS, ORC_SINGLE, ORC_DOUBLE, ORC_ERF, ORC_SINC, ORC_NOT,
    ORC_GAMMA, ORC_FACTORIAL, ORC_NEXTPOW2, ORC_GAMMALN, ORC_ERFCINV, ORC_NAN_TO_ZERO, ORC_NOT_NAN, ORC_UNARY_COUNT
};

/* ---- special functions: crates/runmat-runtime/src/builtins/math/elementwise/{gamma,gammaln,factorial,nextpow2,
 * erfcinv}.rs.  gamma() runs in num-complex arithmetic there even for real arguments (gamma.rs:289-343); with zero
 * imaginary parts every complex product is the real product, `powc` is powf (from_polar(r^w, 0)), and a complex
 * quotient (c+0i)/(d+0i) evaluates (c*d)/(d*d) (num-complex 0.4 Div: re = (a.re*b.re + a.im*b.im) / b.norm_sqr()),
 * which is what is written here.  erfc is libm's (the crate `libm`, a port of the same FreeBSD msun code as glibc's). */
static const double LANCZOS_COEFFS[8] = { /* gamma.rs:30-39, gammaln.rs:30-39 */
    676.5203681218851, -1259.1392167224028, 771.3234287776531, -176.6150291621406,
    12.507343278686905, -0.13857109526572012, 9.984369578019572e-6, 1.5056327351493116e-7};

static int close_to_integer(double x) { /* gamma.rs:353-364 */
    if (!isfinite(x)) return 0;
    double nearest = round(x), diff = fabs(x - nearest);
    if (nearest == 0.0) return diff <= 1e-12 * 1e-12;
    return diff <= 1e-12 * fmax(fabs(nearest), 1.0);
}
static double lanczos_gamma(double z) { /* gamma.rs:332-343 */
    double zm1 = z - 1.0, sum = 0.9999999999998099;
    for (int i = 0; i < 8; ++i) {
        double d = zm1 + (double)(i + 1);
        sum += (LANCZOS_COEFFS[i] * d) / (d * d);
    }
    double t = zm1 + (7.0 + 0.5);
    return 2.5066282746310005 * pow(t, zm1 + 0.5) * exp(-t) * sum;
}
static double gamma_real_scalar(double x) { /* gamma.rs:289-330 */
    if (isnan(x)) return NAN;
    if (isinf(x)) return x > 0.0 ? INFINITY : NAN;
    if (x <= 0.0 && close_to_integer(x)) return INFINITY;
    if (x < 0.5) {
        double s = sin(M_PI * x);
        if (s * s <= 1e-12 * 1e-12) return INFINITY;
        double d = s * lanczos_gamma(1.0 - x);
        return (M_PI * d) / (d * d);
    }
    return lanczos_gamma(x);
}
static double lanczos_gammaln(double v) { /* gammaln.rs:273-281 */
    double zm1 = v - 1.0, sum = 0.9999999999998099;
    for (int i = 0; i < 8; ++i) sum += LANCZOS_COEFFS[i] / (zm1 + (double)(i + 1));
    double t = zm1 + 7.0 + 0.5;
    return 0.9189385332046727 + (zm1 + 0.5) * log(t) - t + log(sum);
}
static double gammaln_nonnegative_scalar(double v) { /* gammaln.rs:254-271 */
    if (isnan(v)) return NAN;
    if (v == 0.0 || v == INFINITY) return INFINITY;
    if (v < 0.0) return NAN;
    if (v < 1.0e-305) return -log(v);
    if (v < 0.5) return log(M_PI) - log(sin(M_PI * v)) - lanczos_gammaln(1.0 - v);
    return lanczos_gammaln(v);
}
static double factorial_scalar(double v) { /* factorial.rs:25-34, 272-314 */
    if (isnan(v)) return NAN;
    if (v == 0.0) return 1.0;
    if (isinf(v)) return v > 0.0 ? INFINITY : NAN;
    if (v < 0.0) return NAN;
    double rounded = round(v);
    if (fabs(v - rounded) > DBL_EPSILON * fmax(fabs(v), 1.0)) return NAN;
    if (rounded > 170.0) return INFINITY;
    double acc = 1.0;
    for (int n = 1; n <= (int)rounded; ++n) acc *= (double)n;
    return acc;
}
static double erfcinv_positive_tail(double target) { /* erfcinv.rs:287-308 */
    double lo = 0.0, hi = 1.0;
    while (hi < 32.0 && erfc(hi) > target) {
        lo = hi;
        hi *= 2.0;
    }
    if (erfc(hi) > target) return hi;
    for (int i = 0; i < 110; ++i) {
        double mid = 0.5 * (lo + hi);
        if (erfc(mid) > target) lo = mid;
        else hi = mid;
    }
    return 0.5 * (lo + hi);
}
static double erfcinv_scalar(double v) { /* erfcinv.rs:261-281 */
    if (isnan(v)) return NAN;
    if (!(v >= 0.0 && v <= 2.0)) return NAN;
    if (v == 0.0) return INFINITY;
    if (v == 2.0) return -INFINITY;
    if (v == 1.0) return 0.0;
    if (v > 1.0) return -erfcinv_positive_tail(2.0 - v);
    return erfcinv_positive_tail(v);
}

/* crates/runmat-runtime/src/builtins/math/elementwise/sign.rs:236-246 */
static double sign_real_scalar(double x) {
    if (x > 0.0) return 1.0;
    if (x < 0.0) return -1.0;
    if (x == 0.0) return 0.0;
    return x; /* NaN propagates */
}

static double unary_apply(int op, double v) {
    switch (op) {
        case ORC_SIN: return sin(v);
        case ORC_COS: return cos(v);
        case ORC_TAN: return tan(v);
        case ORC_ASIN: return asin(v);
        case ORC_ACOS: return acos(v);
        case ORC_ATAN: return atan(v);
        case ORC_SINH: return sinh(v);
        case ORC_COSH: return cosh(v);
        case ORC_TANH: return tanh(v);
        case ORC_ASINH: return asinh(v);
        case ORC_ACOSH: return acosh(v);
        case ORC_ATANH: return atanh(v);
        case ORC_EXP: return exp(v);
        case ORC_EXPM1: return expm1(v);  /* Rust f64::exp_m1 */
        case ORC_LOG: return log(v);      /* Rust f64::ln */
        case ORC_LOG2: return log2(v);
        case ORC_LOG10: return log10(v);
        case ORC_LOG1P: return log1p(v);  /* Rust f64::ln_1p */
        case ORC_SQRT: return sqrt(v);
        case ORC_ABS: return fabs(v);
        case ORC_SIGN: return sign_real_scalar(v);
        case ORC_FLOOR: return floor(v);
        case ORC_CEIL: return ceil(v);
        case ORC_ROUND: return round(v);  /* Rust f64::round: half away from zero (round.rs:305) */
        case ORC_FIX: return trunc(v);
        case ORC_NEG: return -v;
        case ORC_EXP2: return exp2(v);
        case ORC_HEAVISIDE:               /* fusion.rs:2945-2953 select chain == CPU heaviside */
            if (v != v) return v;
            return v > 0.0 ? 1.0 : (v == 0.0 ? 0.5 : 0.0);
        case ORC_ISNAN: return (v != v) ? 1.0 : 0.0;
        case ORC_ISINF: return isinf(v) ? 1.0 : 0.0;
        case ORC_ISFINITE: return isfinite(v) ? 1.0 : 0.0;
        case ORC_UPLUS: return v;
        case ORC_SINGLE: return (double)(float)v; /* crates/runmat-builtins/src/lib.rs:426-436 */
        case ORC_DOUBLE: return v;
        case ORC_ERF: return erf(v);              /* libm::erf, elementwise/erf.rs:214-216 */
        case ORC_NOT: return v == 0.0 ? 1.0 : 0.0; /* logical_not, simple_provider.rs:4776-4780 */
        case ORC_GAMMA: return gamma_real_scalar(v);
        case ORC_FACTORIAL: return factorial_scalar(v);
        case ORC_NEXTPOW2: { double ax = fabs(v); return ax == 0.0 ? 0.0 : ceil(log2(ax)); } /* nextpow2.rs:157-164 */
        case ORC_GAMMALN: return gammaln_nonnegative_scalar(v);
        case ORC_ERFCINV: return erfcinv_scalar(v);
        /* map_nan_to_zero / not_nan_mask: crates/runmat-accelerate/src/backend/wgpu/shaders/nan.rs (`select(v, 0, v != v)`,
         * `select(0, 1, !(v != v))`); the CPU omitnan paths they replace skip NaNs the same way (reduction/sum.rs:1058-1066) */
        case ORC_NAN_TO_ZERO: return (v != v) ? 0.0 : v;
        case ORC_NOT_NAN: return (v != v) ? 0.0 : 1.0;
        case ORC_SINC: {                          /* sinc.rs:302-311 */
            if (v == 0.0) return 1.0;
            if (isfinite(v) && v == trunc(v)) return 0.0;
            double scaled = M_PI * v;
            return sin(scaled) / scaled;
        }
        default: return NAN;
    }
}

ORC_API int orc_unary(int op, const double* x, size_t n, double* out) {
    if (op < 0 || op >= ORC_UNARY_COUNT) return 1;
    for (size_t i = 0; i < n; ++i) out[i] = unary_apply(op, x[i]);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Binary maps with MATLAB implicit expansion.
 *   shape rules + index walk: crates/runmat-runtime/src/builtins/common/broadcast.rs:95-228
 *   (BroadcastPlan::new front-pads the shorter shape with 1s; iter() walks column-major and
 *   advances each operand by its stride unless that extent is 1)
 *   per-element ops: math/elementwise/times.rs:682-700 (a*b), plus.rs, minus.rs, rdivide.rs,
 *   power.rs:345-360 (powf), reduction/max.rs:2323-2344 + 1715-1728 (NaN/-0 policy),
 *   reduction/min.rs:1519-1531.
 * ---------------------------------------------------------------------------------------- */
enum { ORC_ADD = 0, ORC_SUB, ORC_MUL, ORC_DIV, ORC_POW, ORC_MAX, ORC_MIN, ORC_HYPOT, ORC_ATAN2,
       ORC_MOD, ORC_REM, ORC_EQ, ORC_NE, ORC_LT, ORC_LE, ORC_GT, ORC_GE, ORC_AND, ORC_OR, ORC_XOR };

static double elem_max(double a, double b) { /* max.rs:2323-2344, Include-NaN, Auto comparison */
    if (a != a || b != b) return NAN;
    if (b > a) return b;
    if (b < a) return a;
    if (b == 0.0 && a == 0.0) return (!signbit(b) && signbit(a)) ? b : a;
    return a;
}
static double elem_min(double a, double b) { /* min.rs:1519-1531 */
    if (a != a || b != b) return NAN;
    if (b < a) return b;
    if (b > a) return a;
    if (b == 0.0 && a == 0.0) return (signbit(b) && !signbit(a)) ? b : a;
    return a;
}
/* mod/rem follow the select chains the fused generator emits (fusion.rs:2954-2970), which the
 * reference's VM tests check against the CPU builtins (crates/runmat-vm/tests/fusion_gpu.rs:2965-3220). */
static double elem_mod(double l, double r) {
    if (isinf(r) && isfinite(l)) {
        return (l == 0.0 || sign_real_scalar(l) == sign_real_scalar(r)) ? l : r;
    }
    return l - r * floor(l / r);
}
static double elem_rem(double l, double r) {
    if (isinf(r) && isfinite(l)) return l;
    return l - r * trunc(l / r);
}

static double binary_apply(int op, double a, double b) {
    switch (op) {
        case ORC_ADD: return a + b;
        case ORC_SUB: return a - b;
        case ORC_MUL: return a * b;
        case ORC_DIV: return a / b;
        case ORC_POW: return pow(a, b);
        case ORC_MAX: return elem_max(a, b);
        case ORC_MIN: return elem_min(a, b);
        case ORC_HYPOT: return hypot(a, b);
        case ORC_ATAN2: return atan2(a, b);
        case ORC_MOD: return elem_mod(a, b);
        case ORC_REM: return elem_rem(a, b);
        /* comparisons and logicals: simple_provider.rs:4468-4760 (`lhs < rhs`, `lhs != 0.0 && rhs != 0.0`, ...) */
        case ORC_EQ: return a == b ? 1.0 : 0.0;
        case ORC_NE: return a != b ? 1.0 : 0.0;
        case ORC_LT: return a < b ? 1.0 : 0.0;
        case ORC_LE: return a <= b ? 1.0 : 0.0;
        case ORC_GT: return a > b ? 1.0 : 0.0;
        case ORC_GE: return a >= b ? 1.0 : 0.0;
        case ORC_AND: return (a != 0.0 && b != 0.0) ? 1.0 : 0.0;
        case ORC_OR: return (a != 0.0 || b != 0.0) ? 1.0 : 0.0;
        case ORC_XOR: return ((a != 0.0) != (b != 0.0)) ? 1.0 : 0.0;
        default: return NAN;
    }
}

#define ORC_MAX_RANK 16

/* broadcast.rs:8-47 (broadcast_shapes): returns rank, or (size_t)-1 on mismatch. */
ORC_API size_t orc_broadcast_shape(const size_t* sa, size_t ra, const size_t* sb, size_t rb,
                                   size_t* out) {
    size_t rank = ra > rb ? ra : rb;
    if (rank > ORC_MAX_RANK) return (size_t)-1;
    for (size_t d = 0; d < rank; ++d) {
        size_t a = d < rank - ra ? 1 : sa[d - (rank - ra)];
        size_t b = d < rank - rb ? 1 : sb[d - (rank - rb)];
        if (a == b) out[d] = a;
        else if (a == 1) out[d] = b;
        else if (b == 1) out[d] = a;
        else if (a == 0 || b == 0) out[d] = 0;
        else return (size_t)-1;
    }
    return rank;
}

ORC_API int orc_binary(int op, const double* a, const size_t* sa, size_t ra, const double* b,
                       const size_t* sb, size_t rb, double* out, size_t* out_shape,
                       size_t* out_rank) {
    size_t oshape[ORC_MAX_RANK];
    size_t rank = orc_broadcast_shape(sa, ra, sb, rb, oshape);
    if (rank == (size_t)-1) return 1;
    size_t ext_a[ORC_MAX_RANK], ext_b[ORC_MAX_RANK], adv_a[ORC_MAX_RANK], adv_b[ORC_MAX_RANK];
    size_t stride_a = 1, stride_b = 1, len = 1;
    for (size_t d = 0; d < rank; ++d) {
        ext_a[d] = d < rank - ra ? 1 : sa[d - (rank - ra)];
        ext_b[d] = d < rank - rb ? 1 : sb[d - (rank - rb)];
        adv_a[d] = ext_a[d] <= 1 ? 0 : stride_a; /* broadcast.rs:141-150 */
        adv_b[d] = ext_b[d] <= 1 ? 0 : stride_b;
        stride_a *= ext_a[d] > 1 ? ext_a[d] : 1;  /* compute_strides, broadcast.rs:50-58 */
        stride_b *= ext_b[d] > 1 ? ext_b[d] : 1;
        len *= oshape[d];
    }
    if (out_shape) memcpy(out_shape, oshape, rank * sizeof(size_t));
    if (out_rank) *out_rank = rank;
    size_t coords[ORC_MAX_RANK] = {0};
    size_t ia = 0, ib = 0;
    for (size_t off = 0; off < len; ++off) { /* BroadcastIter::next, broadcast.rs:196-228 */
        out[off] = binary_apply(op, a[ia], b[ib]);
        if (off + 1 == len) break;
        for (size_t d = 0; d < rank; ++d) {
            if (oshape[d] == 0) continue;
            coords[d] += 1;
            if (coords[d] < oshape[d]) {
                ia += adv_a[d];
                ib += adv_b[d];
                break;
            }
            coords[d] = 0;
            ia -= adv_a[d] * (oshape[d] - 1);
            ib -= adv_b[d] * (oshape[d] - 1);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * sum / mean  -- crates/runmat-runtime/src/builtins/math/reduction/sum.rs:996-1079 (sum_tensor)
 * One pass over the column-major linear index; each element is added into the output slot whose
 * reduced coordinates are zeroed => per output the additions happen in ascending linear order.
 * nan_mode 0 = Include (any NaN => NaN), 1 = Omit.
 * `reduce_mask[d]` nonzero marks a reduced dimension.
 * mean: crates/runmat-runtime/src/builtins/math/reduction/mean.rs:1134-1151 divides the sum by the
 * element count (a division, not a reciprocal multiply).
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_sum(const double* x, const size_t* shape, size_t rank, const int* reduce_mask,
                    int nan_mode, int mean, double* out) {
    if (rank > ORC_MAX_RANK) return 1;
    size_t oshape[ORC_MAX_RANK], total = 1, out_len = 1, reduce_count = 1;
    for (size_t d = 0; d < rank; ++d) {
        oshape[d] = reduce_mask[d] ? 1 : shape[d];
        total *= shape[d];
        out_len *= oshape[d];
        if (reduce_mask[d]) reduce_count *= shape[d];
    }
    double* sums = (double*)calloc(out_len ? out_len : 1, sizeof(double));
    unsigned char* saw_nan = (unsigned char*)calloc(out_len ? out_len : 1, 1);
    size_t* counts = (size_t*)calloc(out_len ? out_len : 1, sizeof(size_t));
    if (!sums || !saw_nan || !counts) return 2;
    size_t coords[ORC_MAX_RANK];
    for (size_t linear = 0; linear < total; ++linear) {
        size_t rem = linear; /* linear_to_multi */
        for (size_t d = 0; d < rank; ++d) {
            coords[d] = shape[d] ? rem % shape[d] : 0;
            if (shape[d]) rem /= shape[d];
        }
        size_t out_idx = 0, stride = 1; /* multi_to_linear over the output shape */
        for (size_t d = 0; d < rank; ++d) {
            size_t c = reduce_mask[d] ? 0 : coords[d];
            out_idx += c * stride;
            stride *= oshape[d];
        }
        double value = x[linear];
        if (value != value) {
            if (nan_mode == 0) saw_nan[out_idx] = 1;
        } else {
            sums[out_idx] += value;
            counts[out_idx] += 1;
        }
    }
    for (size_t i = 0; i < out_len; ++i) {
        double r;
        if (nan_mode == 0 && saw_nan[i]) r = NAN;
        else r = sums[i]; /* saw_value false => 0.0, which sums[i] already is */
        if (mean) {
            if (nan_mode == 0) r = saw_nan[i] ? NAN : r / (double)reduce_count;
            else r = counts[i] ? r / (double)counts[i] : NAN;
        }
        out[i] = r;
    }
    free(sums);
    free(saw_nan);
    free(counts);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * RNG -- crates/runmat-runtime/src/builtins/common/random.rs
 *   constants :7-13, mix_seed :128-141, advance_state :238-256, next_uniform_state :271-278,
 *   next_normal_pair :279-288, generate_normal :530-543.
 * 64-bit LCG s <- s*6364136223846793005 + 1; uniform = (s >> 11) * 2^-53.
 * ---------------------------------------------------------------------------------------- */
#define RNG_MULT 6364136223846793005ULL
#define RNG_INC 1ULL
#define DEFAULT_RNG_SEED 0x9e3779b97f4a7c15ULL

ORC_API uint64_t orc_rng_default_seed(void) { return DEFAULT_RNG_SEED; }

ORC_API uint64_t orc_rng_mix_seed(uint64_t seed) {
    if (seed == 0) return DEFAULT_RNG_SEED;
    uint64_t z = seed + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    uint64_t mixed = z ^ (z >> 31);
    return mixed == 0 ? DEFAULT_RNG_SEED : mixed;
}

ORC_API uint64_t orc_rng_advance(uint64_t state, uint64_t delta) {
    if (delta == 0) return state;
    uint64_t cur_mult = RNG_MULT, cur_plus = RNG_INC, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult = acc_mult * cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = cur_plus * (cur_mult + 1);
        cur_mult = cur_mult * cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

static double next_uniform_state(uint64_t* state) {
    *state = (*state) * RNG_MULT + RNG_INC;
    uint64_t bits = *state >> 11;
    return (double)bits * (1.0 / 9007199254740992.0);
}

/* generate_uniform: `len` draws; returns the advanced state through *state. */
ORC_API void orc_rng_uniform(uint64_t* state, size_t len, double* out) {
    for (size_t i = 0; i < len; ++i) out[i] = next_uniform_state(state);
}

/* generate_normal: Box-Muller pairs (z0, z1) emitted consecutively; odd len drops the last z1
 * but the state has still advanced by 2 per pair (random.rs:530-543). */
ORC_API void orc_rng_normal(uint64_t* state, size_t len, double* out) {
    size_t n = 0;
    while (n < len) {
        double u1 = next_uniform_state(state);
        if (u1 <= 0.0) u1 = 2.2250738585072014e-308; /* f64::MIN_POSITIVE, random.rs:13,281-283 */
        double u2 = next_uniform_state(state);
        double radius = sqrt(-2.0 * log(u1));
        double angle = 2.0 * M_PI * u2;
        out[n++] = radius * cos(angle);
        if (n < len) out[n++] = radius * sin(angle);
    }
}

/* generate_uniform_scaled (random.rs:514-528; simple_provider.rs:3607-3626): a + (b - a) * u, one draw per element */
ORC_API void orc_rng_unifrnd(uint64_t* state, double a, double b, size_t len, double* out) {
    for (size_t i = 0; i < len; ++i) out[i] = a + (b - a) * next_uniform_state(state);
}

/* generate_exponential (random.rs:290-300; simple_provider.rs:3560-3580): -mu * ln(max(u, MIN_POSITIVE)) */
ORC_API void orc_rng_exponential(uint64_t* state, double mu, size_t len, double* out) {
    for (size_t i = 0; i < len; ++i) {
        double u = next_uniform_state(state);
        if (!(u >= 2.2250738585072014e-308)) u = 2.2250738585072014e-308;
        out[i] = -mu * log(u);
    }
}

/* generate_normal_scaled (random.rs:302-320; simple_provider.rs:3582-3605): mu + sigma * z over whole Box-Muller pairs */
ORC_API void orc_rng_normrnd(uint64_t* state, double mu, double sigma, size_t len, double* out) {
    size_t n = 0;
    while (n < len) {
        double u1 = next_uniform_state(state);
        if (u1 <= 0.0) u1 = 2.2250738585072014e-308;
        double u2 = next_uniform_state(state);
        double radius = sqrt(-2.0 * log(u1));
        double angle = 2.0 * M_PI * u2;
        out[n++] = mu + sigma * (radius * cos(angle));
        if (n < len) out[n++] = mu + sigma * (radius * sin(angle));
    }
}

/* random_integer_range (simple_provider.rs:3683-3725): lower + min(floor(u * span), span - 1), span = upper - lower + 1 <= 2^53;
 * a span of one consumes no draws.  Returns 0, or 1 for a refused range (lower > upper, span > 2^53). */
ORC_API int orc_rng_integer_range(uint64_t* state, long long lower, long long upper, size_t len, double* out) {
    if (lower > upper) return 1;
    const __int128 span128 = (__int128)upper - (__int128)lower + 1;
    if (span128 > ((__int128)1 << 53)) return 1;
    const uint64_t span = (uint64_t)span128;
    if (span == 1) {
        for (size_t i = 0; i < len; ++i) out[i] = (double)lower;
        return 0;
    }
    const double span_f = (double)span;
    for (size_t i = 0; i < len; ++i) {
        uint64_t offset = (uint64_t)floor(next_uniform_state(state) * span_f);
        if (offset >= span) offset = span - 1;
        out[i] = (double)((__int128)lower + (__int128)offset);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * LU -- crates/runmat-accelerate/src/host_lu.rs:19-119 (lu_factor_host)
 * Doolittle with partial pivoting on a row-major working copy; pivot = FIRST row with strictly
 * larger |a| (:38-47); |pivot| <= 1e-12 zeroes the sub-column and skips the update (:54-59).
 * Outputs (all column-major): combined rows x cols, lower rows x rows (unit diagonal),
 * upper rows x cols, perm matrix rows x rows, pivot vector rows x 1 (1-based row ids).
 * Any output pointer may be NULL.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_lu(const double* data, size_t rows, size_t cols, double* combined, double* lower,
                   double* upper, double* perm_matrix, double* pivot_vector) {
    double* m = (double*)malloc(sizeof(double) * (rows * cols + 1));
    size_t* perm = (size_t*)malloc(sizeof(size_t) * (rows ? rows : 1));
    if (!m || !perm) return 2;
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) m[r * cols + c] = data[r + c * rows];
    for (size_t r = 0; r < rows; ++r) perm[r] = r;
    size_t min_dim = rows < cols ? rows : cols;
    for (size_t k = 0; k < min_dim; ++k) {
        size_t pivot_row = k;
        double pivot_abs = 0.0;
        for (size_t r = k; r < rows; ++r) {
            double a = fabs(m[r * cols + k]);
            if (a > pivot_abs) {
                pivot_abs = a;
                pivot_row = r;
            }
        }
        if (pivot_row != k) {
            for (size_t c = 0; c < cols; ++c) {
                double t = m[k * cols + c];
                m[k * cols + c] = m[pivot_row * cols + c];
                m[pivot_row * cols + c] = t;
            }
            size_t t = perm[k];
            perm[k] = perm[pivot_row];
            perm[pivot_row] = t;
        }
        if (pivot_abs <= 1.0e-12) {
            for (size_t r = k + 1; r < rows; ++r) m[r * cols + k] = 0.0;
            continue;
        }
        double pivot = m[k * cols + k];
        for (size_t r = k + 1; r < rows; ++r) {
            double factor = m[r * cols + k] / pivot;
            m[r * cols + k] = factor;
            for (size_t c = k + 1; c < cols; ++c) m[r * cols + c] -= factor * m[k * cols + c];
        }
    }
    if (combined)
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) combined[r + c * rows] = m[r * cols + c];
    if (lower) {
        size_t limit = min_dim;
        for (size_t i = 0; i < rows; ++i)
            for (size_t j = 0; j < rows; ++j) {
                double v = 0.0;
                if (i == j) v = 1.0;
                else if (i > j && j < limit) v = m[i * cols + j];
                lower[i + j * rows] = v;
            }
    }
    if (upper)
        for (size_t i = 0; i < rows; ++i)
            for (size_t j = 0; j < cols; ++j) upper[i + j * rows] = (i <= j) ? m[i * cols + j] : 0.0;
    if (perm_matrix) {
        memset(perm_matrix, 0, sizeof(double) * rows * rows);
        for (size_t r = 0; r < rows; ++r) perm_matrix[r + perm[r] * rows] = 1.0;
    }
    if (pivot_vector)
        for (size_t r = 0; r < rows; ++r) pivot_vector[r] = (double)(perm[r] + 1);
    free(m);
    free(perm);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * mldivide -- crates/runmat-runtime/src/builtins/math/linalg/ops/mldivide.rs:317-404
 * Reference: scalar lhs => rhs * (1/lhs) (:321-325); else x = pinv_tol(A) * B via nalgebra SVD
 * with tol = eps * max(m,n) * max(sigma_max, 1) (:396-404).  nalgebra is third-party and absent
 * from /root/reference, so the SVD here is a one-sided (Hestenes) Jacobi SVD -- same
 * mathematical result (minimum-norm least-squares solution with singular values <= tol dropped),
 * different rounding. PARITY UNPINNED at bit level; pinned by residuals only.
 * A is m x n, B is m x nrhs, X is n x nrhs. Returns 0, or 1 on shape error.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_mldivide_svd(const double* A, size_t m, size_t n, const double* B, size_t brows,
                             size_t nrhs, double* X) {
    if (m == 1 && n == 1) {
        double r = 1.0 / A[0];
        for (size_t i = 0; i < brows * nrhs; ++i) X[i] = B[i] * r;
        return 0;
    }
    if (brows != m) return 1;
    if (m == 0) {
        memset(X, 0, sizeof(double) * n * nrhs);
        return 0;
    }
    /* One-sided Jacobi on W (m x n, or on A^T when m < n so the working matrix is tall). */
    int transposed = m < n;
    size_t p = transposed ? n : m, q = transposed ? m : n; /* W is p x q, p >= q */
    double* W = (double*)malloc(sizeof(double) * p * q);
    double* V = (double*)calloc(q * q, sizeof(double));
    double* sig = (double*)malloc(sizeof(double) * q);
    if (!W || !V || !sig) return 2;
    for (size_t j = 0; j < q; ++j)
        for (size_t i = 0; i < p; ++i) W[i + j * p] = transposed ? A[j + i * m] : A[i + j * m];
    for (size_t j = 0; j < q; ++j) V[j + j * q] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (size_t a = 0; a + 1 < q; ++a) {
            for (size_t b = a + 1; b < q; ++b) {
                double alpha = 0, beta = 0, gamma = 0;
                for (size_t i = 0; i < p; ++i) {
                    alpha += W[i + a * p] * W[i + a * p];
                    beta += W[i + b * p] * W[i + b * p];
                    gamma += W[i + a * p] * W[i + b * p];
                }
                if (gamma == 0.0) continue;
                double lim = fabs(gamma) / sqrt(alpha * beta);
                if (lim > off) off = lim;
                if (lim < 1e-15) continue;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (size_t i = 0; i < p; ++i) {
                    double wa = W[i + a * p], wb = W[i + b * p];
                    W[i + a * p] = c * wa - s * wb;
                    W[i + b * p] = s * wa + c * wb;
                }
                for (size_t i = 0; i < q; ++i) {
                    double va = V[i + a * q], vb = V[i + b * q];
                    V[i + a * q] = c * va - s * vb;
                    V[i + b * q] = s * va + c * vb;
                }
            }
        }
        if (off < 1e-15) break;
    }
    double smax = 0.0;
    for (size_t j = 0; j < q; ++j) {
        double s2 = 0;
        for (size_t i = 0; i < p; ++i) s2 += W[i + j * p] * W[i + j * p];
        sig[j] = sqrt(s2);
        if (sig[j] > smax) smax = sig[j];
    }
    double maxdim = (double)(m > n ? m : n);
    double tol = 2.220446049250313e-16 * maxdim * (smax > 1.0 ? smax : 1.0); /* :396-404 */
    /* W = U*diag(sig) (columns), so U_j = W_j / sig_j.
     * not transposed: A = U S V^T  => X = V S^-1 U^T B
     * transposed:     A^T = U S V^T => A = V S U^T => X = U S^-1 V^T B              */
    for (size_t r = 0; r < nrhs; ++r) {
        for (size_t i = 0; i < n; ++i) X[i + r * n] = 0.0;
        for (size_t j = 0; j < q; ++j) {
            if (sig[j] <= tol) continue;
            double coef = 0.0;
            if (!transposed) {
                for (size_t i = 0; i < m; ++i) coef += (W[i + j * p] / sig[j]) * B[i + r * m];
                coef /= sig[j];
                for (size_t i = 0; i < n; ++i) X[i + r * n] += V[i + j * q] * coef;
            } else {
                for (size_t i = 0; i < m; ++i) coef += V[i + j * q] * B[i + r * m];
                coef /= sig[j];
                for (size_t i = 0; i < n; ++i) X[i + r * n] += (W[i + j * p] / sig[j]) * coef;
            }
        }
    }
    free(W);
    free(V);
    free(sig);
    return 0;
}

/* LU-based A\b for square well-conditioned A: the algorithm the HIP path implements (LU with the
 * host_lu.rs pivot rule, then forward/back substitution). Used as the bit-pattern-adjacent
 * comparator at sizes where the SVD restatement is too slow. Returns 3 if a pivot is <= 1e-12. */
/* ---- linsolve (triangular hints) --------------------------------------------------------------
 * crates/runmat-runtime/src/builtins/math/linalg/solve/linsolve.rs:769-800 (forward) / 802-833
 * (backward): per rhs column, accum = sum_j T[i,j]*x[j] in ascending j, x[i] = (b[i]-accum)/d;
 * rcond = min|d|/max|d| (common/linalg.rs:232-238, f64::min/max ignore NaN); a zero diagonal entry
 * is the singular error (return 3).  TRANSA (linsolve.rs:698-705) is the caller's job: transpose
 * (orc_transpose) and swap lower<->upper. */
ORC_API int orc_linsolve_tri(int lower, const double* T, size_t n, const double* B, size_t nrhs, double* X, double* rcond) {
    double min_diag = INFINITY, max_diag = 0.0;
    memcpy(X, B, sizeof(double) * n * nrhs);
    for (size_t col = 0; col < nrhs; ++col) {
        for (size_t step = 0; step < n; ++step) {
            const size_t i = lower ? step : n - 1 - step;
            const double diag = T[i + i * n];
            const double da = fabs(diag);
            min_diag = fmin(min_diag, da);
            max_diag = fmax(max_diag, da);
            if (da == 0.0) return 3;
            double accum = 0.0;
            if (lower) for (size_t j = 0; j < i; ++j) accum += T[i + j * n] * X[j + col * n];
            else for (size_t j = i + 1; j < n; ++j) accum += T[i + j * n] * X[j + col * n];
            X[i + col * n] = (X[i + col * n] - accum) / diag;
        }
    }
    if (rcond) *rcond = max_diag == 0.0 ? 0.0 : min_diag / max_diag;
    return 0;
}

/* image_normalize: simple_provider.rs:7893-7993 == cpu_image_normalize (accelerate/tests/image_normalize.rs:7-66);
 * data is [batch, height, width] column-major (batch fastest). */
ORC_API void orc_image_normalize(const double* data, size_t batch, size_t height, size_t width, double epsilon, int has_gain,
                                 double gain, int has_bias, double bias, int clamp_zero, int has_gamma, double gamma,
                                 double* out) {
    const size_t plane = height * width, stride_h = batch, stride_w = batch * height;
    for (size_t b = 0; b < batch; ++b) {
        double sum = 0.0;
        for (size_t w = 0; w < width; ++w)
            for (size_t h = 0; h < height; ++h) sum += data[b + h * stride_h + w * stride_w];
        const double mean = sum / (double)plane;
        double sq_sum = 0.0;
        for (size_t w = 0; w < width; ++w)
            for (size_t h = 0; h < height; ++h) {
                double diff = data[b + h * stride_h + w * stride_w] - mean;
                sq_sum += diff * diff;
            }
        const double variance = sq_sum / (double)plane;
        const double sigma = sqrt(variance + epsilon);
        const double inv_sigma = sigma > 0.0 ? 1.0 / sigma : 0.0;
        for (size_t w = 0; w < width; ++w)
            for (size_t h = 0; h < height; ++h) {
                const size_t idx = b + h * stride_h + w * stride_w;
                double value = (data[idx] - mean) * inv_sigma;
                if (has_gain) value *= gain;
                if (has_bias) value += bias;
                if (clamp_zero) value = fmax(value, 0.0);
                if (has_gamma) value = pow(value, gamma);
                out[idx] = value;
            }
    }
}

/* matmul_power_step: simple_provider.rs:7852-7891 */
ORC_API int orc_matmul_power_step(const double* a, size_t m, size_t k, const double* b, size_t kb, size_t n, double epsilon,
                                  double* out) {
    int rc = orc_matmul(a, m, k, b, kb, n, out);
    if (rc) return rc;
    for (size_t col = 0; col < n; ++col) {
        double acc = 0.0;
        for (size_t row = 0; row < m; ++row) {
            double val = out[row + col * m];
            acc += val * val;
        }
        acc += epsilon;
        const double norm = sqrt(acc);
        for (size_t row = 0; row < m; ++row) out[row + col * m] /= norm;
    }
    return 0;
}

/* covariance_dense with CovWeightSpec::Scalar (cov.rs:916-953), covariance_unweighted_pair (:1080-1100),
 * sanitize_covariance (:1218-1227).  data rows x cols column-major; out cols x cols. */
ORC_API void orc_covariance(const double* data, size_t rows, size_t cols, int biased, double* out) {
    for (size_t i = 0; i < cols * cols; ++i) out[i] = NAN;
    if (cols == 0) return;
    const double denom = biased ? (double)rows : (double)rows - 1.0;
    if (denom <= 0.0) return;
    double* means = (double*)malloc(sizeof(double) * cols);
    for (size_t col = 0; col < cols; ++col) {
        double sum = 0.0;
        int valid = 1;
        for (size_t r = 0; r < rows; ++r) {
            double v = data[r + col * rows];
            if (!isfinite(v)) { valid = 0; break; }
            sum += v;
        }
        means[col] = valid ? sum / (double)rows : NAN;
    }
    for (size_t i = 0; i < cols; ++i)
        for (size_t j = i; j < cols; ++j) {
            double value;
            if (!isfinite(means[i]) || !isfinite(means[j])) {
                value = NAN;
            } else {
                double acc = 0.0;
                int bad = 0;
                for (size_t r = 0; r < rows; ++r) {
                    double x = data[r + i * rows], y = data[r + j * rows];
                    if (!isfinite(x) || !isfinite(y)) { bad = 1; break; }
                    acc += (x - means[i]) * (y - means[j]);
                }
                value = bad ? NAN : acc / denom;
            }
            if (isfinite(value) && i == j && value < 0.0 && value > -1.0e-12) value = 0.0;
            out[i + j * cols] = value;
            out[j + i * cols] = value;
        }
    free(means);
}

/* syrk: A' * A, the reference's own CPU comparator crates/runmat-accelerate/tests/syrk.rs:14-31
 * (upper triangle accumulated over k in order, unfused, mirrored into the lower triangle). */
ORC_API void orc_syrk(const double* a, size_t rows, size_t cols, double* out) {
    for (size_t col = 0; col < cols; ++col)
        for (size_t row = 0; row <= col; ++row) {
            double acc = 0.0;
            for (size_t k = 0; k < rows; ++k) {
                double lhs = a[k + row * rows], rhs = a[k + col * rows];
                acc += lhs * rhs;
            }
            out[row + col * cols] = acc;
            if (row != col) out[col + row * cols] = acc;
        }
}

/* transpose_tensor, linsolve.rs:1067-1077 */
ORC_API void orc_transpose(const double* A, size_t rows, size_t cols, double* out) {
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) out[c + r * cols] = A[r + c * rows];
}

ORC_API int orc_mldivide_lu(const double* A, size_t n, const double* B, size_t nrhs, double* X) {
    double* comb = (double*)malloc(sizeof(double) * n * n);
    double* piv = (double*)malloc(sizeof(double) * n);
    if (!comb || !piv) return 2;
    orc_lu(A, n, n, comb, NULL, NULL, NULL, piv);
    for (size_t k = 0; k < n; ++k)
        if (fabs(comb[k + k * n]) <= 1.0e-12) {
            free(comb);
            free(piv);
            return 3;
        }
    for (size_t r = 0; r < nrhs; ++r) {
        double* x = X + r * n;
        for (size_t i = 0; i < n; ++i) x[i] = B[(size_t)(piv[i] - 1.0) + r * n];
        for (size_t i = 0; i < n; ++i) { /* L y = Pb, unit lower */
            double s = x[i];
            for (size_t k = 0; k < i; ++k) s -= comb[i + k * n] * x[k];
            x[i] = s;
        }
        for (size_t ii = n; ii-- > 0;) { /* U x = y */
            double s = x[ii];
            for (size_t k = ii + 1; k < n; ++k) s -= comb[ii + k * n] * x[k];
            x[ii] = s / comb[ii + ii * n];
        }
    }
    free(comb);
    free(piv);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Benchmark workloads restated on the CPU path (one temporary array per op, serial), used as the
 * `cpu_baseline` leg of bench.py and for parity at BASELINE.json sizes.
 * ---------------------------------------------------------------------------------------- */

/* D = sin(A).*B + C : three CPU builtin calls (sin.rs:245-255, times.rs:682-700, plus.rs), each
 * allocating a fresh output like the reference's `.collect()`. */
ORC_API int orc_sin_mul_add(const double* A, const double* B, const double* C, size_t n, double* D) {
    double* t0 = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* t1 = (double*)malloc(sizeof(double) * (n ? n : 1));
    if (!t0 || !t1) return 2;
    for (size_t i = 0; i < n; ++i) t0[i] = sin(A[i]);
    for (size_t i = 0; i < n; ++i) t1[i] = t0[i] * B[i];
    for (size_t i = 0; i < n; ++i) D[i] = t1[i] + C[i];
    free(t0);
    free(t1);
    return 0;
}

/* benchmarks/elementwise-math/runmat.m:10-13 in f64 (BASELINE.json configs[0]):
 *   y0 = sin(x).*exp(-x/10); y1 = y0.*cos(x/4) + 0.25.*(y0.^2); y2 = tanh(y1) + 0.1.*y1
 * evaluated op by op with temporaries (14 CPU builtin passes). y0.^2 uses powf (power.rs:358). */
ORC_API int orc_elementwise_math_chain(const double* x, size_t n, double* y2) {
    double* t[6];
    for (int k = 0; k < 6; ++k) {
        t[k] = (double*)malloc(sizeof(double) * (n ? n : 1));
        if (!t[k]) return 2;
    }
    for (size_t i = 0; i < n; ++i) t[0][i] = sin(x[i]);
    for (size_t i = 0; i < n; ++i) t[1][i] = -x[i];
    for (size_t i = 0; i < n; ++i) t[2][i] = t[1][i] / 10.0;
    for (size_t i = 0; i < n; ++i) t[1][i] = exp(t[2][i]);
    for (size_t i = 0; i < n; ++i) t[2][i] = t[0][i] * t[1][i]; /* y0 */
    for (size_t i = 0; i < n; ++i) t[0][i] = x[i] / 4.0;
    for (size_t i = 0; i < n; ++i) t[1][i] = cos(t[0][i]);
    for (size_t i = 0; i < n; ++i) t[3][i] = t[2][i] * t[1][i];
    for (size_t i = 0; i < n; ++i) t[0][i] = pow(t[2][i], 2.0);
    for (size_t i = 0; i < n; ++i) t[1][i] = 0.25 * t[0][i];
    for (size_t i = 0; i < n; ++i) t[4][i] = t[3][i] + t[1][i]; /* y1 */
    for (size_t i = 0; i < n; ++i) t[0][i] = tanh(t[4][i]);
    for (size_t i = 0; i < n; ++i) t[1][i] = 0.1 * t[4][i];
    for (size_t i = 0; i < n; ++i) y2[i] = t[0][i] + t[1][i];
    for (int k = 0; k < 6; ++k) free(t[k]);
    return 0;
}

/* benchmarks/monte-carlo-analysis/runmat_rng.m in f64 with the CPU randn stream:
 *   S = S0; for t: Z = randn(M,1); S = S .* exp(drift + scale.*Z); end
 *   price = mean(max(S-K,0),'all') * exp(-mu*T*dt)
 * `state` is the LCG state on entry (advanced on exit). */
ORC_API double orc_monte_carlo_price(uint64_t* state, size_t M, size_t T, double S0, double mu,
                                     double sigma, double dt, double K) {
    double* S = (double*)malloc(sizeof(double) * (M ? M : 1));
    double* Z = (double*)malloc(sizeof(double) * (M ? M : 1));
    if (!S || !Z) return NAN;
    double drift = (mu - 0.5 * sigma * sigma) * dt;
    double scale = sigma * sqrt(dt);
    for (size_t i = 0; i < M; ++i) S[i] = S0;
    for (size_t t = 0; t < T; ++t) {
        orc_rng_normal(state, M, Z);
        for (size_t i = 0; i < M; ++i) S[i] = S[i] * exp(drift + scale * Z[i]);
    }
    double sum = 0.0;
    for (size_t i = 0; i < M; ++i) sum += elem_max(S[i] - K, 0.0);
    double price = (sum / (double)M) * exp(-mu * (double)T * dt);
    free(S);
    free(Z);
    return price;
}

/* stochastic_evolution_host, builtins/stats/random/stochastic_evolution.rs:10-30 */
ORC_API int orc_stochastic_evolution(uint64_t* state, double* data, size_t len, double drift, double scale, uint32_t steps) {
    if (len == 0 || steps == 0) return 0;
    double* z = (double*)malloc(sizeof(double) * len);
    if (!z) return 2;
    for (uint32_t s = 0; s < steps; ++s) {
        orc_rng_normal(state, len, z);
        for (size_t i = 0; i < len; ++i) {
            double term = drift + scale * z[i];
            data[i] *= exp(term);
        }
    }
    free(z);
    return 0;
}

/* splitmix64-based uniform fill used by bench/tests to build identical inputs on host and device
 * (SURVEY.md 8(d) config 2/3): value = lo + (hi-lo) * (next53 * 2^-53), one splitmix64 step per
 * element with state = seed + (i+1)*0x9e3779b97f4a7c15 (counter form => order independent). */
ORC_API void orc_fill_uniform(uint64_t seed, double lo, double hi, size_t n, double* out) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t z = seed + (uint64_t)(i + 1) * 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z = z ^ (z >> 31);
        out[i] = lo + (hi - lo) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
    }
}

/* ---- reductions next to sum / mean on the provider trait (round 3) ------------------------------------------------------
 * All take the [pre, red, post] view of a column-major tensor (element (i, k, j) at i + pre * (k + red * j)) and reduce /
 * scan the middle extent, walking it in ascending order like the CPU's element loop does per output slot. */

/* min / max with indices: runtime/builtins/math/reduction/min.rs:1437-1477 (`update_best_real`), :1519-1531
 * (`should_replace_real`, Auto comparison), output :1056-1075; max.rs:1715-1727.  includenan: the first NaN fixes value NaN and
 * its index; omitnan: NaNs are skipped, nothing left gives (NaN, NaN).  Indices are 1-based positions along the reduced dim. */
static int orc_should_replace(int is_max, double current, double candidate) {
    if (is_max) {
        if (candidate > current) return 1;
        if (candidate < current) return 0;
        if (candidate == 0.0 && current == 0.0) return !signbit(candidate) && signbit(current);
        return 0;
    }
    if (candidate < current) return 1;
    if (candidate > current) return 0;
    if (candidate == 0.0 && current == 0.0) return signbit(candidate) && !signbit(current);
    return 0;
}
ORC_API int orc_minmax_dim(const double* x, size_t pre, size_t red, size_t post, int is_max, int omitnan, double* values,
                           double* indices) {
    for (size_t j = 0; j < post; ++j)
        for (size_t i = 0; i < pre; ++i) {
            double best = 0.0;
            size_t best_k = 0;
            int has_value = 0, nan_fixed = 0;
            for (size_t k = 0; k < red; ++k) {
                const double v = x[i + pre * (k + red * j)];
                if (isnan(v)) {
                    if (!omitnan && !nan_fixed) {
                        best_k = k;
                        has_value = 1;
                        nan_fixed = 1;
                    }
                    continue;
                }
                if (nan_fixed) continue;
                if (!has_value) {
                    best = v;
                    best_k = k;
                    has_value = 1;
                    continue;
                }
                if (orc_should_replace(is_max, best, v)) {
                    best = v;
                    best_k = k;
                }
            }
            const size_t o = i + pre * j;
            if (nan_fixed) {
                values[o] = NAN;
                indices[o] = (double)(best_k + 1);
            } else if (!has_value) {
                values[o] = NAN;
                indices[o] = NAN;
            } else {
                values[o] = best;
                indices[o] = (double)(best_k + 1);
            }
        }
    return 0;
}

/* std: std.rs:858-935 - Welford's update per element, NaNs set saw_nan (include) or are skipped (omit); sample: m2 / (n - 1),
 * 0 for one value; population: m2 / n; variance clamped at 0; no values -> NaN. */
ORC_API int orc_std_dim(const double* x, size_t pre, size_t red, size_t post, int population, int omitnan, double* out) {
    for (size_t j = 0; j < post; ++j)
        for (size_t i = 0; i < pre; ++i) {
            size_t count = 0;
            double mean = 0.0, m2 = 0.0;
            int saw_nan = 0;
            for (size_t k = 0; k < red; ++k) {
                const double v = x[i + pre * (k + red * j)];
                if (isnan(v)) {
                    if (!omitnan) saw_nan = 1;
                    continue;
                }
                count += 1;
                const double delta = v - mean;
                mean += delta / (double)count;
                const double delta2 = v - mean;
                m2 += delta * delta2;
            }
            double r;
            if ((saw_nan && !omitnan) || count == 0) r = NAN;
            else {
                double variance;
                if (population) variance = fmax(m2 / (double)count, 0.0);
                else variance = count > 1 ? fmax(m2 / (double)(count - 1), 0.0) : 0.0;
                r = sqrt(variance);
            }
            out[i + pre * j] = r;
        }
    return 0;
}

/* nnz (nnz.rs:358: NaN counts as non-zero), any (any.rs:620-621, 722-733), all (all.rs:568-569, 671-703: NaNs skipped in BOTH
 * modes, nothing left -> true).  op 0 / 1 / 2; results are f64 counts / 0-1 values. */
ORC_API int orc_truth_dim(const double* x, size_t pre, size_t red, size_t post, int op, int omitnan, double* out) {
    for (size_t j = 0; j < post; ++j)
        for (size_t i = 0; i < pre; ++i) {
            size_t nz = 0, nn = 0;
            for (size_t k = 0; k < red; ++k) {
                const double v = x[i + pre * (k + red * j)];
                if (isnan(v)) nn++;
                else if (v != 0.0) nz++;
            }
            double r;
            if (op == 0) r = (double)(nz + nn);
            else if (op == 1) r = (omitnan ? nz : nz + nn) > 0 ? 1.0 : 0.0;
            else r = (red - nz - nn) == 0 ? 1.0 : 0.0;
            out[i + pre * j] = r;
        }
    return 0;
}

/* cumsum / cumprod: cumsum.rs:586-650, cumprod.rs:606-670 - running value per line, forward or reverse; include: NaN from the
 * first NaN input on; omit: NaN inputs leave the running value unchanged. */
ORC_API int orc_cumulative(const double* x, size_t pre, size_t len, size_t post, int prod, int reverse, int omitnan, double* y) {
    for (size_t j = 0; j < post; ++j)
        for (size_t i = 0; i < pre; ++i) {
            double run = prod ? 1.0 : 0.0;
            int is_nan = 0;
            for (size_t s = 0; s < len; ++s) {
                const size_t k = reverse ? len - 1 - s : s;
                const size_t idx = i + pre * (k + len * j);
                const double v = x[idx];
                if (!omitnan) {
                    if (is_nan) {
                        y[idx] = NAN;
                        continue;
                    }
                    if (isnan(v)) {
                        is_nan = 1;
                        y[idx] = NAN;
                    } else {
                        run = prod ? run * v : run + v;
                        y[idx] = run;
                    }
                } else {
                    if (!isnan(v)) run = prod ? run * v : run + v;
                    y[idx] = run;
                }
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Shape / indexing hooks -- crates/runmat-accelerate/src/simple_provider.rs
 * ---------------------------------------------------------------------------------------- */

/* repmat_numeric, simple_provider.rs:2174-2240.  reps: nreps >= 1 factors; one factor = every dimension of
 * max(rank, 2) dimensions.  Writes the tiled shape to out_shape (capacity >= max(rank, nreps, 2)) and its rank to *out_rank;
 * `out` may be NULL to query the shape only.  Returns 1 for an empty factor list. */
ORC_API int orc_repmat(const double* data, const size_t* shape, size_t rank_in, const size_t* reps, size_t nreps, double* out,
                       size_t* out_shape, size_t* out_rank) {
    if (nreps == 0) return 1;
    const size_t orig_rank = rank_in == 0 ? 1 : rank_in;
    size_t rank = nreps == 1 ? (orig_rank > 2 ? orig_rank : 2) : (orig_rank > nreps ? orig_rank : nreps);
    size_t base[32], factors[32], strides[32];
    if (rank > 32) return 2;
    for (size_t i = 0; i < rank; ++i) base[i] = i < rank_in ? shape[i] : 1;
    for (size_t i = 0; i < rank; ++i) factors[i] = nreps == 1 ? reps[0] : (i < nreps ? reps[i] : 1);
    size_t new_total = 1;
    for (size_t i = 0; i < rank; ++i) {
        out_shape[i] = base[i] * factors[i];
        new_total *= out_shape[i];
    }
    *out_rank = rank;
    if (!out || new_total == 0) return 0;
    size_t s = 1;
    for (size_t i = 0; i < rank; ++i) { /* compute_strides */
        strides[i] = s;
        s *= base[i];
    }
    for (size_t idx = 0; idx < new_total; ++idx) {
        size_t rem = idx, src = 0;
        for (size_t d = 0; d < rank; ++d) {
            const size_t dim_size = out_shape[d];
            const size_t coord = rem % dim_size;
            rem /= dim_size;
            const size_t orig = base[d] == 0 ? 0 : coord % base[d];
            src += orig * strides[d];
        }
        out[idx] = data[src];
    }
    return 0;
}

/* permute_data, simple_provider.rs:1645-1740 (lane factor 1).  order: zero-based permutation of 0..norder-1, norder >= rank_in.
 * Returns 1 empty order, 2 order shorter than the rank, 3 index out of range, 4 duplicate. */
ORC_API int orc_permute(const double* data, const size_t* shape, size_t rank_in, const size_t* order, size_t norder, double* out,
                        size_t* out_shape) {
    if (norder == 0) return 1;
    if (rank_in > norder) return 2;
    if (norder > 32) return 5;
    int seen[32] = {0};
    for (size_t d = 0; d < norder; ++d) {
        if (order[d] >= norder) return 3;
        if (seen[order[d]]) return 4;
        seen[order[d]] = 1;
    }
    size_t src_shape[32], src_strides[32], dst_coords[32], src_coords[32];
    size_t total = 1, s = 1;
    for (size_t d = 0; d < norder; ++d) src_shape[d] = d < rank_in ? shape[d] : 1;
    for (size_t d = 0; d < norder; ++d) {
        src_strides[d] = s;
        s *= src_shape[d];
        out_shape[d] = src_shape[order[d]];
    }
    for (size_t d = 0; d < norder; ++d) total *= out_shape[d];
    for (size_t idx = 0; idx < total; ++idx) {
        size_t rem = idx;
        for (size_t d = 0; d < norder; ++d) {
            dst_coords[d] = rem % out_shape[d];
            rem /= out_shape[d];
        }
        for (size_t d = 0; d < norder; ++d) src_coords[order[d]] = dst_coords[d];
        size_t src = 0;
        for (size_t d = 0; d < norder; ++d) src += src_coords[d] * src_strides[d];
        out[idx] = data[src];
    }
    return 0;
}

/* gather_linear, simple_provider.rs:2609-2654: returns position + 1 of the first out-of-range index, 0 on success */
ORC_API size_t orc_gather_linear(const double* data, size_t len, const uint32_t* indices, size_t n, double* out) {
    for (size_t pos = 0; pos < n; ++pos) {
        if ((size_t)indices[pos] >= len) return pos + 1;
        out[pos] = data[indices[pos]];
    }
    return 0;
}

/* scatter_linear, simple_provider.rs:2656-2720: sequential, so a repeated index keeps its LAST value */
ORC_API size_t orc_scatter_linear(double* target, size_t len, const uint32_t* indices, size_t n, const double* values) {
    for (size_t pos = 0; pos < n; ++pos)
        if ((size_t)indices[pos] >= len) return pos + 1;
    for (size_t pos = 0; pos < n; ++pos) target[indices[pos]] = values[pos];
    return 0;
}

/* linspace, simple_provider.rs:3488-3503 */
ORC_API void orc_linspace(double start, double stop, size_t count, double* out) {
    if (count == 0) return;
    if (count == 1) {
        out[0] = stop;
        return;
    }
    const double step = (stop - start) / (double)(count - 1);
    for (size_t idx = 0; idx < count; ++idx) out[idx] = start + (double)idx * step;
    out[count - 1] = stop;
}

/* identity_data, simple_provider.rs:2293-2336 (shape already normalised by the caller: [] -> [1,1], [n] -> [n,n]) */
ORC_API void orc_eye(const size_t* shape, size_t rank, double* out) {
    size_t total = 1;
    for (size_t d = 0; d < rank; ++d) total *= shape[d];
    for (size_t i = 0; i < total; ++i) out[i] = 0.0;
    if (rank < 2 || total == 0) return;
    const size_t rows = shape[0], cols = shape[1], plane = rows * cols;
    const size_t diag = rows < cols ? rows : cols;
    for (size_t page = 0; page < total / plane; ++page)
        for (size_t d = 0; d < diag; ++d) out[page * plane + d + d * rows] = 1.0;
}

/* flip_data, simple_provider.rs:1739-1776: flags[axis] toggled per occurrence; shape already extended to cover the axes */
ORC_API void orc_flip(const double* data, const size_t* shape, size_t rank, const int* flags, double* out) {
    size_t total = 1;
    for (size_t d = 0; d < rank; ++d) total *= shape[d];
    for (size_t idx = 0; idx < total; ++idx) {
        size_t rem = idx, src = 0, stride = 1;
        for (size_t d = 0; d < rank; ++d) {
            size_t coord = rem % shape[d];
            rem /= shape[d];
            if (flags[d] && shape[d] > 1) coord = shape[d] - 1 - coord;
            src += coord * stride;
            stride *= shape[d];
        }
        out[idx] = data[src];
    }
}

/* circshift_data, simple_provider.rs:2083-2143: shifts has one entry per dimension of `shape` */
ORC_API void orc_circshift(const double* data, const size_t* shape, size_t rank, const long long* shifts, double* out) {
    size_t total = 1;
    for (size_t d = 0; d < rank; ++d) total *= shape[d];
    for (size_t idx = 0; idx < total; ++idx) {
        size_t rem = idx, src = 0, stride = 1;
        for (size_t d = 0; d < rank; ++d) {
            const size_t len = shape[d];
            size_t coord = rem % len;
            rem /= len;
            if (len > 1) {
                long long v = shifts[d] % (long long)len;
                if (v < 0) v += (long long)len;
                if (v != 0) coord = (coord + len - (size_t)v) % len;
            }
            src += coord * stride;
            stride *= len;
        }
        out[idx] = data[src];
    }
}

/* tril_data / triu_data, simple_provider.rs:1974-2081 */
ORC_API void orc_tri(const double* data, const size_t* shape, size_t rank, int upper, long long offset, double* out) {
    const size_t rows = rank > 0 ? shape[0] : 1, cols = rank > 1 ? shape[1] : 1;
    size_t pages = 1;
    for (size_t d = 2; d < rank; ++d) pages *= shape[d];
    const size_t plane = rows * cols;
    for (size_t page = 0; page < pages; ++page)
        for (size_t col = 0; col < cols; ++col)
            for (size_t row = 0; row < rows; ++row) {
                const size_t i = page * plane + col * rows + row;
                const long long r = (long long)row, c = (long long)col;
                const int zero = upper ? (c - r < offset) : (r - c < -offset);
                out[i] = zero ? 0.0 : data[i];
            }
}


/* ---- order hooks: cummin / cummax, diff, median, sort (the CPU paths the GPU builtins fall back to) ---- */

/* cummin_tensor / cummax_tensor, runmat-runtime builtins/math/reduction/cummin.rs:719-876 (cummax.rs mirrors it with `>`):
 * running extreme along dim with the 1-based position of its FIRST occurrence in scan order; include-NaN: from the first NaN on the value
 * is NaN and the index is that NaN's position; omit: NaNs are skipped, and before any number has been seen value and index are NaN. */
ORC_API void orc_cumextreme(const double* data, size_t pre, size_t len, size_t post, int is_max, int reverse, int omit, double* values,
                            double* indices) {
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before) {
            double current = 0.0;
            size_t current_index = 0, nan_index = 0;
            int has_value = 0, nan_fixed = 0;
            for (size_t s = 0; s < len; ++s) {
                const size_t k = reverse ? len - 1 - s : s;
                const size_t idx = after * pre * len + before + k * pre;
                const double value = data[idx];
                const size_t position = k + 1;
                if (!omit) {
                    if (nan_fixed) {
                        values[idx] = NAN;
                        indices[idx] = (double)nan_index;
                        continue;
                    }
                    if (value != value) {
                        nan_fixed = 1;
                        nan_index = position;
                        values[idx] = NAN;
                        indices[idx] = (double)position;
                        continue;
                    }
                } else if (value != value) {
                    values[idx] = has_value ? current : NAN;
                    indices[idx] = has_value ? (double)current_index : NAN;
                    continue;
                }
                if (!has_value || (is_max ? value > current : value < current)) {
                    has_value = 1;
                    current = value;
                    current_index = position;
                }
                values[idx] = current;
                indices[idx] = (double)current_index;
            }
        }
}

/* diff_tensor_once, builtins/math/reduction/diff.rs:474-507: out.push order is (after, before, k) - k FASTEST - whatever the dimension:
 * for dim >= 2 with leading dimensions > 1 the output buffer is therefore NOT the column-major array of the differences (the wgpu shader
 * shaders/diff.rs:26-45 writes the same order).  column_major != 0 gives the column-major layout instead (k at stride pre). */
ORC_API void orc_diff_once(const double* data, size_t pre, size_t len, size_t post, int column_major, double* out) {
    if (len <= 1) return;
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before)
            for (size_t k = 0; k + 1 < len; ++k) {
                const size_t idx0 = before + after * pre * len + k * pre;
                const double d = data[idx0 + pre] - data[idx0];
                if (column_major) out[before + after * pre * (len - 1) + k * pre] = d;
                else out[(after * pre + before) * (len - 1) + k] = d;
            }
}

/* stable insertion-free merge sort of (key order given by cmp) used by both restatements: Rust's slice::sort_by is stable */
typedef struct { size_t k; double v; } orc_pair;
static int orc_cmp_mode_desc, orc_cmp_mode_abs;
static int orc_sort_cmp(double a, double b) { /* compare_real_values, sorting_sets/sort.rs:538-574; <0: a first */
    const int an = a != a, bn = b != b;
    if (an && bn) return 0;
    if (an) return orc_cmp_mode_desc ? -1 : 1;
    if (bn) return orc_cmp_mode_desc ? 1 : -1;
    if (orc_cmp_mode_abs) {
        const double fa = fabs(a), fb = fabs(b);
        if (fa != fb) {
            const int c = fa < fb ? -1 : 1;
            return orc_cmp_mode_desc ? -c : c;
        }
    }
    if (a == b) return 0;
    if (orc_cmp_mode_desc) return b < a ? -1 : 1;
    return a < b ? -1 : 1;
}
static void orc_merge_sort(orc_pair* a, orc_pair* tmp, size_t n) {
    if (n < 2) return;
    const size_t h = n / 2;
    orc_merge_sort(a, tmp, h);
    orc_merge_sort(a + h, tmp, n - h);
    size_t i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = orc_sort_cmp(a[j].v, a[i].v) < 0 ? a[j++] : a[i++];  /* ties: the left (earlier) one first */
    while (i < h) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof(orc_pair));
}

/* sort_real_tensor, sorting_sets/sort.rs:413-468: every line along dim sorted stably; indices = 1-based original positions */
ORC_API void orc_sort_dim(const double* data, size_t pre, size_t len, size_t post, int descend, int by_abs, double* sorted, double* indices) {
    orc_pair* buf = (orc_pair*)malloc((len ? len : 1) * sizeof(orc_pair));
    orc_pair* tmp = (orc_pair*)malloc((len ? len : 1) * sizeof(orc_pair));
    orc_cmp_mode_desc = descend;
    orc_cmp_mode_abs = by_abs;
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before) {
            for (size_t k = 0; k < len; ++k) {
                buf[k].k = k;
                buf[k].v = data[before + k * pre + after * pre * len];
            }
            orc_merge_sort(buf, tmp, len);
            for (size_t pos = 0; pos < len; ++pos) {
                const size_t target = before + pos * pre + after * pre * len;
                sorted[target] = buf[pos].v;
                indices[target] = (double)(buf[pos].k + 1);
            }
        }
    free(buf);
    free(tmp);
}

/* reduce_tensor_median_dim (include-NaN: the only mode the provider hook is called with), builtins/math/reduction/median.rs:644-741:
 * a NaN in the slice -> NaN; else the slice sorted by partial_cmp (stable), middle element or 0.5 * (lower + upper).  len == 0 -> NaN. */
ORC_API void orc_median_dim(const double* data, size_t pre, size_t len, size_t post, double* out) {
    orc_pair* buf = (orc_pair*)malloc((len ? len : 1) * sizeof(orc_pair));
    orc_pair* tmp = (orc_pair*)malloc((len ? len : 1) * sizeof(orc_pair));
    orc_cmp_mode_desc = 0;
    orc_cmp_mode_abs = 0;
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before) {
            int saw_nan = 0;
            for (size_t k = 0; k < len; ++k) {
                buf[k].k = k;
                buf[k].v = data[before + k * pre + after * pre * len];
                if (buf[k].v != buf[k].v) saw_nan = 1;
            }
            double m = NAN;
            if (!saw_nan && len > 0) {
                orc_merge_sort(buf, tmp, len);
                m = (len % 2 == 1) ? buf[len / 2].v : 0.5 * (buf[len / 2 - 1].v + buf[len / 2].v);
            }
            out[after * pre + before] = m;
        }
    free(buf);
    free(tmp);
}

/* ---- small linear-algebra / construction hooks ---- */

/* diag_from_vector_sized, simple_provider.rs:3241-3281: element idx of the vector goes to (idx, idx + offset) or (idx - offset, idx) when
 * that lies inside rows x cols; everything else is zero */
ORC_API void orc_diag_from_vector(const double* v, size_t len, long long offset, size_t rows, size_t cols, double* out) {
    for (size_t i = 0; i < rows * cols; ++i) out[i] = 0.0;
    for (size_t idx = 0; idx < len; ++idx) {
        const size_t row = offset >= 0 ? idx : idx + (size_t)(-offset);
        const size_t col = offset >= 0 ? idx + (size_t)offset : idx;
        if (row < rows && col < cols) out[row + col * rows] = v[idx];
    }
}

/* kron_tensor, builtins/array/shape/kron.rs:358-382, 415-485: shapes padded with ones to a common rank; out coordinate = a * extent_b + b */
ORC_API void orc_kron(const double* a, const size_t* shape_a, const double* b, const size_t* shape_b, size_t rank, double* out) {
    size_t na = 1, nb = 1, stride[16], cur = 1;
    for (size_t d = 0; d < rank; ++d) {
        na *= shape_a[d];
        nb *= shape_b[d];
        stride[d] = cur;
        cur *= shape_a[d] * shape_b[d];
    }
    for (size_t ia = 0; ia < na; ++ia)
        for (size_t ib = 0; ib < nb; ++ib) {
            size_t ra = ia, rb = ib, o = 0;
            for (size_t d = 0; d < rank; ++d) {
                const size_t ca = ra % shape_a[d], cb = rb % shape_b[d];
                ra /= shape_a[d];
                rb /= shape_b[d];
                o += (ca * shape_b[d] + cb) * stride[d];
            }
            out[o] = a[ia] * b[ib];
        }
}

/* cross_real_tensor, builtins/math/linalg/ops/cross.rs:332-364: the three components sit `pre` apart */
ORC_API void orc_cross(const double* a, const double* b, size_t pre, size_t post, double* out) {
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before) {
            const size_t i1 = after * pre * 3 + before, i2 = i1 + pre, i3 = i2 + pre;
            out[i1] = a[i2] * b[i3] - a[i3] * b[i2];
            out[i2] = a[i3] * b[i1] - a[i1] * b[i3];
            out[i3] = a[i1] * b[i2] - a[i2] * b[i1];
        }
}

/* gradient_real_tensor_host_with_spacing, builtins/math/reduction/gradient.rs:650-720, 814-833: one-sided differences at the ends,
 * central ones inside; the denominator is the scalar spacing (doubled inside) or coordinate differences.  coords == NULL: scalar. */
ORC_API void orc_gradient(const double* x, size_t pre, size_t len, size_t post, double spacing, const double* coords, double* out) {
    for (size_t i = 0; i < pre * len * post; ++i) out[i] = 0.0;
    if (len <= 1) return;
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before)
            for (size_t k = 0; k < len; ++k) {
                const size_t idx = after * pre * len + before + k * pre;
                double den;
                if (coords) den = k == 0 ? coords[1] - coords[0] : (k + 1 == len ? coords[len - 1] - coords[len - 2] : coords[k + 1] - coords[k - 1]);
                else den = (k == 0 || k + 1 == len) ? spacing : 2.0 * spacing;
                if (k == 0) out[idx] = (x[idx + pre] - x[idx]) / den;
                else if (k + 1 == len) out[idx] = (x[idx] - x[idx - pre]) / den;
                else out[idx] = (x[idx + pre] - x[idx - pre]) / den;
            }
}

/* is_symmetric_real + real_within, builtins/math/linalg/structure/issymmetric.rs:461-487, 517-526 (rows != cols -> false: :546-548) */
ORC_API int orc_issymmetric(const double* data, size_t rows, size_t cols, int skew, double tol) {
    if (rows != cols) return 0;
    for (size_t col = 0; col < cols; ++col) {
        if (skew) {
            const double d = data[col + col * rows];
            if (!(d == 0.0) && (!isfinite(d) || !(fabs(d - 0.0) <= tol))) return 0;
        }
        for (size_t row = 0; row < col; ++row) {
            const double v = data[row + col * rows], r = skew ? -data[col + row * rows] : data[col + row * rows];
            if (v == r) continue;
            if (!isfinite(v) || !isfinite(r)) return 0;
            if (!(fabs(v - r) <= tol)) return 0;
        }
    }
    return 1;
}

/* moving-window statistics over count windows: `moving_real` + `count_window` + `reduce_real_window(_with_fill)` + `RealVarianceAccumulator`
 * + `median(_with_fill)`, builtins/math/reduction/moving.rs:737-825, 1441-1480, 929-1003, 1198-1237, 1282-1323.  op: 0 sum 1 mean 2 prod 3 min
 * 4 max 5 median 6 std 7 var; endpoints: 0 shrink 1 discard 2 fill(fill); nan_mode: 0 include 1 omit; normalization: 0 sample 1 population.
 * `Iterator::sum` of f64 folds from -0.0 on the reference's toolchain (1.90; since 1.83), `product` from 1.0; min / max fold from +-inf with
 * `f64::min / max`.  pre / len / post: the input's extents around the dimension; out_len: the output's extent there. */
static int orc_cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static double orc_kth_fill(const double* sorted, size_t n, double fill, size_t fill_count, size_t k) {
    size_t less = 0;
    while (less < n && sorted[less] < fill) ++less;
    size_t equal = 0;
    while (less + equal < n && sorted[less + equal] == fill) ++equal;
    if (k < less) return sorted[k];
    if (k < less + equal + fill_count) return fill;
    return sorted[k - fill_count];
}
ORC_API void orc_moving_window(const double* x, size_t pre, size_t len, size_t post, size_t out_len, size_t before, size_t after, int op, int endpoints,
                               double fill, int nan_mode, int normalization, double* out) {
    double* vals = (double*)malloc((before + after + 2) * sizeof(double));
    for (size_t o = 0; o < post; ++o)
        for (size_t p = 0; p < out_len; ++p) {
            const long long center = (long long)(endpoints == 1 ? p + before : p);
            const long long start = center - (long long)before, end = center + (long long)after;
            long long s0 = start < 0 ? 0 : start, e0 = end + 1;
            if (s0 > (long long)len) s0 = (long long)len;
            if (e0 < 0) e0 = 0;
            if (e0 > (long long)len) e0 = (long long)len;
            size_t fill_count = 0;
            if (endpoints == 2) fill_count = (size_t)(start < 0 ? -start : 0) + (size_t)(end >= (long long)len ? end - (long long)len + 1 : 0);
            for (size_t i = 0; i < pre; ++i) {
                size_t n = 0;
                int saw_nan = 0;
                for (long long pos = s0; pos < e0; ++pos) {
                    const double v = x[i + (size_t)pos * pre + o * pre * len];
                    if (isnan(v)) {
                        if (nan_mode == 0) {
                            saw_nan = 1;
                            break;
                        }
                        continue;
                    }
                    vals[n++] = v;
                }
                size_t fc = fill_count;
                double r;
                if (fc && (saw_nan || (isnan(fill) && nan_mode == 0))) {
                    r = NAN;
                } else {
                    if (fc && isnan(fill)) fc = 0, saw_nan = 0;  /* omitted NaN padding: reduce the values alone */
                    if (saw_nan) {
                        r = NAN;
                    } else if (n == 0 && fc == 0) {
                        r = op == 0 ? 0.0 : (op == 2 ? 1.0 : NAN);
                    } else {
                        double sum = -0.0, prod = 1.0, mn = INFINITY, mx = -INFINITY;
                        for (size_t k = 0; k < n; ++k) {
                            sum = sum + vals[k];
                            prod = prod * vals[k];
                            mn = fmin(mn, vals[k]);
                            mx = fmax(mx, vals[k]);
                        }
                        if (op == 0) r = fc ? sum + fill * (double)fc : sum;
                        else if (op == 1) r = fc ? (sum + fill * (double)fc) / (double)(n + fc) : sum / (double)n;
                        else if (op == 2) r = fc ? prod * pow(fill, (double)fc) : prod;
                        else if (op == 3) r = fc ? fmin(mn, fill) : mn;
                        else if (op == 4) r = fc ? fmax(mx, fill) : mx;
                        else if (op == 5) {
                            qsort(vals, n, sizeof(double), orc_cmp_double);
                            const size_t total = n + fc, mid = total / 2;
                            if (fc == 0) r = total % 2 ? vals[mid] : (vals[mid - 1] + vals[mid]) / 2.0;
                            else r = total % 2 ? orc_kth_fill(vals, n, fill, fc, mid) : (orc_kth_fill(vals, n, fill, fc, mid - 1) + orc_kth_fill(vals, n, fill, fc, mid)) / 2.0;
                        } else {
                            size_t cnt = 0;
                            double mean = 0.0, m2 = 0.0;
                            for (size_t k = 0; k < n; ++k) {
                                cnt += 1;
                                const double d = vals[k] - mean;
                                mean = mean + d / (double)cnt;
                                const double d2 = vals[k] - mean;
                                m2 = m2 + d * d2;
                            }
                            if (fc) {
                                if (cnt == 0) {
                                    cnt = fc, mean = fill, m2 = 0.0;
                                } else {
                                    const size_t total = cnt + fc;
                                    const double d = fill - mean;
                                    mean = mean + d * ((double)fc / (double)total);
                                    m2 = m2 + d * d * ((double)cnt * (double)fc / (double)total);
                                    cnt = total;
                                }
                            }
                            const double den = normalization == 1 ? (double)cnt : (cnt > 1 ? (double)(cnt - 1) : (double)cnt);
                            r = m2 / den;
                            if (op == 6) r = sqrt(r);
                        }
                    }
                }
                out[i + p * pre + o * pre * out_len] = r;
            }
        }
    free(vals);
}

/* conv (1-D): `convolve` + `apply_mode`, builtins/math/signal/conv.rs:481-517 (the same loops on real data in the in-process provider,
 * simple_provider.rs:1789-1842): out[i + j] += a[i] * b[j] over i ascending then j ascending, from zeros - so output n receives its terms in
 * order of increasing i, each product rounded before it is added.  mode 0 full, 1 same (start (len_b - 1) / 2, len_a points), 2 valid
 * (start len_b - 1, len_a - len_b + 1 points; none when len_a < len_b).  Returns the number of points written to `out`. */
ORC_API size_t orc_conv1d(const double* a, size_t la, const double* b, size_t lb, int mode, double* out) {
    if (la == 0 || lb == 0) return 0;
    const size_t full = la + lb - 1;
    double* f = (double*)calloc(full, sizeof(double));
    for (size_t i = 0; i < la; ++i)
        for (size_t j = 0; j < lb; ++j) {
            const double p = a[i] * b[j];
            f[i + j] = f[i + j] + p;
        }
    size_t start = 0, len = full;
    if (mode == 1) start = (lb - 1) / 2, len = la;
    if (mode == 2) {
        if (la < lb) {
            free(f);
            return 0;
        }
        start = lb - 1, len = la - lb + 1;
    }
    if (start + len > full) len = full > start ? full - start : 0;
    memcpy(out, f + start, len * sizeof(double));
    free(f);
    return len;
}

/* conv2: `conv2_matrices`, builtins/math/signal/conv2.rs:595-640 - full[ar + br, ac + bc] += a[ar, ac] * b[B_r - 1 - br, B_c - 1 - bc] over
 * ac, ar, bc, br ascending (the reference's loops, kernel indexing included: its `conv2_same_flips_kernel` test pins it), then the 'same'
 * ((B - 1) / 2 offsets, a's shape) or 'valid' (B - 1 offsets) window.  Column-major; writes out_rows x out_cols, returns their product. */
ORC_API size_t orc_conv2d(const double* a, size_t ar_n, size_t ac_n, const double* b, size_t br_n, size_t bc_n, int mode, double* out, size_t* out_rows,
                          size_t* out_cols) {
    *out_rows = *out_cols = 0;
    if (ar_n == 0 || ac_n == 0 || br_n == 0 || bc_n == 0) {
        if (mode == 1) *out_rows = ar_n, *out_cols = ac_n;
        return 0;
    }
    const size_t fr = ar_n + br_n - 1, fc = ac_n + bc_n - 1;
    double* f = (double*)calloc(fr * fc, sizeof(double));
    for (size_t ac = 0; ac < ac_n; ++ac)
        for (size_t ar = 0; ar < ar_n; ++ar) {
            const double av = a[ac * ar_n + ar];
            for (size_t bc = 0; bc < bc_n; ++bc)
                for (size_t br = 0; br < br_n; ++br) {
                    const double p = av * b[(bc_n - 1 - bc) * br_n + (br_n - 1 - br)];
                    double* d = f + (ac + bc) * fr + (ar + br);
                    *d = *d + p;
                }
        }
    size_t r0 = 0, c0 = 0, rows = fr, cols = fc;
    if (mode == 1) r0 = (br_n - 1) / 2, c0 = (bc_n - 1) / 2, rows = ar_n, cols = ac_n;
    if (mode == 2) {
        if (ar_n < br_n || ac_n < bc_n) {
            free(f);
            return 0;
        }
        r0 = br_n - 1, c0 = bc_n - 1, rows = ar_n - br_n + 1, cols = ac_n - bc_n + 1;
    }
    for (size_t c = 0; c < cols; ++c)
        for (size_t r = 0; r < rows; ++r) out[c * rows + r] = f[(c0 + c) * fr + (r0 + r)];
    free(f);
    *out_rows = rows, *out_cols = cols;
    return rows * cols;
}

/* hann / hamming / blackman windows: generate_window_data, runmat-accelerate/src/simple_provider.rs:95-120 (kind 0, 1, 2): len 0 -> empty,
 * 1 -> [1]; otherwise phase = 2 pi idx / (L - 1) with L = len + 1 when periodic (the extra point is dropped). */
ORC_API void orc_window(int kind, size_t len, int periodic, double* out) {
    if (len == 0) return;
    if (len == 1) {
        out[0] = 1.0;
        return;
    }
    const double denom = (double)((periodic ? len + 1 : len) - 1);
    for (size_t i = 0; i < len; ++i) {
        const double phase = 2.0 * 3.14159265358979323846 * (double)i / denom;
        if (kind == 0) out[i] = 0.5 - 0.5 * cos(phase);
        else if (kind == 1) out[i] = 0.54 - 0.46 * cos(phase);
        else out[i] = 0.42 - 0.5 * cos(phase) + 0.08 * cos(2.0 * phase);
    }
}

/* fft_dim / ifft_dim: the reference transforms each line with rustfft 6.4.1 (`FftPlanner::plan_fft_forward/inverse(len).process`,
 * builtins/math/fft/common.rs and the wgpu provider's host form ops/fft/fallback.rs:84-125) - a third-party crate that is not under
 * /root/reference (Cargo.lock:6116-6118).  What it computes is the discrete Fourier transform X[j] = sum_k x[k] exp(-+2 pi i jk / n);
 * restated here by DIRECT evaluation in long double (the angle index jk is reduced mod n in integers), with the reference's framing:
 * lines along a dimension of a column-major tensor (inner = product of the leading extents), zero-padded or truncated to n points
 * (fallback.rs:36-47, 100-108), the inverse scaled by 1 / n (:113-118), complex-interleaved output (:127-131).  Bit-level parity is
 * unpinned (rustfft's operation order is its own); the reference's tests pin 1e-12 / 1e-10 absolute on small vectors. */
ORC_API void orc_dft_dim(const double* in, int in_complex, size_t inner, size_t len_in, size_t outer, size_t n, int inverse, double* out) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    long double* wr = (long double*)malloc((n ? n : 1) * sizeof(long double));
    long double* wi = (long double*)malloc((n ? n : 1) * sizeof(long double));
    for (size_t k = 0; k < n; ++k) {
        const long double a = two_pi * (long double)k / (long double)n;
        wr[k] = cosl(a);
        wi[k] = inverse ? sinl(a) : -sinl(a);
    }
    const size_t copy = len_in < n ? len_in : n;
    for (size_t o = 0; o < outer; ++o)
        for (size_t i = 0; i < inner; ++i)
            for (size_t j = 0; j < n; ++j) {
                long double sr = 0.0L, si = 0.0L;
                for (size_t k = 0; k < copy; ++k) {
                    const size_t src = i + inner * (k + len_in * o);
                    const long double xr = in_complex ? in[2 * src] : in[src], xi = in_complex ? in[2 * src + 1] : 0.0;
                    const size_t t = (size_t)(((unsigned __int128)j * k) % n);
                    sr += xr * wr[t] - xi * wi[t];
                    si += xr * wi[t] + xi * wr[t];
                }
                if (inverse) sr /= (long double)n, si /= (long double)n;
                const size_t dst = i + inner * (j + n * o);
                out[2 * dst] = (double)sr;
                out[2 * dst + 1] = (double)si;
            }
    free(wr);
    free(wi);
}

/* is_hermitian_real, ishermitian.rs:455-482 (real_within :522-530; rows != cols -> false :550-552) */
ORC_API int orc_ishermitian(const double* data, size_t rows, size_t cols, int skew, double tol) {
    if (rows != cols) return 0;
    for (size_t col = 0; col < cols; ++col) {
        const double d = data[col + col * rows];
        if (isnan(d)) return 0;
        if (skew && !(d == 0.0) && (!isfinite(d) || !(fabs(d - 0.0) <= tol))) return 0;
        for (size_t row = 0; row < col; ++row) {
            const double v = data[row + col * rows], r = skew ? -data[col + row * rows] : data[col + row * rows];
            if (v == r) continue;
            if (!isfinite(v) || !isfinite(r)) return 0;
            if (!(fabs(v - r) <= tol)) return 0;
        }
    }
    return 1;
}

/* compute_real_bandwidth, bandwidth.rs:341-365 */
ORC_API void orc_bandwidth(const double* data, size_t rows, size_t cols, size_t* lower, size_t* upper) {
    *lower = *upper = 0;
    if (rows == 0 || cols == 0) return;
    for (size_t col = 0; col < cols; ++col)
        for (size_t row = 0; row < rows; ++row) {
            const double v = data[row + col * rows];
            if (v != 0.0 || isnan(v)) {
                if (row >= col) { if (row - col > *lower) *lower = row - col; }
                else if (col - row > *upper) *upper = col - row;
            }
        }
}

/* inv: the reference calls nalgebra 0.32.6 `DMatrix::try_inverse` (inv.rs:224-228) - a third-party dependency that is not under
 * /root/reference (Cargo.lock pins it).  Its published algorithm for dynamic sizes: LU with partial (row) pivoting, pivot = the entry
 * of largest magnitude in the column, `None` when a pivot is exactly zero, then the inverse by substitutions on the permuted identity.
 * Restated here; parity of the solution is by residual / forward error, as for mldivide (bit-level parity unpinned).
 * Returns 0, or 1 when a pivot is exactly zero ("matrix is singular to working precision"). */
ORC_API int orc_inv(const double* a, size_t n, double* out) {
    double* lu = (double*)malloc(n * n * sizeof(double));
    size_t* perm = (size_t*)malloc(n * sizeof(size_t));
    memcpy(lu, a, n * n * sizeof(double));
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    int singular = 0;
    for (size_t k = 0; k < n && !singular; ++k) {
        size_t p = k;
        double best = fabs(lu[k + k * n]);
        for (size_t i = k + 1; i < n; ++i)
            if (fabs(lu[i + k * n]) > best) {
                best = fabs(lu[i + k * n]);
                p = i;
            }
        if (best == 0.0) {
            singular = 1;
            break;
        }
        if (p != k) {
            for (size_t j = 0; j < n; ++j) {
                const double t = lu[k + j * n];
                lu[k + j * n] = lu[p + j * n];
                lu[p + j * n] = t;
            }
            const size_t t = perm[k];
            perm[k] = perm[p];
            perm[p] = t;
        }
        const double piv = lu[k + k * n];
        for (size_t i = k + 1; i < n; ++i) lu[i + k * n] /= piv;
        for (size_t j = k + 1; j < n; ++j) {
            const double u = lu[k + j * n];
            for (size_t i = k + 1; i < n; ++i) lu[i + j * n] -= lu[i + k * n] * u;
        }
    }
    if (!singular)
        for (size_t c = 0; c < n; ++c) {
            double* x = out + c * n;
            for (size_t i = 0; i < n; ++i) x[i] = perm[i] == c ? 1.0 : 0.0;  /* P e_c */
            for (size_t i = 0; i < n; ++i)
                for (size_t j = 0; j < i; ++j) x[i] -= lu[i + j * n] * x[j];
            for (size_t ii = n; ii-- > 0;) {
                for (size_t j = ii + 1; j < n; ++j) x[ii] -= lu[ii + j * n] * x[j];
                x[ii] /= lu[ii + ii * n];
            }
        }
    free(lu);
    free(perm);
    return singular;
}

/* find (builtins/array/indexing/find.rs:593-633; simple_provider.rs:7500-7575): linear indices (1-based) of the elements != 0 (a NaN is
 * nonzero) in ascending order (direction first) or DESCENDING order from the end (last), at most `cap` of them; rows / cols from the
 * first extent.  Returns the number found (<= cap). */
ORC_API size_t orc_find(const double* data, size_t len, size_t row_extent, size_t cap, int last, double* linear, double* rows, double* cols,
                        double* values) {
    size_t n = 0;
    if (row_extent == 0) row_extent = 1;
    for (size_t s = 0; s < len && n < cap; ++s) {
        const size_t idx = last ? len - 1 - s : s;
        if (data[idx] != 0.0) {
            linear[n] = (double)(idx + 1);
            rows[n] = (double)(idx % row_extent + 1);
            cols[n] = (double)(idx / row_extent + 1);
            values[n] = data[idx];
            ++n;
        }
    }
    return n;
}

/* simple_trapezoid, simple_provider.rs:2421-2598 (the runtime's trapz.rs / cumtrapz.rs host loops are the same): per line
 * acc += 0.5 * width * (x[k] + x[k+1]) in order, width = 1, a scalar, coords[k+1] - coords[k] (vector) or the difference of the spacing
 * tensor's neighbours; cumulative writes acc at k + 1 (0 at k = 0), otherwise one value per line.  kind: 0 unit, 1 scalar, 3 vector, 4 tensor. */
ORC_API void orc_trapz(const double* x, size_t pre, size_t len, size_t post, int kind, double scalar, const double* sp, int cumulative, double* out) {
    if (cumulative) for (size_t i = 0; i < pre * len * post; ++i) out[i] = 0.0;
    else for (size_t i = 0; i < pre * post; ++i) out[i] = 0.0;
    for (size_t after = 0; after < post; ++after)
        for (size_t before = 0; before < pre; ++before) {
            double acc = 0.0;
            for (size_t k = 0; k + 1 < len; ++k) {
                const size_t i0 = after * pre * len + before + k * pre, i1 = i0 + pre;
                const double w = kind == 0 ? 1.0 : (kind == 1 ? scalar : (kind == 3 ? sp[k + 1] - sp[k] : sp[i1] - sp[i0]));
                acc += 0.5 * w * (x[i0] + x[i1]);
                if (cumulative) out[i1] = acc;
            }
            if (!cumulative) out[after * pre + before] = acc;
        }
}

/* chol_factor for real data, builtins/math/linalg/factor/chol.rs:374-433 (Cholesky-Crout on a row-major copy; here column-major in and
 * out): per column j first the symmetry of every pair (i, j), i < j, to 1e-12 max(|a|, |b|, 1); then r_ij = (a_ij - sum_k r_ki r_kj) /
 * r_ii in order k = 0 .. i-1, r_jj = sqrt of a positive finite sum.  info = 1-based column of the first failure (a tiny denominator
 * reports ITS row); the rows from info - 1 on are zeroed.  Returns info (0 = success); `upper` is n x n column-major. */
ORC_API unsigned orc_chol(const double* a, size_t n, double* upper) {
    const double EPS = 1.0e-12;
    size_t info = 0;
    for (size_t i = 0; i < n * n; ++i) upper[i] = 0.0;
    for (size_t j = 0; j < n && !info; ++j) {
        for (size_t i = 0; i < j; ++i) {
            const double x = a[i + j * n], y = a[j + i * n];
            const double scale = fmax(fmax(fabs(x), fabs(y)), 1.0);
            if (!(fabs(x - y) <= EPS * scale)) {
                info = j + 1;
                break;
            }
        }
        if (info) break;
        for (size_t i = 0; i <= j; ++i) {
            double sum = a[i + j * n];
            for (size_t k = 0; k < i; ++k) sum -= upper[k + i * n] * upper[k + j * n];
            if (i == j) {
                if (!isfinite(sum) || sum <= 0.0) {
                    info = j + 1;
                    break;
                }
                upper[i + i * n] = sqrt(sum);
            } else {
                const double den = upper[i + i * n];
                if (fabs(den) <= EPS) {
                    info = i + 1;
                    break;
                }
                upper[i + j * n] = sum / den;
            }
        }
    }
    if (info)
        for (size_t row = info - 1; row < n; ++row)
            for (size_t col = row; col < n; ++col) upper[row + col * n] = 0.0;
    return (unsigned)info;
}

/* norm of real data, builtins/math/linalg/norm.rs:269-282, 320-380, 381-411, 413-452, 497-529.  kind: 0 vector, 1 matrix (rows x cols).
 * order: 1 one, 2 two, 3 inf, 4 -inf, 5 zero, 6 fro, 7 nuc, 8 P(p).  Returns NaN for what the CPU refuses or computes with an SVD
 * (matrix 2-norm, nuclear norm, vector p < 1): *refused is set for those. */
static double orc_rss(const double* v, size_t n) { /* root_sum_of_squares, norm.rs:381-411 */
    double scale = 0.0, sumsq = 1.0;
    size_t count = 0;
    for (size_t i = 0; i < n; ++i) {
        const double a = fabs(v[i]);
        if (a != a) return NAN;
        if (isinf(a)) return INFINITY;
        if (a == 0.0) continue;
        if (scale < a) {
            const double ratio = scale == 0.0 ? 0.0 : scale / a;
            sumsq = 1.0 + sumsq * ratio * ratio;
            scale = a;
        } else {
            const double ratio = a / scale;
            sumsq += ratio * ratio;
        }
        ++count;
    }
    return count == 0 ? 0.0 : scale * sqrt(sumsq);
}
ORC_API double orc_norm(const double* data, size_t rows, size_t cols, int is_matrix, int order, double p, int* refused) {
    const size_t n = rows * cols;
    *refused = 0;
    for (size_t i = 0; i < n; ++i)
        if (data[i] != data[i]) return NAN;  /* any NaN: NaN (norm.rs:321-323, 419-421) */
    if (!is_matrix) {
        if (order == 1) { double s = 0.0; for (size_t i = 0; i < n; ++i) s += fabs(data[i]); return s; }
        if (order == 2 || order == 6) return orc_rss(data, n);
        if (order == 3) { double m = 0.0; for (size_t i = 0; i < n; ++i) if (fabs(data[i]) > m) m = fabs(data[i]); return m; }
        if (order == 4) {
            if (n == 0) return 0.0;
            double m = INFINITY;
            for (size_t i = 0; i < n; ++i) if (fabs(data[i]) < m) m = fabs(data[i]);
            return m == INFINITY ? 0.0 : m;
        }
        if (order == 5) { double c = 0.0; for (size_t i = 0; i < n; ++i) if (fabs(data[i]) != 0.0) c += 1.0; return c; }
        if (order == 8) {
            if (!isfinite(p) || p < 1.0) { *refused = 1; return NAN; }
            if (n == 0) return 0.0;
            double s = 0.0;
            for (size_t i = 0; i < n; ++i) s += pow(fabs(data[i]), p);
            return pow(s, 1.0 / p);
        }
        *refused = 1;
        return NAN;
    }
    if (rows == 0 || cols == 0) return 0.0;
    if (order == 1) { /* max_column_sum */
        double best = 0.0;
        for (size_t c = 0; c < cols; ++c) {
            double s = 0.0;
            for (size_t r = 0; r < rows; ++r) s += fabs(data[r + c * rows]);
            if (s > best) best = s;
        }
        return best;
    }
    if (order == 3) { /* max_row_sum */
        double best = 0.0;
        for (size_t r = 0; r < rows; ++r) {
            double s = 0.0;
            for (size_t c = 0; c < cols; ++c) s += fabs(data[r + c * rows]);
            if (s > best) best = s;
        }
        return best;
    }
    if (order == 6) return orc_rss(data, n);
    *refused = 1;
    return NAN;
}
