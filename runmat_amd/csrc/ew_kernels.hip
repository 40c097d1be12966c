// ew_kernels.hip -- ahead-of-time gfx950 kernels for the per-op provider hooks
//   elem_add/sub/mul/div/pow/max/min/hypot/atan2   crates/runmat-accelerate-api/src/lib.rs:1890-1938,1979,2069
//   unary_*                                        lib.rs:2077-2331
//   scalar_*                                       lib.rs:2333-2355
//   zeros/ones/fill                                lib.rs:1468-1522
// These are HBM-bound streaming kernels: 16-byte accesses per lane (1 KiB per wave instruction),
// several independent vectors in flight per thread, grid capped at 8 blocks per CU with a
// grid-stride loop. Arithmetic follows the CPU builtins (see skel_common.h), compiled with
// -ffp-contract=off.
#include "common.h"
#include "skel_common.h"

namespace rmhip {

typedef double v2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 16-byte vector of the storage type T; arithmetic is always f64 (see Buffer::dtype, common.h)
template <class T>
struct VecOf;
template <>
struct VecOf<double> {
    typedef v2 type;
    static constexpr int N = 2;
};
template <>
struct VecOf<float> {
    typedef v4f type;
    static constexpr int N = 4;
};

static constexpr int kBlock = 256;    // broadcast kernel: a block spans kBlock elements of dim 0
static constexpr int kBcastE = 4;     // elements per thread in the broadcast kernel (their loads overlap); 8 for f32 storage
template <class T>
struct BcastE {
    static constexpr int value = sizeof(T) == 4 ? 2 * kBcastE : kBcastE;
};
static constexpr int kStream = 1024;  // streaming kernels: 1024-thread blocks measured 8-20 % faster than 256 (interleaved
                                      // A/B on the generated kernels, scripts/tune_ew_ab.py; same skeleton here)
static constexpr int kUnroll = 1;  // one 16-byte vector per stream per thread (interleaved A/B: unrolling never helped)

static inline unsigned stream_grid(const Context* c, size_t nvec) {
    size_t want = (nvec + (size_t)kStream * kUnroll - 1) / ((size_t)kStream * kUnroll);
    size_t cap = (size_t)c->num_cus * 16;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

// ---- special functions: the CPU builtins' formulas restated (tolerance, not bit parity: pow / exp / log / sin / erfc
// come from ocml here and from libm / num-complex there) ------------------------------------------------------------
// Lanczos g = 7, 9 terms (math/elementwise/gamma.rs:25-38, gammaln.rs:25-39)
__device__ const double kLanczos[8] = {676.5203681218851,   -1259.1392167224028,  771.3234287776531,     -176.6150291621406,
                                       12.507343278686905,  -0.13857109526572012, 9.984369578019572e-6,  1.5056327351493116e-7};
// gamma.rs:353-364
__device__ __forceinline__ bool rm_close_to_integer(double x) {
    if (!rm_isfinite(x)) return false;
    const double nearest = round(x);
    const double diff = fabs(x - nearest);
    return nearest == 0.0 ? diff <= 1e-24 : diff <= 1e-12 * fmax(fabs(nearest), 1.0);
}
// gamma.rs:332-343 for a real argument >= 0.5.  The reference evaluates it in complex arithmetic with zero imaginary
// parts: a complex quotient (c + 0i) / (d + 0i) is (c*d) / (d*d) there (num-complex Div), kept as is.
__device__ __forceinline__ double rm_lanczos_gamma(double z) {
    const double zm1 = z - 1.0;
    double sum = 0.9999999999998099;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double d = zm1 + (double)(i + 1);
        sum += (kLanczos[i] * d) / (d * d);
    }
    const double t = zm1 + 7.5;
    return 2.5066282746310005 * pow(t, zm1 + 0.5) * exp(-t) * sum;
}
// gamma.rs:289-330
__device__ __forceinline__ double rm_gamma(double x) {
    if (rm_isnan(x)) return __builtin_nan("");
    if (rm_isinf(x)) return x > 0.0 ? x : __builtin_nan("");
    if (x <= 0.0 && rm_close_to_integer(x)) return __builtin_inf();
    if (x < 0.5) {  // reflection: pi / (sin(pi x) * gamma(1 - x))
        const double s = sin(M_PI * x);
        if (s * s <= 1e-24) return __builtin_inf();
        const double d = s * rm_lanczos_gamma(1.0 - x);
        return (M_PI * d) / (d * d);
    }
    return rm_lanczos_gamma(x);
}
// gammaln.rs:254-281
__device__ __forceinline__ double rm_lanczos_gammaln(double v) {
    const double zm1 = v - 1.0;
    double sum = 0.9999999999998099;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += kLanczos[i] / (zm1 + (double)(i + 1));
    const double t = zm1 + 7.0 + 0.5;
    return 0.9189385332046727 + (zm1 + 0.5) * log(t) - t + log(sum);
}
__device__ __forceinline__ double rm_gammaln(double v) {
    if (rm_isnan(v)) return __builtin_nan("");
    if (v == 0.0 || v == __builtin_inf()) return __builtin_inf();
    if (v < 0.0) return __builtin_nan("");  // the builtin raises for negative input before it gets here (gammaln.rs:283-292)
    if (v < 1.0e-305) return -log(v);
    if (v < 0.5) return log(M_PI) - log(sin(M_PI * v)) - rm_lanczos_gammaln(1.0 - v);
    return rm_lanczos_gammaln(v);
}
// factorial.rs:25-34, 272-314: n! as the running product 1*2*...*n (each step rounded), NaN unless n is a non-negative
// integer to within eps * max(|n|, 1), Inf beyond 170
__device__ __forceinline__ double rm_factorial(double v) {
    if (rm_isnan(v)) return __builtin_nan("");
    if (v == 0.0) return 1.0;
    if (rm_isinf(v)) return v > 0.0 ? v : __builtin_nan("");
    if (v < 0.0) return __builtin_nan("");
    const double rounded = round(v);
    if (fabs(v - rounded) > 2.220446049250313e-16 * fmax(fabs(v), 1.0)) return __builtin_nan("");
    if (rounded > 170.0) return __builtin_inf();
    double acc = 1.0;
    const int n = (int)rounded;
    for (int i = 2; i <= n; ++i) acc *= (double)i;
    return acc;
}
// nextpow2.rs:157-164
__device__ __forceinline__ double rm_nextpow2(double x) {
    const double ax = fabs(x);
    return ax == 0.0 ? 0.0 : ceil(log2(ax));
}
// erfcinv.rs:261-308
__device__ __forceinline__ double rm_erfcinv_tail(double target) {
    double lo = 0.0, hi = 1.0;
    while (hi < 32.0 && erfc(hi) > target) {
        lo = hi;
        hi *= 2.0;
    }
    if (erfc(hi) > target) return hi;
    for (int i = 0; i < 110; ++i) {
        const double mid = 0.5 * (lo + hi);
        if (erfc(mid) > target) lo = mid;
        else hi = mid;
    }
    return 0.5 * (lo + hi);
}
__device__ __forceinline__ double rm_erfcinv(double v) {
    if (rm_isnan(v) || !(v >= 0.0 && v <= 2.0)) return __builtin_nan("");
    if (v == 0.0) return __builtin_inf();
    if (v == 2.0) return -__builtin_inf();
    if (v == 1.0) return 0.0;
    return v > 1.0 ? -rm_erfcinv_tail(2.0 - v) : rm_erfcinv_tail(v);
}

template <int OP>
__device__ __forceinline__ double unary_op(double v) {
    switch (OP) {
        case RMHIP_SIN: return sin(v);
        case RMHIP_COS: return cos(v);
        case RMHIP_TAN: return tan(v);
        case RMHIP_ASIN: return asin(v);
        case RMHIP_ACOS: return acos(v);
        case RMHIP_ATAN: return atan(v);
        case RMHIP_SINH: return sinh(v);
        case RMHIP_COSH: return cosh(v);
        case RMHIP_TANH: return tanh(v);
        case RMHIP_ASINH: return asinh(v);
        case RMHIP_ACOSH: return acosh(v);
        case RMHIP_ATANH: return atanh(v);
        case RMHIP_EXP: return exp(v);
        case RMHIP_EXPM1: return expm1(v);
        case RMHIP_LOG: return log(v);
        case RMHIP_LOG2: return log2(v);
        case RMHIP_LOG10: return log10(v);
        case RMHIP_LOG1P: return log1p(v);
        case RMHIP_SQRT: return sqrt(v);
        case RMHIP_ABS: return fabs(v);
        case RMHIP_SIGN: return rm_sign(v);
        case RMHIP_FLOOR: return floor(v);
        case RMHIP_CEIL: return ceil(v);
        case RMHIP_ROUND: return round(v);
        case RMHIP_FIX: return trunc(v);
        case RMHIP_NEG: return -v;
        case RMHIP_EXP2: return exp2(v);
        case RMHIP_HEAVISIDE: return rm_heaviside(v);
        case RMHIP_ISNAN: return rm_isnan(v) ? 1.0 : 0.0;
        case RMHIP_ISINF: return rm_isinf(v) ? 1.0 : 0.0;
        case RMHIP_ISFINITE: return rm_isfinite(v) ? 1.0 : 0.0;
        case RMHIP_SINGLE: return rm_f32(v);  // `single`: f64 storage rounded through f32 (runmat-builtins lib.rs:426-436)
        case RMHIP_ERF: return erf(v);        // libm::erf (elementwise/erf.rs:214-216)
        case RMHIP_SINC: return rm_sinc(v);
        case RMHIP_NOT: return v == 0.0 ? 1.0 : 0.0;
        case RMHIP_GAMMA: return rm_gamma(v);
        case RMHIP_FACTORIAL: return rm_factorial(v);
        case RMHIP_NEXTPOW2: return rm_nextpow2(v);
        case RMHIP_GAMMALN: return rm_gammaln(v);
        case RMHIP_ERFCINV: return rm_erfcinv(v);
        case RMHIP_NAN_TO_ZERO: return rm_isnan(v) ? 0.0 : v;  // backend/wgpu/shaders/nan.rs: select(v, 0, v != v)
        case RMHIP_NOT_NAN: return rm_isnan(v) ? 0.0 : 1.0;
        default: return v;
    }
}

template <int OP>
__device__ __forceinline__ double binary_op(double a, double b) {
    switch (OP) {
        case RMHIP_ADD: return a + b;
        case RMHIP_SUB: return a - b;
        case RMHIP_MUL: return a * b;
        case RMHIP_DIV: return a / b;
        case RMHIP_POW: return rm_pow(a, b);  // skel_common.h: exponent 2 -> the exact product
        case RMHIP_MAX: return rm_max(a, b);
        case RMHIP_MIN: return rm_min(a, b);
        case RMHIP_HYPOT: return hypot(a, b);
        case RMHIP_ATAN2: return atan2(a, b);
        case RMHIP_MOD: return rm_mod(a, b);
        case RMHIP_REM: return rm_rem(a, b);
        case RMHIP_EQ: return a == b ? 1.0 : 0.0;
        case RMHIP_NE: return a != b ? 1.0 : 0.0;
        case RMHIP_LT: return a < b ? 1.0 : 0.0;
        case RMHIP_LE: return a <= b ? 1.0 : 0.0;
        case RMHIP_GT: return a > b ? 1.0 : 0.0;
        case RMHIP_GE: return a >= b ? 1.0 : 0.0;
        case RMHIP_AND: return (a != 0.0 && b != 0.0) ? 1.0 : 0.0;
        case RMHIP_OR: return (a != 0.0 || b != 0.0) ? 1.0 : 0.0;
        default: return ((a != 0.0) != (b != 0.0)) ? 1.0 : 0.0;
    }
}

// scalar_*: a op s, with rsub = s - a, rdiv = s ./ a (lib.rs:2333-2338)
template <int OP>
__device__ __forceinline__ double scalar_op(double a, double s) {
    switch (OP) {
        case RMHIP_SADD: return a + s;
        case RMHIP_SSUB: return a - s;
        case RMHIP_SMUL: return a * s;
        case RMHIP_SDIV: return a / s;
        case RMHIP_SRSUB: return s - a;
        case RMHIP_SRDIV: return s / a;
        case RMHIP_SMAX: return rm_max(a, s);
        default: return rm_min(a, s);
    }
}

// ---- streaming skeleton: out[i] = f(i) over 16-byte vectors -------------------------------------
template <class T, class F>
__global__ void __launch_bounds__(kStream) k_stream1(const T* __restrict__ a, T* __restrict__ out, size_t n, F f) {
    typedef typename VecOf<T>::type V;
    constexpr int N = VecOf<T>::N;
    const size_t nvec = n / N;
    const size_t stride = (size_t)gridDim.x * kStream;
    size_t i = (size_t)blockIdx.x * kStream + threadIdx.x;
    const V* __restrict__ av = (const V*)a;
    V* __restrict__ ov = (V*)out;
    for (; i < nvec; i += stride) {
        V x = __builtin_nontemporal_load(av + i), r;
#pragma unroll
        for (int l = 0; l < N; ++l) r[l] = (T)f((double)x[l]);
        __builtin_nontemporal_store(r, ov + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * N) {
        const size_t t = nvec * N + threadIdx.x;
        out[t] = (T)f((double)a[t]);
    }
}

template <class T, class F>
__global__ void __launch_bounds__(kStream) k_stream2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                    size_t n, F f) {
    typedef typename VecOf<T>::type V;
    constexpr int N = VecOf<T>::N;
    const size_t nvec = n / N;
    const size_t stride = (size_t)gridDim.x * kStream;
    size_t i = (size_t)blockIdx.x * kStream + threadIdx.x;
    const V* __restrict__ av = (const V*)a;
    const V* __restrict__ bv = (const V*)b;
    V* __restrict__ ov = (V*)out;
    for (; i < nvec; i += stride) {
        V x = __builtin_nontemporal_load(av + i), y = __builtin_nontemporal_load(bv + i), r;
#pragma unroll
        for (int l = 0; l < N; ++l) r[l] = (T)f((double)x[l], (double)y[l]);
        __builtin_nontemporal_store(r, ov + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * N) {
        const size_t t = nvec * N + threadIdx.x;
        out[t] = (T)f((double)a[t], (double)b[t]);
    }
}

template <int OP>
struct UnaryF {
    __device__ __forceinline__ double operator()(double v) const { return unary_op<OP>(v); }
};
template <int OP>
struct BinaryF {
    __device__ __forceinline__ double operator()(double a, double b) const { return binary_op<OP>(a, b); }
};
template <int OP>
struct ScalarF {
    double s;
    __device__ __forceinline__ double operator()(double a) const { return scalar_op<OP>(a, s); }
};

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Unaligned (externally wrapped) memory: plain element-sized accesses.
template <class T, class F>
__global__ void __launch_bounds__(kStream) k_plain1(const T* __restrict__ a, T* __restrict__ out, size_t n, F f) {
    const size_t stride = (size_t)gridDim.x * kStream;
    for (size_t i = (size_t)blockIdx.x * kStream + threadIdx.x; i < n; i += stride) out[i] = (T)f((double)a[i]);
}
template <class T, class F>
__global__ void __launch_bounds__(kStream) k_plain2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                   size_t n, F f) {
    const size_t stride = (size_t)gridDim.x * kStream;
    for (size_t i = (size_t)blockIdx.x * kStream + threadIdx.x; i < n; i += stride) out[i] = (T)f((double)a[i], (double)b[i]);
}

template <class T, class F>
static int run1(Context* c, const T* a, T* out, size_t n, F f) {
    if (n == 0) return RMHIP_OK;
    if (aligned16(a) && aligned16(out))
        hipLaunchKernelGGL((k_stream1<T, F>), dim3(stream_grid(c, n / VecOf<T>::N)), dim3(kStream), 0, c->stream, a, out, n, f);
    else
        hipLaunchKernelGGL((k_plain1<T, F>), dim3(stream_grid(c, n)), dim3(kStream), 0, c->stream, a, out, n, f);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
template <class T, class F>
static int run2(Context* c, const T* a, const T* b, T* out, size_t n, F f) {
    if (n == 0) return RMHIP_OK;
    if (aligned16(a) && aligned16(b) && aligned16(out))
        hipLaunchKernelGGL((k_stream2<T, F>), dim3(stream_grid(c, n / VecOf<T>::N)), dim3(kStream), 0, c->stream, a, b, out, n, f);
    else
        hipLaunchKernelGGL((k_plain2<T, F>), dim3(stream_grid(c, n)), dim3(kStream), 0, c->stream, a, b, out, n, f);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

template <int OP, class T>
static int unary_dispatch(Context* c, int op, const T* a, T* out, size_t n) {
    if (op == OP) return run1(c, a, out, n, UnaryF<OP>());
    if constexpr (OP + 1 < RMHIP_UNARY_OP_COUNT) return unary_dispatch<OP + 1>(c, op, a, out, n);
    return fail(RMHIP_ERR_UNSUPPORTED, "unary op %d not supported by provider", op);
}
int launch_unary(Context* c, int op, const double* a, double* out, size_t n) {
    return unary_dispatch<0>(c, op, a, out, n);
}
int launch_unary_f32(Context* c, int op, const float* a, float* out, size_t n) {
    return unary_dispatch<0>(c, op, a, out, n);
}

template <int OP, class T>
static int scalar_dispatch(Context* c, int op, const T* a, double s, T* out, size_t n) {
    if (op == OP) return run1(c, a, out, n, ScalarF<OP>{s});
    if constexpr (OP + 1 < RMHIP_SCALAR_OP_COUNT) return scalar_dispatch<OP + 1>(c, op, a, s, out, n);
    return fail(RMHIP_ERR_UNSUPPORTED, "scalar op %d not supported by provider", op);
}
int launch_scalar(Context* c, int op, const double* a, double s, double* out, size_t n) {
    return scalar_dispatch<0>(c, op, a, s, out, n);
}
int launch_scalar_f32(Context* c, int op, const float* a, double s, float* out, size_t n) {
    return scalar_dispatch<0>(c, op, a, s, out, n);
}

template <int OP, class T>
static int binary_same_dispatch(Context* c, int op, const T* a, const T* b, T* out, size_t n) {
    if (op == OP) return run2(c, a, b, out, n, BinaryF<OP>());
    if constexpr (OP + 1 < RMHIP_BINARY_OP_COUNT) return binary_same_dispatch<OP + 1>(c, op, a, b, out, n);
    return fail(RMHIP_ERR_UNSUPPORTED, "binary op %d not supported by provider", op);
}
int launch_binary_same(Context* c, int op, const double* a, const double* b, double* out, size_t n) {
    return binary_same_dispatch<0>(c, op, a, b, out, n);
}
int launch_binary_same_f32(Context* c, int op, const float* a, const float* b, float* out, size_t n) {
    return binary_same_dispatch<0>(c, op, a, b, out, n);
}

// ---- storage conversions (precision 32: widen f32 operands for the f64-only kernels, narrow their results) ----------
__global__ void __launch_bounds__(kStream) k_widen(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
    const size_t nvec = n >> 1, stride = (size_t)gridDim.x * kStream;
    typedef float v2f __attribute__((ext_vector_type(2)));
    for (size_t i = (size_t)blockIdx.x * kStream + threadIdx.x; i < nvec; i += stride) {
        const v2f x = __builtin_nontemporal_load((const v2f*)src + i);
        v2 r = {(double)x.x, (double)x.y};
        __builtin_nontemporal_store(r, (v2*)dst + i);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = (double)src[n - 1];
}
__global__ void __launch_bounds__(kStream) k_narrow(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t nvec = n >> 1, stride = (size_t)gridDim.x * kStream;
    typedef float v2f __attribute__((ext_vector_type(2)));
    for (size_t i = (size_t)blockIdx.x * kStream + threadIdx.x; i < nvec; i += stride) {
        const v2 x = __builtin_nontemporal_load((const v2*)src + i);
        v2f r = {(float)x.x, (float)x.y};
        __builtin_nontemporal_store(r, (v2f*)dst + i);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = (float)src[n - 1];
}
int launch_widen(Context* c, const float* src, double* dst, size_t n) {
    if (n == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_widen, dim3(stream_grid(c, (n + 1) / 2)), dim3(kStream), 0, c->stream, src, dst, n);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}
int launch_narrow(Context* c, const double* src, float* dst, size_t n) {
    if (n == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_narrow, dim3(stream_grid(c, (n + 1) / 2)), dim3(kStream), 0, c->stream, src, dst, n);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// ---- broadcast binary: dim 0 along threads (coalesced when stride 1, uniform when 0), the outer
// dims decoded once per block with scalar arithmetic (no per-element div/mod, unlike the
// reference's per-thread 128-iteration coordinate loops, fusion.rs:1606-1616). ----------------------
struct BcastParams {
    unsigned long long d0, nchunks;
    int rank;
    unsigned long long shape[8], sa[8], sb[8];
};

template <class T, class F>
__global__ void __launch_bounds__(kBlock) k_bcast2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                   BcastParams p, F f) {
    const unsigned long long blk = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;
    const unsigned long long chunk = blk % p.nchunks;
    const unsigned long long outer = blk / p.nchunks;
    unsigned long long offa = 0, offb = 0, rem = outer;
    for (int d = 1; d < p.rank; ++d) {
        const unsigned long long cdim = rem % p.shape[d];
        rem /= p.shape[d];
        offa += cdim * p.sa[d];
        offb += cdim * p.sb[d];
    }
    if (rem != 0) return;
    const unsigned long long obase = outer * p.d0;
    constexpr int E = BcastE<T>::value;
    const unsigned long long i0 = chunk * (unsigned long long)(kBlock * E) + threadIdx.x;
    // the loads of the E elements are issued together, then ONE rolled loop applies the op (a single copy of
    // the body: pow / atan2 / hypot are large)
    double x[E], y[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long i = i0 + (unsigned long long)e * kBlock;
        const bool ok = i < p.d0;
        x[e] = ok ? (double)a[offa + i * p.sa[0]] : 0.0;
        y[e] = ok ? (double)b[offb + i * p.sb[0]] : 1.0;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long i = i0 + (unsigned long long)e * kBlock;
        if (i < p.d0) __builtin_nontemporal_store((T)f(x[e], y[e]), out + obase + i);  // streamed once: 134 -> us for the 512 MiB of A(8192x1) .* B(1x8192)
    }
}

// Short dim 0 (a 32 x N matrix with a 1 x N or 32 x 1 operand): the kernel above gives every outer index a block of its own - 32 active
// lanes per block, half a million blocks: 520 us where the bytes take 40.  Here the threads run over the FLAT output (coalesced
// stores; a full-size operand is read coalesced too) and each decodes its own coordinates with 32-bit arithmetic - a division per
// element and one per outer dimension, which a streaming kernel can afford when the alternative is empty lanes.
template <class T, class F>
__global__ void __launch_bounds__(kBlock) k_bcast2_flat(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, BcastParams p,
                                                        unsigned n, F f) {
    const unsigned d0 = (unsigned)p.d0, stride = gridDim.x * kBlock;
    for (unsigned idx = blockIdx.x * kBlock + threadIdx.x; idx < n; idx += stride) {
        unsigned rem = idx / d0;
        const unsigned i = idx - rem * d0;
        unsigned long long offa = (unsigned long long)i * p.sa[0], offb = (unsigned long long)i * p.sb[0];
        for (int d = 1; d < p.rank; ++d) {
            const unsigned sd = (unsigned)p.shape[d];
            const unsigned q = rem / sd, cdim = rem - q * sd;
            rem = q;
            offa += (unsigned long long)cdim * p.sa[d];
            offb += (unsigned long long)cdim * p.sb[d];
        }
        out[idx] = (T)f((double)a[offa], (double)b[offb]);
    }
}

template <int OP, class T>
static int binary_bcast_dispatch(Context* c, int op, const T* a, const T* b, T* out, size_t n, const BroadcastDesc& d) {
    if (op == OP) {
        if (n == 0) return RMHIP_OK;
        BcastParams p;
        p.rank = d.rank;
        p.d0 = d.out_shape[0];
        p.nchunks = (p.d0 + (unsigned long long)kBlock * BcastE<T>::value - 1) / ((unsigned long long)kBlock * BcastE<T>::value);
        unsigned long long outer = 1;
        for (int i = 0; i < 8; ++i) {
            p.shape[i] = i < d.rank ? d.out_shape[i] : 1;
            p.sa[i] = i < d.rank ? d.stride_a[i] : 0;
            p.sb[i] = i < d.rank ? d.stride_b[i] : 0;
            if (i >= 1 && i < d.rank) outer *= d.out_shape[i];
        }
        if (p.d0 < 128 && outer >= 64 && n < 0x80000000ULL) {  // short dim 0, many outer indices: flat threads (32-bit index + stride cannot wrap below 2^31)
            const unsigned long long want = (n + kBlock - 1) / kBlock, cap = (unsigned long long)c->num_cus * 16;
            hipLaunchKernelGGL((k_bcast2_flat<T, BinaryF<OP>>), dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, c->stream, a, b, out, p,
                               (unsigned)n, BinaryF<OP>());
            c->tel.kernel_launches++;
            RMHIP_HIP_CHECK(hipGetLastError());
            return RMHIP_OK;
        }
        const unsigned long long blocks = p.nchunks * outer;
        const unsigned long long gx = blocks < 1048576ULL ? blocks : 1048576ULL;
        const unsigned long long gy = (blocks + gx - 1) / gx;
        if (gy > 65535ULL) return fail(RMHIP_ERR_UNSUPPORTED, "broadcast grid too large");
        hipLaunchKernelGGL((k_bcast2<T, BinaryF<OP>>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, c->stream, a, b,
                           out, p, BinaryF<OP>());
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    if constexpr (OP + 1 < RMHIP_BINARY_OP_COUNT) return binary_bcast_dispatch<OP + 1>(c, op, a, b, out, n, d);
    return fail(RMHIP_ERR_UNSUPPORTED, "binary op %d not supported by provider", op);
}
int launch_binary_bcast(Context* c, int op, const double* a, const double* b, double* out, size_t n,
                        const BroadcastDesc& d) {
    return binary_bcast_dispatch<0>(c, op, a, b, out, n, d);
}
int launch_binary_bcast_f32(Context* c, int op, const float* a, const float* b, float* out, size_t n,
                            const BroadcastDesc& d) {
    return binary_bcast_dispatch<0>(c, op, a, b, out, n, d);
}

// ---- fills ---------------------------------------------------------------------------------------
// Write-only kernels reach the HBM write ceiling with SMALL blocks that each own one contiguous chunk and a grid that covers the
// buffer once - 1024 doubles per 256-thread block: 6.1 TB/s at 512 MiB against 4.6-4.8 for capped grid-stride loops, 16-byte or
// non-temporal stores alike (scripts/micro/write_patterns.hip, profiles/r04_write_patterns.txt).
__global__ void __launch_bounds__(256) k_fill(double* __restrict__ out, size_t n, double value) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (b + e * 256 < n) out[b + e * 256] = value;
}

// counter-based splitmix64 (identical to oracle.c orc_fill_uniform)
__global__ void __launch_bounds__(kStream) k_fill_uniform(double* __restrict__ out, size_t n, unsigned long long seed,
                                                         double lo, double hi) {
    const size_t stride = (size_t)gridDim.x * kStream;
    for (size_t i = (size_t)blockIdx.x * kStream + threadIdx.x; i < n; i += stride) {
        unsigned long long z = seed + (unsigned long long)(i + 1) * 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        z = z ^ (z >> 31);
        out[i] = lo + (hi - lo) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
    }
}

int launch_fill(Context* c, double* dst, size_t n, double value) {
    if (n == 0) return RMHIP_OK;
    const size_t blocks = (n + 1023) / 1024;
    if (blocks > 0x7fffffffULL) return fail(RMHIP_ERR_UNSUPPORTED, "fill: %zu elements exceed the launch limits", n);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, c->stream, dst, n, value);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int launch_fill_uniform(Context* c, double* dst, size_t n, uint64_t seed, double lo, double hi) {
    if (n == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_fill_uniform, dim3(stream_grid(c, n)), dim3(kStream), 0, c->stream, dst, n,
                       (unsigned long long)seed, lo, hi);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

}  // namespace rmhip
