// svdsolve.hip -- the reference's own definition of `A\b`, on the device, for what the LU paths must refuse.
//
// The CPU builtin answers EVERY shape with the minimum-norm least-squares solution of an SVD,
//   x = V diag(1/s_i, s_i > tol) U' b,   tol = eps * max(m, n) * max(s_max, 1)
// (crates/runmat-runtime/src/builtins/math/linalg/ops/mldivide.rs:380-404; nalgebra `SVD::solve`).  The LU solve and the
// Gram-matrix least squares (rmhip_ops.cpp) reproduce that answer only for full-rank, reasonably conditioned systems and hand
// everything else back to the caller (RMHIP_ERR_SINGULAR / UNSUPPORTED).  For systems whose smaller dimension is at most
// svd_max_cols() (4096, RMHIP_SVD_MAX_COLS) this file computes the same thing the reference computes: a one-sided Jacobi SVD of the tall orientation W (p x q,
// p >= q) - rotations of column pairs until every pair is orthogonal to 1e-15, singular values = column norms, right vectors
// accumulated in V - and the pseudo-inverse applied with the reference's tolerance rule.  One-sided Jacobi is what oracle.c
// restates nalgebra's SVD with, it is accurate for small singular values (relative, not absolute, accuracy), and it is a chain of
// embarrassingly parallel steps: a sweep is q - 1 steps of q / 2 independent column pairs (round-robin tournament), one workgroup per
// pair - two dot products and a norm by a block reduction, then the rotation of the two columns of W and of V.  Launch-bound
// (sweeps x (q - 1) launches of a few us) up to q ~ 1024 - 0.15 s at 512, 0.45 s at 1024 -, bound by the traffic of the column pairs
// beyond (1.7 s at 2048, 9.6 s at 4096: every step streams W and V once and a half); that is the price of exact rank semantics,
// against a host round trip of the whole matrix plus a CPU SVD of the same order (minutes at 4096).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace rmhip {

static constexpr int JAC_THREADS = 256;

int svd_max_cols() {
    static const int v = [] {
        const char* e = std::getenv("RMHIP_SVD_MAX_COLS");
        const long x = e ? std::strtol(e, nullptr, 10) : 0;
        return x > 0 && x <= 32768 ? (int)x : kSvdMaxColsDefault;
    }();
    return v;
}

__device__ __forceinline__ double jac_block_sum(double v, double* lds) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // lds may still be read from the previous call
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    return ((lds[0] + lds[1]) + lds[2]) + lds[3];  // every thread, fixed order
}

// pair `b` of step `t` in the round-robin tournament over Q (even) players: player Q-1 stays, the others rotate
__device__ __forceinline__ void jac_pair(int Q, int t, int b, int* x, int* y) {
    const int m = Q - 1;
    if (b == 0) {
        *x = m;
        *y = t % m;
    } else {
        *x = (t + b) % m;
        *y = (t - b + m) % m;
    }
}

__global__ void __launch_bounds__(JAC_THREADS) k_jacobi_step(double* __restrict__ W, size_t p, int q, int Q, double* __restrict__ V, int step,
                                                             unsigned long long* __restrict__ off_bits) {
    __shared__ double lds[4];
    __shared__ double s_cs[2];
    int a, b;
    jac_pair(Q, step, blockIdx.x, &a, &b);
    if (a > b) {
        const int tmp = a;
        a = b;
        b = tmp;
    }
    if (b >= q) return;  // the padding player of an odd q
    double* wa = W + (size_t)a * p;
    double* wb = W + (size_t)b * p;
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
    for (size_t i = threadIdx.x; i < p; i += JAC_THREADS) {
        const double x = wa[i], y = wb[i];
        alpha += x * x;
        beta += y * y;
        gamma += x * y;
    }
    alpha = jac_block_sum(alpha, lds);
    beta = jac_block_sum(beta, lds);
    gamma = jac_block_sum(gamma, lds);
    if (gamma == 0.0) return;  // uniform
    const double lim = fabs(gamma) / sqrt(alpha * beta);
    if (threadIdx.x == 0) {
        if (lim == lim) atomicMax(off_bits, (unsigned long long)__double_as_longlong(lim));
        else atomicMax(off_bits, 0x7ff8000000000000ull);  // NaN / Inf data: the host gives up
    }
    if (!(lim >= 1e-15)) return;  // oracle.c: pairs orthogonal to 1e-15 are left alone
    if (threadIdx.x == 0) {
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t);
        s_cs[0] = c;
        s_cs[1] = c * t;
    }
    __syncthreads();
    const double c = s_cs[0], s = s_cs[1];
    for (size_t i = threadIdx.x; i < p; i += JAC_THREADS) {
        const double x = wa[i], y = wb[i];
        wa[i] = c * x - s * y;
        wb[i] = s * x + c * y;
    }
    double* va = V + (size_t)a * q;
    double* vb = V + (size_t)b * q;
    for (int i = threadIdx.x; i < q; i += JAC_THREADS) {
        const double x = va[i], y = vb[i];
        va[i] = c * x - s * y;
        vb[i] = s * x + c * y;
    }
}

__global__ void __launch_bounds__(JAC_THREADS) k_jacobi_identity(double* __restrict__ V, int q) {
    const size_t i = (size_t)blockIdx.x * JAC_THREADS + threadIdx.x;
    if (i < (size_t)q * q) V[i] = (i / q == i % q) ? 1.0 : 0.0;
}

// sig[j] = ||W_j||, *smax_bits = max_j sig[j]
__global__ void __launch_bounds__(JAC_THREADS) k_jacobi_sigma(const double* __restrict__ W, size_t p, double* __restrict__ sig,
                                                              unsigned long long* __restrict__ smax_bits) {
    __shared__ double lds[4];
    const double* w = W + (size_t)blockIdx.x * p;
    double s2 = 0.0;
    for (size_t i = threadIdx.x; i < p; i += JAC_THREADS) s2 += w[i] * w[i];
    s2 = jac_block_sum(s2, lds);
    if (threadIdx.x == 0) {
        const double s = sqrt(s2);
        sig[blockIdx.x] = s;
        atomicMax(smax_bits, (unsigned long long)__double_as_longlong(s == s ? s : __builtin_inf()));
    }
}

// coef[j, r] *= (sig[j] > tol) ? 1 / sig[j]^2 : 0   (u_j = W_j / sig_j, and one more 1 / sig_j from the pseudo-inverse)
__global__ void __launch_bounds__(JAC_THREADS) k_jacobi_scale(double* __restrict__ coef, int q, size_t nrhs, const double* __restrict__ sig, double tol,
                                                              int* __restrict__ rank_out) {
    const size_t i = (size_t)blockIdx.x * JAC_THREADS + threadIdx.x;
    if (i >= (size_t)q * nrhs) return;
    const int j = (int)(i % q);
    const double s = sig[j];
    const bool keep = s > tol;
    coef[i] = keep ? (coef[i] / s) / s : 0.0;
    if (i < (size_t)q && keep) atomicAdd(rank_out, 1);
}

// X (n x nrhs) = pinv(A) B for A m x n (lda = m), B m x nrhs (ldb = m); *rank_out = numerical rank by the reference's tolerance.
// the decomposition step: W (p x q, tall) with orthogonal columns, V, the singular values and their maximum (on the host)
struct JacobiSvd {
    std::shared_ptr<Allocation> w_mem, v_mem, aux_mem;
    double *W = nullptr, *V = nullptr, *sig = nullptr;
    unsigned long long* ctl = nullptr;  // [0] off, [1] smax, [2] rank (int)
    size_t p = 0, q = 0;
    bool transposed = false;
    double smax = 0.0;
};

static int jacobi_svd(Context* c, const char* who, const double* A, size_t m, size_t n, JacobiSvd* s) {
    s->transposed = m < n;
    const size_t p = s->transposed ? n : m, q = s->transposed ? m : n;  // W is p x q, tall
    s->p = p, s->q = q;
    if (q > (size_t)svd_max_cols()) return fail(RMHIP_ERR_UNSUPPORTED, "%s: the SVD path handles min(rows, cols) <= %d, got %zu", who, svd_max_cols(), q);
    RMHIP_TRY(c->alloc_device(p * q, &s->w_mem));
    RMHIP_TRY(c->alloc_device(q * q, &s->v_mem));
    RMHIP_TRY(c->alloc_device(q + 8, &s->aux_mem));
    double* W = s->W = s->w_mem->ptr;
    double* V = s->V = s->v_mem->ptr;
    s->sig = s->aux_mem->ptr;
    unsigned long long* ctl = s->ctl = reinterpret_cast<unsigned long long*>(s->aux_mem->ptr + q);
    if (s->transposed) RMHIP_TRY(transpose_device(c, A, m, m, n, W, n));
    else RMHIP_HIP_CHECK(hipMemcpyAsync(W, A, p * q * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_jacobi_identity, dim3((unsigned)((q * q + JAC_THREADS - 1) / JAC_THREADS)), dim3(JAC_THREADS), 0, c->stream, V, (int)q);
    const int Q = (int)((q + 1) & ~(size_t)1);
    bool converged = q == 1;
    for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        RMHIP_HIP_CHECK(hipMemsetAsync(ctl, 0, sizeof(unsigned long long), c->stream));
        for (int step = 0; step < Q - 1; ++step)
            hipLaunchKernelGGL(k_jacobi_step, dim3((unsigned)(Q / 2)), dim3(JAC_THREADS), 0, c->stream, W, p, (int)q, Q, V, step, ctl);
        c->tel.kernel_launches += (uint64_t)(Q - 1);
        unsigned long long bits = 0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&bits, ctl, sizeof bits, hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        double off;
        std::memcpy(&off, &bits, sizeof off);
        if (!(off == off) || off > 1.0e300) return fail(RMHIP_ERR_UNSUPPORTED, "%s: non-finite data in the SVD path", who);
        converged = off < 1e-15;
    }
    // (like the oracle, 60 sweeps that did not converge still give the best decomposition found; in practice 6-10 suffice)
    RMHIP_HIP_CHECK(hipMemsetAsync(ctl + 1, 0, 2 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_jacobi_sigma, dim3((unsigned)q), dim3(JAC_THREADS), 0, c->stream, (const double*)W, p, s->sig, ctl + 1);
    unsigned long long sbits = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&sbits, ctl + 1, sizeof sbits, hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::memcpy(&s->smax, &sbits, sizeof s->smax);
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}

// X (n x nrhs) = V diag(1 / s_i, s_i > tol) U' B from the decomposition of the m x n matrix; *rank_out = values kept
static int svd_apply(Context* c, const JacobiSvd& s, size_t m, size_t n, const double* B, size_t nrhs, double tol, double* X, int* rank_out) {
    const size_t p = s.p, q = s.q;
    std::shared_ptr<Allocation> coef_mem;
    RMHIP_TRY(c->alloc_device(q * nrhs, &coef_mem));
    double* coef = coef_mem->ptr;
    int* rank_dev = reinterpret_cast<int*>(s.ctl + 2);
    if (!s.transposed) {
        // A = U S V' with U S = W:  X = V S^-1 U' B = V diag(1/s^2) W' B
        RMHIP_TRY(launch_dgemm_trans(c, true, false, q, nrhs, p, 1.0, s.W, p, B, m, 0.0, coef, q));
        hipLaunchKernelGGL(k_jacobi_scale, dim3((unsigned)((q * nrhs + JAC_THREADS - 1) / JAC_THREADS)), dim3(JAC_THREADS), 0, c->stream, coef, (int)q, nrhs,
                           (const double*)s.sig, tol, rank_dev);
        RMHIP_TRY(launch_dgemm(c, n, nrhs, q, 1.0, s.V, q, coef, q, 0.0, X, n));
    } else {
        // A' = U S V' with U S = W (n x m):  A = V S U',  X = U S^-1 V' B = W diag(1/s^2) V' B
        RMHIP_TRY(launch_dgemm_trans(c, true, false, q, nrhs, q, 1.0, s.V, q, B, m, 0.0, coef, q));
        hipLaunchKernelGGL(k_jacobi_scale, dim3((unsigned)((q * nrhs + JAC_THREADS - 1) / JAC_THREADS)), dim3(JAC_THREADS), 0, c->stream, coef, (int)q, nrhs,
                           (const double*)s.sig, tol, rank_dev);
        RMHIP_TRY(launch_dgemm(c, n, nrhs, q, 1.0, s.W, p, coef, q, 0.0, X, n));
    }
    int h_rank = 0;
    RMHIP_HIP_CHECK(hipMemcpyAsync(&h_rank, rank_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // also: the temporaries go back to the pool on return
    if (rank_out) *rank_out = h_rank;
    c->tel.kernel_launches += 3;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int svd_solve_device(Context* c, const double* A, size_t m, size_t n, const double* B, size_t nrhs, double* X, int* rank_out) {
    if (m == 0 || n == 0 || nrhs == 0) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: empty system");
    JacobiSvd s;
    RMHIP_TRY(jacobi_svd(c, "mldivide", A, m, n, &s));
    const double maxdim = (double)(m > n ? m : n);
    const double tol = 2.220446049250313e-16 * maxdim * (s.smax > 1.0 ? s.smax : 1.0);  // mldivide.rs:396-404
    return svd_apply(c, s, m, n, B, nrhs, tol, X, rank_out);
}

// `eps(x)` as common/linalg.rs:218-228 defines it: the gap to the next double above |x|
static double eps_like(double v) {
    if (v != v) return v;
    if (std::isinf(v)) return INFINITY;
    const double a = std::fabs(v);
    unsigned long long bits;
    std::memcpy(&bits, &a, sizeof bits);
    ++bits;
    double next;
    std::memcpy(&next, &bits, sizeof next);
    return next - a;
}

// the singular values of an m x n matrix on the host (min(m, n) of them, unordered), by the same decomposition
int svd_values_host(Context* c, const char* who, const double* A, size_t m, size_t n, std::vector<double>* values) {
    JacobiSvd s;
    RMHIP_TRY(jacobi_svd(c, who, A, m, n, &s));
    values->resize(s.q);
    RMHIP_HIP_CHECK(hipMemcpyAsync(values->data(), s.sig, s.q * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RMHIP_OK;
}

// `svd_default_tolerance` (common/linalg.rs:209-215): max(m, n) * eps(largest singular value)
double svd_default_tolerance(const std::vector<double>& values, size_t m, size_t n) {
    double mx = 0.0;
    for (double v : values) mx = std::fabs(v) > mx ? std::fabs(v) : mx;
    return (double)(m > n ? m : n) * eps_like(mx);
}

// X (n x m) = pinv(A) with the cutoff `tol` (< 0: the default rule)
int svd_pinv_device(Context* c, const double* A, size_t m, size_t n, double tol, double* X) {
    JacobiSvd s;
    RMHIP_TRY(jacobi_svd(c, "pinv", A, m, n, &s));
    if (tol < 0.0) tol = (double)(m > n ? m : n) * eps_like(s.smax);
    std::shared_ptr<Allocation> eye;
    RMHIP_TRY(c->alloc_device(m * m, &eye));
    hipLaunchKernelGGL(k_jacobi_identity, dim3((unsigned)((m * m + JAC_THREADS - 1) / JAC_THREADS)), dim3(JAC_THREADS), 0, c->stream, eye->ptr, (int)m);
    c->tel.kernel_launches++;
    return svd_apply(c, s, m, n, eye->ptr, m, tol, X, nullptr);
}

}  // namespace rmhip
