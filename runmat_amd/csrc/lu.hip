// lu.hip -- LU factorisation with partial pivoting and the triangular solves behind
// `AccelProvider::lu` / `mldivide` (crates/runmat-accelerate-api/src/lib.rs:2477-2500, 649-698).
//
// Reference behaviour restated here:
//   * pivot rule / singular cut-off: crates/runmat-accelerate/src/host_lu.rs:37-59 -- pivot = FIRST
//     row with strictly larger |a|; |pivot| <= 1e-12 zeroes the sub-column and skips the update;
//     multipliers are a DIVISION by the pivot (factor = a / pivot), products are not fused.
//   * the reference has no GPU LU at all: its wgpu provider downloads, runs the host code and
//     re-uploads (backend/wgpu/provider/ops/solve.rs:144-168).
//
// Algorithm: recursive (Toledo) right-looking LU.  getrf(j0, w): factor the left half, swap +
// triangular-solve + MFMA dgemm-update the right half, factor the right half, swap the left half.
// Every flop outside the <=16-column base panels runs in launch_dgemm (dgemm.hip) with K equal to
// the half-width, so the top levels (where the flops are) see K in the thousands.
// Triangular solves recurse the same way down to a 32x32 substitution kernel.
#include <vector>

#include "common.h"

namespace rmhip {

static constexpr double LU_EPS = 1.0e-12;  // host_lu.rs:3
static constexpr int BASE_W = 16;          // base panel width (columns factored one by one)
static constexpr int TRSM_W = 32;          // base triangular solve size

struct LuState {
    Context* c;
    double* A;
    size_t rows, cols, lda;
    int* ipiv;   // device: ipiv[k] = row swapped with k at step k (LAPACK style, 0-based)
    int* info;   // device: number of pivots with |p| <= LU_EPS
    double* piv; // device scratch: [0] = pivot value of the current column, [1] = skip flag
};

// ---- base panel: one column at a time -----------------------------------------------------------
// (1) pivot search over A[k..rows, k] by ONE block (the column is contiguous: coalesced), then the
//     row swap restricted to the base panel's columns [c0, c1), host_lu.rs:38-52.
__global__ void __launch_bounds__(1024) k_lu_pivot(double* __restrict__ A, size_t lda, size_t rows, size_t k,
                                                   size_t c0, size_t c1, int* __restrict__ ipiv, int* __restrict__ info,
                                                   double* __restrict__ piv) {
    __shared__ double s_abs[16];
    __shared__ unsigned long long s_idx[16];
    __shared__ unsigned long long s_prow;
    const double* col = A + k * lda;
    double best = 0.0;
    unsigned long long bidx = k;
    for (size_t r = k + threadIdx.x; r < rows; r += blockDim.x) {  // ascending per thread: first max kept
        const double a = fabs(col[r]);
        if (a > best) {
            best = a;
            bidx = r;
        }
    }
    // wave reduce with (abs desc, index asc) order == "first strictly larger" over the whole column
    for (int off = 32; off > 0; off >>= 1) {
        const double oa = __shfl_down(best, off, 64);
        const unsigned long long oi = __shfl_down(bidx, off, 64);
        if (oa > best || (oa == best && oi < bidx)) {
            best = oa;
            bidx = oi;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_abs[wave] = best;
        s_idx[wave] = bidx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w) {
            if (s_abs[w] > best || (s_abs[w] == best && s_idx[w] < bidx)) {
                best = s_abs[w];
                bidx = s_idx[w];
            }
        }
        if (best == 0.0) bidx = k;  // all-zero column: pivot_row stays k (host_lu.rs:38)
        s_prow = bidx;
        ipiv[k] = (int)bidx;
        const bool skip = best <= LU_EPS;
        if (skip) atomicAdd(info, 1);
        piv[1] = skip ? 1.0 : 0.0;
    }
    __syncthreads();
    const size_t p = s_prow;
    if (p != k) {
        for (size_t cc = c0 + threadIdx.x; cc < c1; cc += blockDim.x) {
            const double t = A[k + cc * lda];
            A[k + cc * lda] = A[p + cc * lda];
            A[p + cc * lda] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) piv[0] = A[k + k * lda];
}

// (2) multipliers + rank-1 update of the remaining base-panel columns (host_lu.rs:54-70).
__global__ void __launch_bounds__(256) k_lu_update(double* __restrict__ A, size_t lda, size_t rows, size_t k,
                                                   size_t c1, const double* __restrict__ piv) {
    const size_t r = k + 1 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    if (piv[1] != 0.0) {  // singular column: zero it, no update
        A[r + k * lda] = 0.0;
        return;
    }
    const double pivot = piv[0];
    const double factor = A[r + k * lda] / pivot;
    A[r + k * lda] = factor;
    for (size_t cc = k + 1; cc < c1; ++cc) {
        const double prod = factor * A[k + cc * lda];
        A[r + cc * lda] -= prod;
    }
}

// Apply the row interchanges ipiv[k0..k1) to columns [c0, c1).
__global__ void __launch_bounds__(256) k_laswp(double* __restrict__ A, size_t lda, size_t c0, size_t c1, size_t k0,
                                               size_t k1, const int* __restrict__ ipiv) {
    const size_t cc = c0 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cc >= c1) return;
    double* col = A + cc * lda;
    for (size_t k = k0; k < k1; ++k) {
        const size_t p = (size_t)ipiv[k];
        if (p != k) {
            const double t = col[k];
            col[k] = col[p];
            col[p] = t;
        }
    }
}

// ---- small triangular solves (w <= 32): one thread per right-hand-side column ---------------------
// lower, unit diagonal: B <- L^-1 B ; T is w x w at T[0], ldt.
__global__ void __launch_bounds__(64) k_trsm_lower_unit(const double* __restrict__ T, size_t ldt, int w,
                                                        double* __restrict__ B, size_t ldb, size_t ncols) {
    __shared__ double Ls[TRSM_W * TRSM_W];
    for (int e = threadIdx.x; e < w * w; e += 64) Ls[e] = T[(e % w) + (size_t)(e / w) * ldt];
    __syncthreads();
    const size_t cc = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (cc >= ncols) return;
    double* b = B + cc * ldb;
    double x[TRSM_W];
#pragma unroll
    for (int i = 0; i < TRSM_W; ++i) x[i] = i < w ? b[i] : 0.0;
#pragma unroll
    for (int i = 0; i < TRSM_W; ++i) {
        if (i < w) {
            double s = x[i];
#pragma unroll
            for (int k = 0; k < TRSM_W; ++k)
                if (k < i) s -= Ls[i + k * w] * x[k];
            x[i] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < TRSM_W; ++i)
        if (i < w) b[i] = x[i];
}

// upper, non-unit diagonal: B <- U^-1 B.
__global__ void __launch_bounds__(64) k_trsm_upper(const double* __restrict__ T, size_t ldt, int w,
                                                   double* __restrict__ B, size_t ldb, size_t ncols) {
    __shared__ double Us[TRSM_W * TRSM_W];
    for (int e = threadIdx.x; e < w * w; e += 64) Us[e] = T[(e % w) + (size_t)(e / w) * ldt];
    __syncthreads();
    const size_t cc = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (cc >= ncols) return;
    double* b = B + cc * ldb;
    double x[TRSM_W];
#pragma unroll
    for (int i = 0; i < TRSM_W; ++i) x[i] = i < w ? b[i] : 0.0;
#pragma unroll
    for (int ii = TRSM_W - 1; ii >= 0; --ii) {
        if (ii < w) {
            double s = x[ii];
#pragma unroll
            for (int k = 0; k < TRSM_W; ++k)
                if (k > ii && k < w) s -= Us[ii + k * w] * x[k];
            x[ii] = s / Us[ii + ii * w];
        }
    }
#pragma unroll
    for (int i = 0; i < TRSM_W; ++i)
        if (i < w) b[i] = x[i];
}

static int launch_check(Context* c) {
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// B[w x nc] <- L^-1 B with L = unit-lower part of T[w x w]; recursive halving, dgemm in between.
static int trsm_lower_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= (size_t)TRSM_W) {
        hipLaunchKernelGGL(k_trsm_lower_unit, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, c->stream, T, ldt, (int)w,
                           B, ldb, nc);
        return launch_check(c);
    }
    size_t h = ((w / 2 + 15) / 16) * 16;
    if (h >= w) h = w / 2;
    RMHIP_TRY(trsm_lower_rec(c, T, ldt, h, B, ldb, nc));
    RMHIP_TRY(launch_dgemm(c, w - h, nc, h, -1.0, T + h, ldt, B, ldb, 1.0, B + h, ldb));
    return trsm_lower_rec(c, T + h + h * ldt, ldt, w - h, B + h, ldb, nc);
}

static int trsm_upper_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= (size_t)TRSM_W) {
        hipLaunchKernelGGL(k_trsm_upper, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, c->stream, T, ldt, (int)w, B,
                           ldb, nc);
        return launch_check(c);
    }
    size_t h = ((w / 2 + 15) / 16) * 16;
    if (h >= w) h = w / 2;
    RMHIP_TRY(trsm_upper_rec(c, T + h + h * ldt, ldt, w - h, B + h, ldb, nc));
    RMHIP_TRY(launch_dgemm(c, h, nc, w - h, -1.0, T + h * ldt, ldt, B + h, ldb, 1.0, B, ldb));
    return trsm_upper_rec(c, T, ldt, h, B, ldb, nc);
}

static int laswp(LuState& s, size_t c0, size_t c1, size_t k0, size_t k1) {
    if (c1 <= c0 || k1 <= k0) return RMHIP_OK;
    hipLaunchKernelGGL(k_laswp, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, s.c->stream, s.A, s.lda, c0, c1,
                       k0, k1, s.ipiv);
    return launch_check(s.c);
}

// Factor columns [j0, j0+w) over rows [j0, rows); swaps are applied inside that column range only.
static int getrf_rec(LuState& s, size_t j0, size_t w) {
    if (w == 0 || j0 >= s.rows) return RMHIP_OK;
    if (w <= (size_t)BASE_W) {
        const size_t c1 = j0 + w;
        for (size_t k = j0; k < c1 && k < s.rows; ++k) {
            hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(1024), 0, s.c->stream, s.A, s.lda, s.rows, k, j0, c1, s.ipiv,
                               s.info, s.piv);
            RMHIP_TRY(launch_check(s.c));
            if (k + 1 < s.rows) {
                hipLaunchKernelGGL(k_lu_update, dim3((unsigned)((s.rows - k - 1 + 255) / 256)), dim3(256), 0,
                                   s.c->stream, s.A, s.lda, s.rows, k, c1, s.piv);
                RMHIP_TRY(launch_check(s.c));
            }
        }
        return RMHIP_OK;
    }
    size_t h = ((w / 2 + 15) / 16) * 16;
    if (h >= w) h = w / 2;
    const size_t hk = (j0 + h <= s.rows) ? h : (s.rows - j0);  // pivots produced by the left half
    RMHIP_TRY(getrf_rec(s, j0, h));
    RMHIP_TRY(laswp(s, j0 + h, j0 + w, j0, j0 + hk));
    double* A11 = s.A + j0 + j0 * s.lda;
    double* A12 = s.A + j0 + (j0 + h) * s.lda;
    RMHIP_TRY(trsm_lower_rec(s.c, A11, s.lda, hk, A12, s.lda, w - h));
    if (j0 + h < s.rows) {
        double* A21 = s.A + (j0 + h) + j0 * s.lda;
        double* A22 = s.A + (j0 + h) + (j0 + h) * s.lda;
        RMHIP_TRY(launch_dgemm(s.c, s.rows - j0 - h, w - h, h, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda));
        RMHIP_TRY(getrf_rec(s, j0 + h, w - h));
        const size_t k1 = (j0 + w <= s.rows) ? (j0 + w) : s.rows;
        RMHIP_TRY(laswp(s, j0, j0 + h, j0 + h, k1));
    }
    return RMHIP_OK;
}

// In-place LU of A (rows x cols, lda). perm_dev[rows] receives the row permutation as the
// reference reports it (perm[k] = original row now at position k, host_lu.rs:50,107).
// *info_host = number of pivots that hit the singular cut-off.
int lu_factor_device(Context* c, double* A, size_t rows, size_t cols, size_t lda, int* perm_dev, int* info_host) {
    const size_t kmin = rows < cols ? rows : cols;
    int* ipiv = nullptr;
    RMHIP_HIP_CHECK(hipMalloc((void**)&ipiv, sizeof(int) * (rows + 4) + sizeof(double) * 2));
    int* info = ipiv + rows;
    double* piv = (double*)(((uintptr_t)(ipiv + rows + 2) + 7) & ~(uintptr_t)7);
    hipError_t e = hipMemsetAsync(ipiv, 0, sizeof(int) * (rows + 4) + sizeof(double) * 2, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(ipiv);
        return fail(RMHIP_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    LuState s{c, A, rows, cols, lda, ipiv, info, piv};
    int rc = getrf_rec(s, 0, kmin);
    if (rc == RMHIP_OK && cols > rows) {  // wide: finish U's right block
        rc = laswp(s, rows, cols, 0, rows);
        if (rc == RMHIP_OK) rc = trsm_lower_rec(c, A, lda, rows, A + rows * lda, lda, cols - rows);
    }
    std::vector<int> h_ipiv(rows + 1, 0);
    if (rc == RMHIP_OK) {
        e = hipMemcpyAsync(h_ipiv.data(), ipiv, sizeof(int) * (rows + 1), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "lu: reading pivots: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(ipiv);
    if (rc != RMHIP_OK) return rc;
    if (info_host) *info_host = h_ipiv[rows];
    std::vector<int> perm(rows);
    for (size_t r = 0; r < rows; ++r) perm[r] = (int)r;
    for (size_t k = 0; k < kmin; ++k) {
        const int p = h_ipiv[k];
        if ((size_t)p != k) std::swap(perm[k], perm[p]);
    }
    if (perm_dev && rows) {
        RMHIP_HIP_CHECK(hipMemcpyAsync(perm_dev, perm.data(), sizeof(int) * rows, hipMemcpyHostToDevice, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return RMHIP_OK;
}

__global__ void __launch_bounds__(256) k_gather_rows(const double* __restrict__ B, size_t ldb, const int* __restrict__ perm,
                                                     size_t n, size_t nrhs, double* __restrict__ X, size_t ldx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t j = blockIdx.y;
    if (i < n && j < nrhs) X[i + j * ldx] = B[(size_t)perm[i] + j * ldb];
}

// X = U^-1 L^-1 (P B) for square LU (n x n).
int lu_solve_device(Context* c, const double* LU, size_t n, size_t lda, const int* perm_dev, const double* B,
                    size_t nrhs, size_t ldb, double* X, size_t ldx) {
    if (n == 0 || nrhs == 0) return RMHIP_OK;
    if (nrhs > 65535) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: more than 65535 right-hand sides");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256), (unsigned)nrhs), dim3(256), 0, c->stream, B, ldb,
                       perm_dev, n, nrhs, X, ldx);
    RMHIP_TRY(launch_check(c));
    RMHIP_TRY(trsm_lower_rec(c, LU, lda, n, X, ldx, nrhs));
    return trsm_upper_rec(c, LU, lda, n, X, ldx, nrhs);
}

// Split the packed factors into the five outputs of ProviderLuResult (host_lu.rs:72-118):
// L rows x rows (unit diagonal), U rows x cols, P rows x rows, pivot vector rows x 1 (1-based).
__global__ void __launch_bounds__(256) k_lu_extract(const double* __restrict__ LU, size_t rows, size_t cols,
                                                    const int* __restrict__ perm, double* __restrict__ L,
                                                    double* __restrict__ U, double* __restrict__ P,
                                                    double* __restrict__ piv) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t j = blockIdx.y;
    if (i >= rows) return;
    const size_t limit = rows < cols ? rows : cols;
    if (j < rows) {
        if (L) {
            double v = 0.0;
            if (i == j) v = 1.0;
            else if (i > j && j < limit) v = LU[i + j * rows];
            L[i + j * rows] = v;
        }
        if (P) P[i + j * rows] = ((size_t)perm[i] == j) ? 1.0 : 0.0;
    }
    if (j < cols && U) U[i + j * rows] = (i <= j) ? LU[i + j * rows] : 0.0;
    if (j == 0 && piv) piv[i] = (double)(perm[i] + 1);
}

int lu_extract_device(Context* c, const double* LU, size_t rows, size_t cols, const int* perm_dev, double* L, double* U,
                      double* P, double* piv) {
    if (rows == 0) return RMHIP_OK;
    const size_t ny = rows > cols ? rows : cols;
    if (ny > 65535) {
        // split the y range
        for (size_t y0 = 0; y0 < ny; y0 += 65535) {
            (void)y0;
        }
        return fail(RMHIP_ERR_UNSUPPORTED, "lu: matrices wider than 65535 not supported by the extract kernel");
    }
    hipLaunchKernelGGL(k_lu_extract, dim3((unsigned)((rows + 255) / 256), (unsigned)(ny ? ny : 1)), dim3(256), 0,
                       c->stream, LU, rows, cols, perm_dev, L, U, P, piv);
    return launch_check(c);
}

}  // namespace rmhip
