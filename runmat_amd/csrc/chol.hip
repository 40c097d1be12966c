// chol.hip -- `chol(a, lower)` (crates/runmat-accelerate-api/src/lib.rs:2502-2508 -> ProviderCholResult { factor, info } :658-662).
// The reference's host code is its own Cholesky-Crout (builtins/math/linalg/factor/chol.rs:374-433): column by column, a symmetry
// check of every pair first (|a_ij - a_ji| <= 1e-12 max(|a_ij|, |a_ji|, 1)), info = the first column whose pivot is not positive and
// finite, the rows from there on zeroed.  Here the SUCCESS path runs on the device as a recursive blocked factorisation A = R'R:
//     R11 = chol(A11);  R12 = R11^-T A12 (lower non-unit solve with the transposed block);  A22 -= R12' R12 (MFMA);  R22 = chol(A22)
// so that almost all of the n^3 / 3 flops are deep products, with 64 x 64 leaves factored by one workgroup in LDS.  A matrix that is
// not symmetric to the reference's tolerance or not positive definite is RMHIP_ERR_UNSUPPORTED: the caller's host code then produces
// the reference's `info` and partial factor itself (chol.rs:331-342 falls back on any provider error).
#include <algorithm>
#include <cstring>

#include "common.h"

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c);                                             \
    NarrowScope _ns(c)

namespace rmhip {
namespace {

typedef unsigned long long u64;
constexpr int LEAF = 64;

// ONE WAVE factors a w x w block (w <= 64) in place: lane i keeps row i of the lower triangle - read from column i of the (symmetric)
// upper triangle - in registers, and column k of the factor reaches the other lanes as SCALAR broadcasts (v_readlane with a constant
// lane: no LDS, no barrier): 2016 fused updates of three instructions each, ~13 us against 85 us for the LDS form with three barriers
// per column.  A pivot that is not positive and finite records its global column (atomic minimum) and stops the block.
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)__double2loint(v), lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}
template <int K>
__device__ __forceinline__ bool potf2_steps(double (&a)[LEAF], const int lane, const int w, u64 col0, unsigned long long* first_bad) {
    if constexpr (K < LEAF) {
        if (K >= w) return true;  // uniform
        const double d = lane_bcast(a[K], K);
        if (!(d > 0.0 && isfinite(d))) {
            if (lane == 0) atomicMin(first_bad, col0 + (u64)K);
            return false;
        }
        const double r = sqrt(d), rinv = 1.0 / r;  // (uniform: every lane holds the same pivot)
        a[K] = lane == K ? r : a[K] * rinv;        // l(lane, K) = r(K, lane) for lane > K; lanes above the diagonal hold don't-cares
        const double lk = a[K];
        // a(lane, j) -= l(lane, K) l(j, K) for EVERY j > K: entries with j > lane (above the diagonal) and lanes < K compute don't-cares
        // that nothing reads - no per-element predicate, four instructions per update
#pragma unroll
        for (int j = K + 1; j < LEAF; ++j) a[j] = a[j] - lk * lane_bcast(lk, j);
        return potf2_steps<K + 1>(a, lane, w, col0, first_bad);
    } else {
        return true;
    }
}
__global__ void __launch_bounds__(64) k_potf2(double* __restrict__ A, u64 lda, int w, u64 col0, unsigned long long* __restrict__ first_bad) {
    const int lane = threadIdx.x;
    double a[LEAF];
#pragma unroll
    for (int j = 0; j < LEAF; ++j) a[j] = (lane < w && j <= lane) ? A[j + (u64)lane * lda] : 0.0;  // a(lane, j) = a(j, lane)
    potf2_steps<0>(a, lane, w, col0, first_bad);
    if (lane < w) {
#pragma unroll
        for (int j = 0; j < LEAF; ++j)
            if (j < w) A[j + (u64)lane * lda] = j <= lane ? a[j] : 0.0;  // r(j, lane) = l(lane, j); zeros below the diagonal
    }
}

// hermitian_pair_matches for real data (chol.rs:366-372) over every pair: the smallest column j (> i) of a failing pair
__global__ void __launch_bounds__(256) k_sym_rel(const double* __restrict__ a, u64 n, unsigned long long* __restrict__ first_bad) {
    __shared__ double up[32][33], lo[32][33];
    const u64 p = blockIdx.x;
    u64 tj = (u64)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (tj * (tj + 1) / 2 > p) --tj;
    while ((tj + 1) * (tj + 2) / 2 <= p) ++tj;
    const u64 ti = p - tj * (tj + 1) / 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int q = 0; q < 4; ++q) {
        const int cc = ty + 8 * q;
        const u64 ru = ti * 32 + tx, cu = tj * 32 + cc, rl = tj * 32 + tx, cl = ti * 32 + cc;
        up[cc][tx] = (ru < n && cu < n) ? a[ru + cu * n] : 0.0;
        lo[cc][tx] = (rl < n && cl < n) ? a[rl + cl * n] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int cc = ty + 8 * q;
        const u64 row = ti * 32 + tx, col = tj * 32 + cc;
        if (row >= n || col >= n || row >= col) continue;
        const double x = up[cc][tx], y = lo[tx][cc];
        const double scale = fmax(fmax(fabs(x), fabs(y)), 1.0);
        if (!(fabs(x - y) <= 1.0e-12 * scale)) atomicMin(first_bad, col);
    }
}

struct CholCtx {
    Context* c;
    double* W;
    u64 ld;
    unsigned long long* first_bad;
    double* tbuf;  // transposed copy of a diagonal block (largest: (n / 2)^2)
};

int chol_rec(const CholCtx& k, u64 j0, u64 n) {
    Context* c = k.c;
    double* A = k.W + j0 + j0 * k.ld;
    if (n <= (u64)LEAF) {
        hipLaunchKernelGGL(k_potf2, dim3(1), dim3(64), 0, c->stream, A, k.ld, (int)n, j0, k.first_bad);
        c->tel.kernel_launches++;
        return RMHIP_OK;
    }
    u64 n1 = ((n / 2 + LEAF - 1) / LEAF) * LEAF;
    if (n1 >= n) n1 = n - LEAF < n ? ((n - 1) / LEAF) * LEAF : n / 2;
    const u64 n2 = n - n1;
    RMHIP_TRY(chol_rec(k, j0, n1));
    double* A12 = A + n1 * k.ld;
    double* A22 = A + n1 + n1 * k.ld;
    // R12 = R11^-T A12: the transposed block is lower with a stored diagonal
    RMHIP_TRY(transpose_device(c, A, k.ld, n1, n1, k.tbuf, n1));
    RMHIP_TRY(trsm_lower_nonunit_device(c, k.tbuf, n1, n1, A12, k.ld, n2));
    // A22 -= R12' R12  (R12 is stored k x m = n1 x n2: the transposed-A form).  Only the upper triangle is ever read again: wide
    // updates go column panel by column panel, rows down to the panel's last column only - half of the square's flops.
    const u64 panel = 1024;
    if (n2 <= 2 * panel) {
        RMHIP_TRY(launch_dgemm_trans(c, true, false, n2, n2, n1, -1.0, A12, k.ld, A12, k.ld, 1.0, A22, k.ld));
    } else {
        for (u64 p0 = 0; p0 < n2; p0 += panel) {
            const u64 pw = std::min(panel, n2 - p0), rows = p0 + pw;
            RMHIP_TRY(launch_dgemm_trans(c, true, false, rows, pw, n1, -1.0, A12, k.ld, A12 + p0 * k.ld, k.ld, 1.0, A22 + p0 * k.ld, k.ld));
        }
    }
    return chol_rec(k, j0 + n1, n2);
}

}  // namespace
}  // namespace rmhip

int rmhip_chol(rmhip_ctx* ctx, rmhip_buf a, int lower, rmhip_buf* factor, unsigned* info) {
    CTX_OR_FAIL(ctx);
    if (!factor || !info) return fail(RMHIP_ERR_INVALID, "chol: null output");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t>& s = ab.shape;
    for (size_t d = 2; d < s.size(); ++d)
        if (s[d] != 1) return fail(RMHIP_ERR_UNSUPPORTED, "chol: input must be 2-D");
    const size_t rows = s.empty() ? 1 : s[0], cols = s.size() < 2 ? 1 : s[1];
    if (rows != cols) return fail(RMHIP_ERR_INVALID, "chol: input matrix must be square");  // chol.rs:324-326
    const size_t n = rows;
    const size_t sq[2] = {n, n};
    *info = 0;
    Buffer ob;
    RMHIP_TRY(c->new_buffer(sq, 2, factor, &ob));
    if (n == 0) return RMHIP_OK;
    std::shared_ptr<Allocation> work, tb, flag;
    int rc = c->alloc_device(n * n, &work);
    const size_t half = ((n / 2 + LEAF - 1) / LEAF) * LEAF + LEAF;
    if (!rc) rc = c->alloc_device(std::max<size_t>(half * half, 1), &tb);
    if (!rc) rc = c->alloc_device(2, &flag);
    unsigned long long init[2] = {~0ull, ~0ull};
    if (!rc && hipMemcpyAsync(flag->ptr, init, sizeof init, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(RMHIP_ERR_HIP, "chol: copy failed");
    if (!rc) {
        unsigned long long* fl = (unsigned long long*)flag->ptr;
        const u64 tiles = (n + 31) / 32, pairs = tiles * (tiles + 1) / 2;
        if (pairs > 0x7fffffffull) rc = fail(RMHIP_ERR_UNSUPPORTED, "chol: %zu rows", n);
        if (!rc) {
            hipLaunchKernelGGL(k_sym_rel, dim3((unsigned)pairs), dim3(256), 0, c->stream, ab.data(), (u64)n, fl);
            c->tel.kernel_launches++;
            if (hipMemcpyAsync(work->ptr, ab.data(), n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = fail(RMHIP_ERR_HIP, "chol: copy failed");
        }
        if (!rc) {
            CholCtx k{c, work->ptr, (u64)n, fl + 1, tb->ptr};
            rc = chol_rec(k, 0, n);
        }
        unsigned long long got[2] = {~0ull, ~0ull};
        if (!rc && (hipMemcpyAsync(got, fl, sizeof got, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
            rc = fail(RMHIP_ERR_HIP, "chol: read-back failed");
        if (!rc && (got[0] != ~0ull || got[1] != ~0ull)) {
            const unsigned long long first = got[0] < got[1] ? got[0] : got[1];
            rc = fail(RMHIP_ERR_UNSUPPORTED, "chol: matrix is not %s at column %llu; the host path reports info and the partial factor",
                      got[0] <= got[1] ? "symmetric" : "positive definite", first + 1);
        }
    }
    if (!rc) {
        // the strictly lower triangle of the workspace holds updated garbage (the products above are full squares): keep R only
        rc = lower ? transpose_device(c, work->ptr, n, n, n, ob.data(), n)
                   : (hipMemcpyAsync(ob.data(), work->ptr, n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? RMHIP_OK : fail(RMHIP_ERR_HIP, "chol: copy failed"));
    }
    if (!rc) {
        rmhip_buf tri = 0;
        rc = rmhip_tri(ctx, *factor, lower ? 0 : 1, 0, &tri);  // triu(R) / tril(L): the leaves already zero their own blocks, the off-diagonal blocks below need it
        if (!rc) {
            rmhip_free(ctx, *factor);
            *factor = tri;
        }
    }
    if (rc) {
        rmhip_free(ctx, *factor);
        *factor = 0;
    }
    return rc;
}
