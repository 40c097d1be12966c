// small_solve.hip -- x = A \ B for a small square system in ONE launch.
//
// The blocked solve path (lu.hip + rmhip_ops.cpp) spends a dozen launches and two host read-backs on any system: 0.15 ms at n = 8,
// 0.20 ms at n = 64 (scripts/solve_small.py), of which the elimination itself is a fraction.  A small augmented matrix [A | B] (the
// kernel: n <= 128 and 16 right-hand sides; used up to n = 64) fits the LDS of one CU: one workgroup loads it, eliminates with partial pivoting (largest |a| of the
// column, first occurrence - the rule of host_lu.rs:37-59; pivots never leave the provider on this path, mldivide.rs:380-404 defines the
// answer, not the factorisation), substitutes back one wave per right-hand side with the finished component broadcast by v_readlane
// (no barrier inside the substitution), and leaves the pivot statistics the caller's singular / nearly-singular tests need
// (rmhip_ops.cpp: a pivot <= 1e-12 -> SINGULAR -> the SVD path; a tiny pivot ratio -> the SVD decides) next to the solution: one launch,
// one read-back.  Deterministic: fixed reduction order, no atomics.
#include "common.h"

namespace rmhip {

static constexpr int SS_THREADS = 256;
static constexpr double SS_EPS = 1.0e-12;  // host_lu.rs:3

// exact maximum of the wave's keys on the DPP network (the pattern of lu.hip's wave_max_u64), result in every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long ss_dpp(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
__device__ __forceinline__ unsigned long long ss_max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long ss_wave_max(unsigned long long v) {
    v = ss_max(v, ss_dpp<0x111, 0xf>(v));
    v = ss_max(v, ss_dpp<0x112, 0xf>(v));
    v = ss_max(v, ss_dpp<0x114, 0xf>(v));
    v = ss_max(v, ss_dpp<0x118, 0xf>(v));
    v = ss_max(v, ss_dpp<0x142, 0xa>(v));
    v = ss_max(v, ss_dpp<0x143, 0xc>(v));
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// maximum / minimum over each row of 16 lanes on the DPP network, result read from lane 15 (every row of the wave holds the same 16
// candidates, so lane 15 of row 0 speaks for all)
__device__ __forceinline__ unsigned long long ss_row16_max(unsigned long long v) {
    v = ss_max(v, ss_dpp<0x111, 0xf>(v));
    v = ss_max(v, ss_dpp<0x112, 0xf>(v));
    v = ss_max(v, ss_dpp<0x114, 0xf>(v));
    v = ss_max(v, ss_dpp<0x118, 0xf>(v));
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 15);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 15);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int ss_row16_min(int v) {
    const int big = 0x7fffffff;
    int o = __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false);
    v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false);
    v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false);
    v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false);
    v = o < v ? o : v;
    return __builtin_amdgcn_readlane(v, 15);
}

// W (LDS, column-major, leading dimension ld = n | 1): columns 0 .. n-1 = A, n .. n+nrhs-1 = B.
// stats[0] = min |pivot|, stats[1] = max |pivot|, stats[2] = number of pivots <= 1e-12 (or NaN)
//
// Two barriers per column.  A 16 x 16 thread tile walks the trailing block (row offsets along the lanes' low bits: consecutive LDS
// words); the sixteen threads that update column k + 1 leave their best |entry| of it behind as sixteen candidates, so the next
// column's pivot search is a read of those and a reduction over 16 lanes that every wave does for itself - no exchange between waves,
// no barrier of its own.  (First version: search -> barrier -> swap -> barrier -> multipliers -> barrier -> update -> barrier, every
// phase one dependent LDS round trip: 1.5 us per column at any order.)
__global__ void __launch_bounds__(SS_THREADS) k_small_solve(const double* __restrict__ A, const double* __restrict__ B, int n, int nrhs,
                                                            double* __restrict__ X, double* __restrict__ stats) {
    extern __shared__ double W[];
    __shared__ unsigned long long c_key[2][16];
    __shared__ int c_idx[2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ld = n | 1, nc = n + nrhs;
    for (int e = t; e < n * n; e += SS_THREADS) W[(e % n) + (e / n) * ld] = A[e];
    for (int e = t; e < n * nrhs; e += SS_THREADS) W[(e % n) + (n + e / n) * ld] = B[e];
    __syncthreads();
    const int ti = t & 15, tj = t >> 4;
    if (tj == 0) {  // candidates for column 0: thread ti looks at rows ti, ti + 16, ...
        unsigned long long bk = 0;
        int bi = 0x7fffffff;
        for (int r = ti; r < n; r += 16) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(__builtin_fabs(W[r]));
            if (bi == 0x7fffffff || key > bk) {  // strictly larger: the first occurrence wins
                bk = key;
                bi = r;
            }
        }
        c_key[0][ti] = bk;
        c_idx[0][ti] = bi;
    }
    __syncthreads();
    double pmin = __builtin_inf(), pmax = 0.0;
    int bad = 0;
    for (int k = 0; k < n; ++k) {
        const int par = k & 1;
        // ---- pivot of column k from the sixteen candidates: largest |a| (as an integer: order preserving for non-negative doubles, a
        // NaN sorts above +Inf), lowest row among equals ----
        const unsigned long long ck = c_key[par][lane & 15];
        const int ci = c_idx[par][lane & 15];
        const unsigned long long best = ss_row16_max(ck);
        int p = ss_row16_min(ck == best ? ci : 0x7fffffff);
        if (p >= n) p = k;
        const double piv_abs = __longlong_as_double((long long)best);
        const bool skip = !(piv_abs > SS_EPS);  // |pivot| <= 1e-12 or NaN: no elimination with this column (host_lu.rs:54-59)
        if (skip) bad += 1;
        if (piv_abs == piv_abs) {
            pmin = piv_abs < pmin ? piv_abs : pmin;
            pmax = piv_abs > pmax ? piv_abs : pmax;
        }
        // ---- rows k and p change places in the columns that still matter (k .. nc-1) ----
        if (p != k) {
            for (int j = k + t; j < nc; j += SS_THREADS) {
                const double a = W[k + j * ld], b = W[p + j * ld];
                W[k + j * ld] = b;
                W[p + j * ld] = a;
            }
        }
        __syncthreads();
        // ---- trailing update, rows k+1 .. n-1, columns k+1 .. nc-1.  A thread's (at most eight) multipliers stay in registers for all
        // of its columns, and the eight elements of a column are loaded together before any is updated.  One reciprocal per thread
        // instead of eight divisions (the factors are not kept; the multipliers differ from the quotients by an ulp). ----
        double m[8];
        if (!skip) {
            const double inv = 1.0 / W[k + k * ld];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int r = k + 1 + ti + 16 * a;
                m[a] = r < n ? W[r + k * ld] * inv : 0.0;
            }
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a) m[a] = 0.0;
        }
        for (int j = k + 1 + tj; j < nc; j += 16) {
            const double ukj = W[k + j * ld];
            double v[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int r = k + 1 + ti + 16 * a;
                v[a] = r < n ? W[r + j * ld] : 0.0;
            }
            if (!skip) {
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int r = k + 1 + ti + 16 * a;
                    v[a] = v[a] - m[a] * ukj;
                    if (r < n) W[r + j * ld] = v[a];
                }
            }
            if (j == k + 1 && j < n) {  // the next pivot column: this thread's best candidate of it
                unsigned long long bk = 0;
                int bi = 0x7fffffff;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int r = k + 1 + ti + 16 * a;
                    const unsigned long long key = (unsigned long long)__double_as_longlong(__builtin_fabs(v[a]));
                    if (r < n && (bi == 0x7fffffff || key > bk)) {
                        bk = key;
                        bi = r;
                    }
                }
                c_key[par ^ 1][ti] = bk;
                c_idx[par ^ 1][ti] = bi;
            }
        }
        __syncthreads();
    }
    // ---- back substitution: one wave per right-hand side, lane l holds rows l and l + 64; the component finished at step k is
    // broadcast with v_readlane, so a wave never waits for another ----
    for (int r = wave; r < nrhs; r += SS_THREADS / 64) {
        const int col = n + r;
        double b0 = lane < n ? W[lane + col * ld] : 0.0, b1 = lane + 64 < n ? W[lane + 64 + col * ld] : 0.0;
        for (int k = n - 1; k >= 0; --k) {
            const double src = k < 64 ? b0 : b1;
            const int sl = k & 63;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)__double_as_longlong(src), sl);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)__double_as_longlong(src) >> 32), sl);
            const double xk = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) / W[k + k * ld];
            if (lane == sl) {
                if (k < 64) b0 = xk;
                else b1 = xk;
            }
            if (lane < k) b0 -= W[lane + k * ld] * xk;
            if (lane + 64 < k) b1 -= W[lane + 64 + k * ld] * xk;
        }
        if (lane < n) X[lane + (size_t)r * n] = b0;
        if (lane + 64 < n) X[lane + 64 + (size_t)r * n] = b1;
    }
    if (t == 0) {
        stats[0] = pmin;
        stats[1] = pmax;
        stats[2] = (double)bad;
    }
}

// The kernel takes n <= 128; the policy stops at 64, where it still wins (us per solve, one-launch / blocked, scripts/solve_small.py:
// n = 8: 37 / 152, 32: 76 / 172, 64: 145 / 197, 100: 250 / 275, 128: 351 / 299 - one workgroup on an otherwise idle device runs at
// idle clocks, and a column is a chain of LDS round trips that nothing overlaps).
bool small_solve_applies(size_t n, size_t nrhs) { return n >= 2 && n <= 64 && nrhs >= 1 && nrhs <= 16; }

// X (n x nrhs, ld n) = A \ B for A n x n (ld n), B n x nrhs (ld n).  *min_abs / *max_abs = extreme |pivot|, *bad = pivots <= 1e-12 or NaN:
// when *bad != 0 the contents of X mean nothing and the caller takes its singular path.  Synchronises the stream (the read-back).
int small_solve_device(Context* c, const double* A, const double* B, size_t n, size_t nrhs, double* X, double* min_abs, double* max_abs, size_t* bad) {
    std::shared_ptr<Allocation> st;
    RMHIP_TRY(c->alloc_device(4, &st));
    const size_t ld = n | 1, lds_bytes = ld * (n + nrhs) * sizeof(double);
    c->ensure_max_lds((const void*)k_small_solve, 152 * 1024);  // 129 x 144 doubles at the limits (+ the static exchange words: below the 160 KiB of a CU)
    hipLaunchKernelGGL(k_small_solve, dim3(1), dim3(SS_THREADS), lds_bytes, c->stream, A, B, (int)n, (int)nrhs, X, st->ptr);
    RMHIP_HIP_CHECK(hipGetLastError());
    double h[3] = {0, 0, 0};
    RMHIP_HIP_CHECK(hipMemcpyAsync(h, st->ptr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->tel.kernel_launches++;
    *min_abs = h[0];
    *max_abs = h[1];
    *bad = (size_t)h[2];
    return RMHIP_OK;
}

}  // namespace rmhip
