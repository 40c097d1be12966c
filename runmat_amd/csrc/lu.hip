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
// Every flop outside the <=64-column base panels runs in launch_dgemm (dgemm.hip) with K equal to
// the half-width, so the top levels (where the flops are) see K in the thousands.
// Triangular solves recurse the same way down to a 32x32 substitution kernel.
#include <unordered_map>
#include <vector>

#include "common.h"

namespace rmhip {

static constexpr double LU_EPS = 1.0e-12;  // host_lu.rs:3
static constexpr int BASE_W = 64;          // base panel width (columns factored one launch each)
static constexpr int TRSM_W = 128;         // base triangular solve size (one fused launch)

struct LuState {
    Context* c;
    double* A;
    size_t rows, cols, lda;
    int* ipiv;        // device: ipiv[k] = position swapped with k at step k (LAPACK style, 0-based)
    int* info;        // device: number of pivots with |p| <= LU_EPS
    int* pos_of;      // device [rows]: current position of physical row r inside the base panel, -1 once retired
    int* row_at;      // device [rows]: physical row currently at position p (only entries >= k are meaningful)
    int* prow;        // device [rows]: physical pivot row chosen at step k (before the panel's physical interchange)
    int2* plist;      // device [max_panels][PLIST]: (dst, src) row moves of each base panel, applied as one parallel gather
    std::vector<size_t>* panel_start;  // host: first column of every base panel factored so far (ascending)
    double* cand_abs; // device: [2][MAX_PANEL_BLOCKS] per-block arg-max candidates (double buffered by column parity)
    int* cand_pos;    //         position of the candidate row
    int* cand_row;    //         physical row of the candidate
};

static constexpr int MAX_PANEL_BLOCKS = 1024;  // 64 rows per block => up to 65536 rows per panel
static constexpr int PANEL_JT = 8;            // panel columns per thread (unrolled batch; keep the code small)
static constexpr int PANEL_GROUPS = BASE_W / PANEL_JT;
static constexpr int PANEL_ROWS = 64;          // rows per block (32 was measured slower: 278 vs 246 ms at n = 16384)
static constexpr int PANEL_THREADS = PANEL_ROWS * PANEL_GROUPS;

// ---- base panel: ONE launch per column, lazy pivoting ----------------------------------------------
// Inside a base panel rows are NOT moved: a pivot row simply retires where it lies, and the position
// permutation the reference's tie-break needs ("first row in CURRENT order with strictly larger
// |a|", host_lu.rs:38-47) is tracked in two small maps (pos_of / row_at).  The physical interchange
// of the panel columns happens once per panel (k_laswp over ipiv), which yields exactly the layout
// sequential swapping would have produced.
//
// Launch k (j0 <= k < c1) of a panel [j0, c1):
//   1. reduce the candidates the previous launch left for column k -> pivot (prow, ppos)
//   2. eliminate column k from every still-active row (multiplier by division, unfused
//      multiply-subtract over the remaining panel columns, host_lu.rs:61-70)
//   3. while doing so collect the arg-max candidates of column k+1 for the next launch
// Launch "first" (k == j0 - 1) resets the maps and only performs step 3 for column j0.
//
// CODE SIZE IS THE PERFORMANCE KNOB: the kernel runs once per column on cold instruction caches, and
// measured launch-to-launch time grows ~1 us per KB of straight-line code (a 64-column unrolled body
// took ~20 us even with its loads and stores removed).  Hence rolled loops, no per-row record scans,
// and only a PANEL_JT-deep unrolled load/update/store batch.
__global__ void __launch_bounds__(PANEL_THREADS) k_lu_col(double* __restrict__ A, size_t lda, size_t rows, int j0, int k,
                                                          int c1, int first, int nblocks, int* __restrict__ pos_of,
                                                          int* __restrict__ row_at, int* __restrict__ prow_arr,
                                                          int* __restrict__ ipiv, int* __restrict__ info,
                                                          double* __restrict__ cand_abs, int* __restrict__ cand_pos,
                                                          int* __restrict__ cand_row) {
    __shared__ int s_piv[4];  // prow, ppos, occ, skip
    __shared__ double s_prow_vals[BASE_W];
    const int t = threadIdx.x;
    const int lane = t & (PANEL_ROWS - 1), grp = t / PANEL_ROWS;
    const int next_col = first ? j0 : k + 1;
    const bool want_next = next_col < c1;
    const size_t r = (size_t)j0 + (size_t)blockIdx.x * PANEL_ROWS + lane;  // grid covers every row once
    const int jbase = grp * PANEL_JT;
    const int ncols = c1 - k;  // panel columns k .. c1-1 (relative 0 .. ncols-1)
    const bool in_rows = r < rows;

    // ---- independent global reads issued together (ONE memory round trip)
    double v[PANEL_JT];
    double akk = 0.0, nextv = 0.0;
    int pos = (int)r;
    if (in_rows) {
        if (first) {
            if (grp == 0) {
                nextv = A[r + (size_t)next_col * lda];
                pos_of[r] = (int)r;
                row_at[r] = (int)r;
            }
        } else {
            pos = pos_of[r];
            akk = A[r + (size_t)k * lda];
#pragma unroll
            for (int jj = 0; jj < PANEL_JT; ++jj)
                if (jbase + jj < ncols) v[jj] = A[r + (size_t)(k + jbase + jj) * lda];
        }
    }
    if (!first) {
        if (grp == 0) {
            // ---- pivot of column k: fold the per-block candidates, order (abs desc, pos asc)
            double ba = 0.0;
            int bp = 0x7fffffff, br = -1;
            const int base = (k & 1) * MAX_PANEL_BLOCKS;
            const int occ = lane == 0 ? row_at[k] : 0;  // physical row at position k
            for (int b = lane; b < nblocks; b += PANEL_ROWS) {
                const double a = cand_abs[base + b];
                const int p = cand_pos[base + b];
                if (a > ba || (a == ba && a > 0.0 && p < bp)) {
                    ba = a;
                    bp = p;
                    br = cand_row[base + b];
                }
            }
            for (int off = PANEL_ROWS / 2; off > 0; off >>= 1) {
                const double oa = __shfl_down(ba, off, PANEL_ROWS);
                const int op = __shfl_down(bp, off, PANEL_ROWS);
                const int orow = __shfl_down(br, off, PANEL_ROWS);
                if (oa > ba || (oa == ba && oa > 0.0 && op < bp)) {
                    ba = oa;
                    bp = op;
                    br = orow;
                }
            }
            if (lane == 0) {
                if (!(ba > 0.0)) {  // all-zero (or NaN-only) column: pivot_row stays k (host_lu.rs:38)
                    br = occ;
                    bp = k;
                }
                const int skip = (ba <= LU_EPS) ? 1 : 0;
                s_piv[0] = br;
                s_piv[1] = bp;
                s_piv[2] = occ;
                s_piv[3] = skip;
                if (blockIdx.x == 0) {
                    ipiv[k] = bp;
                    prow_arr[k] = br;
                    row_at[bp] = occ;  // position k is final from now on; only the displaced row moves
                    if (skip) atomicAdd(info, 1);
                }
            }
        }
        __syncthreads();
        // ---- second (and last) dependent round trip: the pivot row's panel values
        if (t < ncols) s_prow_vals[t] = A[(size_t)s_piv[0] + (size_t)(k + t) * lda];
        __syncthreads();
        const int prow = s_piv[0];
        if (in_rows && pos >= 0) {
            if ((int)r == prow) {
                pos = -1;  // retires as row k of U
                if (grp == 0) pos_of[r] = -1;
            } else if ((int)r == s_piv[2]) {
                pos = s_piv[1];  // the old occupant of position k moves to the pivot's position
                if (grp == 0) pos_of[r] = pos;
            }
        }
        if (in_rows && pos >= 0) {
            if (s_piv[3]) {
                if (grp == 0) A[r + (size_t)k * lda] = 0.0;
            } else {
                const double factor = akk / s_prow_vals[0];
                if (grp == 0) A[r + (size_t)k * lda] = factor;
#pragma unroll
                for (int jj = 0; jj < PANEL_JT; ++jj) {
                    const int j = jbase + jj;
                    if (j >= 1 && j < ncols) {
                        const double prod = factor * s_prow_vals[j];
                        v[jj] = v[jj] - prod;
                        A[r + (size_t)(k + j) * lda] = v[jj];
                    }
                }
            }
            nextv = v[1];  // column k+1 lives in group 0 (only meaningful there)
        }
    }
    if (!want_next || grp != 0) return;
    double best = 0.0;
    int bpos = 0x7fffffff, brow = -1;
    if (in_rows && pos >= 0) {
        const double a = fabs(nextv);
        if (a > 0.0) {  // NaN or zero never wins (host_lu.rs: `abs > pivot_abs`)
            best = a;
            bpos = pos;
            brow = (int)r;
        }
    }
    for (int off = PANEL_ROWS / 2; off > 0; off >>= 1) {
        const double oa = __shfl_down(best, off, PANEL_ROWS);
        const int op = __shfl_down(bpos, off, PANEL_ROWS);
        const int orow = __shfl_down(brow, off, PANEL_ROWS);
        if (oa > best || (oa == best && oa > 0.0 && op < bpos)) {
            best = oa;
            bpos = op;
            brow = orow;
        }
    }
    if (lane == 0) {
        const int slot = (next_col & 1) * MAX_PANEL_BLOCKS + blockIdx.x;
        cand_abs[slot] = best;
        cand_pos[slot] = bpos;
        cand_row[slot] = brow;
    }
}

static constexpr int PLIST = 2 * BASE_W;  // per base panel: BASE_W pivot rows brought to the top + <= BASE_W displaced rows

// Turn the lazy bookkeeping of one finished base panel [j0, c1) into a list of row moves
// new[dst] = old[src]: position k receives the pivot row prow[k]; a top-block row that was not
// retired ends at pos_of[r].  Unused slots are (-1, -1).
__global__ void __launch_bounds__(PLIST) k_build_plist(int j0, int c1, const int* __restrict__ pos_of,
                                                       const int* __restrict__ prow_arr, int2* __restrict__ list) {
    const int t = threadIdx.x;
    int2 e = make_int2(-1, -1);
    const int w = c1 - j0;
    if (t < BASE_W) {
        if (t < w) {
            const int src = prow_arr[j0 + t];
            if (src != j0 + t) e = make_int2(j0 + t, src);
        }
    } else {
        const int i = t - BASE_W;
        if (i < w) {
            const int r = j0 + i;
            const int p = pos_of[r];
            if (p >= 0 && p != r) e = make_int2(p, r);
        }
    }
    list[t] = e;
}

// Apply the row moves of base panels [p0, p1) to columns [c0, c1): per panel one parallel gather
// (all loads), a barrier, then the stores.  Replaces w sequential dependent swaps per column by
// w/BASE_W phases with PLIST independent accesses each.
static constexpr int LASWP_COLS = 8;  // columns per block
__global__ void __launch_bounds__(2 * PLIST) k_laswp_lists(double* __restrict__ A, size_t lda, size_t c0, size_t c1,
                                                           const int2* __restrict__ lists, int p0, int p1) {
    const int i = threadIdx.x & (PLIST - 1);
    const int half = threadIdx.x / PLIST;  // 0 or 1
    const size_t cbase = c0 + (size_t)blockIdx.x * LASWP_COLS;
    for (int p = p0; p < p1; ++p) {
        const int2 e = lists[(size_t)p * PLIST + i];
        double vals[LASWP_COLS / 2];
#pragma unroll
        for (int q = 0; q < LASWP_COLS / 2; ++q) {
            const size_t cc = cbase + 2 * q + half;
            if (e.x >= 0 && cc < c1) vals[q] = A[(size_t)e.y + cc * lda];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LASWP_COLS / 2; ++q) {
            const size_t cc = cbase + 2 * q + half;
            if (e.x >= 0 && cc < c1) A[(size_t)e.x + cc * lda] = vals[q];
        }
        __syncthreads();
    }
}

// Small triangular solves (w <= TRSM_W = 128) in ONE launch: the triangle is staged in LDS as
// Ts[k][r] (r contiguous), one wave owns one right-hand-side column at a time and lane i holds
// x[i] and x[64 + i].  Step k broadcasts the finished x_k (uniform lane index -> v_readlane) and
// every lane eliminates it from the rows it still owns, reading its multipliers from LDS
// (512 contiguous bytes per wave: conflict free).  Before this kernel the same solve took seven
// launches (four 32-wide substitutions + three MFMA dgemms with k <= 64), and at n = 16384 those
// small dgemms alone cost ~35 ms of a 246 ms factorisation.
//   MODE 0: lower, implicit unit diagonal (LU factors)        B <- L^-1 B
//   MODE 1: lower, stored diagonal (`linsolve` LT, linsolve.rs:769-800)
//   MODE 2: upper, stored diagonal                             B <- U^-1 B
static constexpr int TRSM_THREADS = 512;
static constexpr int TRSM_SW = TRSM_W + 1;  // LDS row stride (doubles)

// value of lane `lane` (wave-uniform index) in every lane: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(TRSM_THREADS) k_trsm_fused(const double* __restrict__ T, size_t ldt, int w,
                                                             double* __restrict__ B, size_t ldb, size_t ncols) {
    extern __shared__ double Ts[];  // [w][TRSM_SW]
    for (int idx = threadIdx.x; idx < w * TRSM_W; idx += TRSM_THREADS) {
        const int r = idx & (TRSM_W - 1), k = idx / TRSM_W;
        Ts[k * TRSM_SW + r] = r < w ? T[r + (size_t)k * ldt] : 0.0;
    }
    __syncthreads();
    const int i = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * TRSM_THREADS + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * TRSM_THREADS) >> 6;
    const bool two = w > 64;
    double d0 = 1.0, d1 = 1.0;
    if (MODE != 0) {
        if (i < w) d0 = Ts[i * TRSM_SW + i];
        if (64 + i < w) d1 = Ts[(64 + i) * TRSM_SW + 64 + i];
    }
    for (size_t cc = wave; cc < ncols; cc += nwaves) {
        double* b = B + cc * ldb;
        double x0 = i < w ? b[i] : 0.0;
        double x1 = 64 + i < w ? b[64 + i] : 0.0;
        // selects, not branches and not zero multipliers: 0 * inf must not poison finished lanes
        if (MODE != 2) {
            const int k0end = w < 64 ? w : 64;
#pragma unroll 1
            for (int k = 0; k < k0end; ++k) {
                const double xk = bcast_lane(MODE == 0 ? x0 : x0 / d0, k);  // final x_k
                const double l0 = Ts[k * TRSM_SW + i], l1 = Ts[k * TRSM_SW + 64 + i];
                const double u0 = x0 - l0 * xk;
                x0 = (MODE != 0 && i == k) ? xk : (i > k ? u0 : x0);
                x1 = two ? x1 - l1 * xk : x1;
            }
#pragma unroll 1
            for (int k = 64; k < w; ++k) {
                const double xk = bcast_lane(MODE == 0 ? x1 : x1 / d1, k - 64);
                const double l1 = Ts[k * TRSM_SW + 64 + i];
                const double u1 = x1 - l1 * xk;
                x1 = (MODE != 0 && 64 + i == k) ? xk : (64 + i > k ? u1 : x1);
            }
        } else {
#pragma unroll 1
            for (int k = w - 1; k >= 64; --k) {
                const double xk = bcast_lane(x1 / d1, k - 64);
                const double u0 = Ts[k * TRSM_SW + i], u1 = Ts[k * TRSM_SW + 64 + i];
                const double v1 = x1 - u1 * xk;
                x1 = (64 + i == k) ? xk : (64 + i < k ? v1 : x1);
                x0 -= u0 * xk;
            }
#pragma unroll 1
            for (int k = (w < 64 ? w : 64) - 1; k >= 0; --k) {
                const double xk = bcast_lane(x0 / d0, k);
                const double u0 = Ts[k * TRSM_SW + i];
                const double v0 = x0 - u0 * xk;
                x0 = (i == k) ? xk : (i < k ? v0 : x0);
            }
        }
        if (i < w) b[i] = x0;
        if (64 + i < w) b[64 + i] = x1;
    }
}

static int launch_check(Context* c);
template <int MODE>
static int launch_trsm_fused(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    const size_t lds_bytes = w * TRSM_SW * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_trsm_fused<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(TRSM_W * TRSM_SW * sizeof(double)));
        attr_set = true;
    }
    size_t want = (nc + 7) / 8;  // 8 waves per block, one column each per pass
    const size_t cap = (size_t)c->num_cus * (w <= 64 ? 2 : 1);
    if (want < 1) want = 1;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(k_trsm_fused<MODE>, dim3(grid), dim3(TRSM_THREADS), lds_bytes, c->stream, T, ldt, (int)w, B, ldb, nc);
    return launch_check(c);
}

// Developer knob: RMHIP_LU_SKIP bitmask drops whole phases (results are then garbage) so wall-clock
// differences attribute time to phases without a profiler: 1 column kernels, 2 dgemm, 4 trsm base
// kernels, 8 row interchanges.
static int lu_skip_mask() {
    static int mask = -1;
    if (mask < 0) {
        const char* v = std::getenv("RMHIP_LU_SKIP");
        mask = v ? std::atoi(v) : 0;
    }
    return mask;
}
static int lu_dgemm(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda, const double* B,
                    size_t ldb, double beta, double* C, size_t ldc) {
    if (lu_skip_mask() & 2) return RMHIP_OK;
    static long kmin = -1, mmax = -1;
    if (kmin < 0) {
        const char* v = std::getenv("RMHIP_LU_SKIP_K_BELOW");
        kmin = v ? std::atol(v) : 0;
        v = std::getenv("RMHIP_LU_SKIP_M_BELOW");
        mmax = v ? std::atol(v) : 0;
    }
    if ((long)k < kmin || (long)m < mmax) return RMHIP_OK;
    return launch_dgemm(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

static int launch_check(Context* c) {
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// split point of a triangular solve wider than TRSM_W: whole base blocks on the left
static size_t trsm_split(size_t w) {
    size_t h = ((w / 2 + TRSM_W - 1) / TRSM_W) * TRSM_W;
    if (h >= w) h = w / 2;
    return h;
}

// B[w x nc] <- L^-1 B with L = unit-lower part of T[w x w]; recursive halving, dgemm in between.
static int trsm_lower_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc,
                          bool unit = true) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= (size_t)TRSM_W) {
        if (lu_skip_mask() & 4) return RMHIP_OK;
        return unit ? launch_trsm_fused<0>(c, T, ldt, w, B, ldb, nc) : launch_trsm_fused<1>(c, T, ldt, w, B, ldb, nc);
    }
    const size_t h = trsm_split(w);
    RMHIP_TRY(trsm_lower_rec(c, T, ldt, h, B, ldb, nc, unit));
    RMHIP_TRY(lu_dgemm(c, w - h, nc, h, -1.0, T + h, ldt, B, ldb, 1.0, B + h, ldb));
    return trsm_lower_rec(c, T + h + h * ldt, ldt, w - h, B + h, ldb, nc, unit);
}

static int trsm_upper_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= (size_t)TRSM_W) {
        if (lu_skip_mask() & 4) return RMHIP_OK;
        return launch_trsm_fused<2>(c, T, ldt, w, B, ldb, nc);
    }
    const size_t h = trsm_split(w);
    RMHIP_TRY(trsm_upper_rec(c, T + h + h * ldt, ldt, w - h, B + h, ldb, nc));
    RMHIP_TRY(lu_dgemm(c, h, nc, w - h, -1.0, T + h * ldt, ldt, B + h, ldb, 1.0, B, ldb));
    return trsm_upper_rec(c, T, ldt, h, B, ldb, nc);
}

static int laswp(LuState& s, size_t c0, size_t c1, size_t k0, size_t k1) {
    if (c1 <= c0 || k1 <= k0) return RMHIP_OK;
    // base panels whose first column lies in [k0, k1): recursion ranges are unions of whole panels
    const std::vector<size_t>& ps = *s.panel_start;
    int p0 = -1, p1 = -1;
    for (size_t i = 0; i < ps.size(); ++i) {
        if (ps[i] >= k0 && ps[i] < k1) {
            if (p0 < 0) p0 = (int)i;
            p1 = (int)i + 1;
        }
    }
    if (p0 < 0 || (lu_skip_mask() & 8)) return RMHIP_OK;
    const size_t ncols = c1 - c0;
    hipLaunchKernelGGL(k_laswp_lists, dim3((unsigned)((ncols + LASWP_COLS - 1) / LASWP_COLS)), dim3(2 * PLIST), 0,
                       s.c->stream, s.A, s.lda, c0, c1, s.plist, p0, p1);
    return launch_check(s.c);
}

// Factor columns [j0, j0+w) over rows [j0, rows); swaps are applied inside that column range only.
static int getrf_rec(LuState& s, size_t j0, size_t w) {
    if (w == 0 || j0 >= s.rows) return RMHIP_OK;
    if (w <= (size_t)BASE_W) {
        const size_t c1 = j0 + w;  // j0 + w <= min(rows, cols) always holds (see lu_factor_device)
        const size_t nb = (s.rows - j0 + PANEL_ROWS - 1) / PANEL_ROWS;  // one block per PANEL_ROWS rows
        if (nb > (size_t)MAX_PANEL_BLOCKS)
            return fail(RMHIP_ERR_UNSUPPORTED, "lu: more than %d rows per panel not supported yet", MAX_PANEL_BLOCKS * PANEL_ROWS);
        // init launch: candidates for column j0; then one launch per column
        hipLaunchKernelGGL(k_lu_col, dim3((unsigned)nb), dim3(PANEL_THREADS), 0, s.c->stream, s.A, s.lda, s.rows, (int)j0,
                           (int)j0 - 1, (int)c1, 1, (int)nb, s.pos_of, s.row_at, s.prow, s.ipiv, s.info, s.cand_abs, s.cand_pos, s.cand_row);
        RMHIP_TRY(launch_check(s.c));
        for (size_t k = j0; k < c1 && !(lu_skip_mask() & 1); ++k) {
            hipLaunchKernelGGL(k_lu_col, dim3((unsigned)nb), dim3(PANEL_THREADS), 0, s.c->stream, s.A, s.lda, s.rows,
                               (int)j0, (int)k, (int)c1, 0, (int)nb, s.pos_of, s.row_at, s.prow, s.ipiv, s.info,
                               s.cand_abs, s.cand_pos, s.cand_row);
            RMHIP_TRY(launch_check(s.c));
        }
        // the panel's row moves as one gather list, then the physical interchange of the panel columns
        const size_t pid = s.panel_start->size();
        s.panel_start->push_back(j0);
        hipLaunchKernelGGL(k_build_plist, dim3(1), dim3(PLIST), 0, s.c->stream, (int)j0, (int)c1, s.pos_of, s.prow,
                           s.plist + pid * PLIST);
        RMHIP_TRY(launch_check(s.c));
        return laswp(s, j0, c1, j0, c1);
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
        RMHIP_TRY(lu_dgemm(s.c, s.rows - j0 - h, w - h, h, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda));
        RMHIP_TRY(getrf_rec(s, j0 + h, w - h));
        const size_t k1 = (j0 + w <= s.rows) ? (j0 + w) : s.rows;
        RMHIP_TRY(laswp(s, j0, j0 + h, j0 + h, k1));
    }
    return RMHIP_OK;
}

// ---- blocked driver with look-ahead -------------------------------------------------------------
// Panels (width nb, factored recursively by getrf_rec) are latency bound: one launch per column.
// Trailing updates are throughput bound (MFMA dgemm).  Running them back to back on one stream adds
// the two; here the update of everything right of the NEXT panel runs on a second, lower-priority
// stream while the main stream already factors the next panel:
//   main:  P_j -> LA_j (swap + trsm + gemm of the next panel's nb columns) -> P_{j+1} -> wait(S_j) -> LA_{j+1} ...
//   side:  wait(P_j) -> S_j (swap + trsm + gemm of columns right of the next panel; swaps of the finished left columns) ...
struct StreamScope {
    Context* c;
    hipStream_t saved;
    StreamScope(Context* ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
    ~StreamScope() { c->stream = saved; }
};

static int update_columns(LuState& s, size_t j, size_t w, size_t c0, size_t c1) {
    // columns [c0, c1) receive the row interchanges of panel [j, j+w), the U block row and the Schur update
    if (c1 <= c0) return RMHIP_OK;
    RMHIP_TRY(laswp(s, c0, c1, j, j + w));
    double* A11 = s.A + j + j * s.lda;
    double* A12 = s.A + j + c0 * s.lda;
    RMHIP_TRY(trsm_lower_rec(s.c, A11, s.lda, w, A12, s.lda, c1 - c0));
    if (j + w < s.rows) {
        double* A21 = s.A + (j + w) + j * s.lda;
        double* A22 = s.A + (j + w) + c0 * s.lda;
        RMHIP_TRY(lu_dgemm(s.c, s.rows - j - w, c1 - c0, w, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda));
    }
    return RMHIP_OK;
}

static int getrf_blocked(LuState& s, size_t kmin, size_t nb) {
    Context* c = s.c;
    hipStream_t main_stream = c->stream;
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipStream_t side = nullptr;
    RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_low));
    std::vector<hipEvent_t> events;
    auto new_event = [&]() {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        events.push_back(e);
        return e;
    };
    int rc = RMHIP_OK;
    hipEvent_t side_done = nullptr;  // S_{j-1} finished
    {
        hipEvent_t e0 = new_event();  // side starts after whatever main already has queued (the copy of A)
        (void)hipEventRecord(e0, main_stream);
        (void)hipStreamWaitEvent(side, e0, 0);
    }
    for (size_t j = 0; j < kmin && rc == RMHIP_OK; j += nb) {
        const size_t w = (kmin - j) < nb ? (kmin - j) : nb;
        rc = getrf_rec(s, j, w);  // P_j on main
        if (rc != RMHIP_OK) break;
        hipEvent_t panel_done = new_event();
        (void)hipEventRecord(panel_done, main_stream);
        const size_t next = j + w;
        size_t la_w = 0;
        if (next < kmin) {  // there is a next panel: update its columns on main right away
            la_w = (kmin - next) < nb ? (kmin - next) : nb;
            if (side_done) (void)hipStreamWaitEvent(main_stream, side_done, 0);
            rc = update_columns(s, j, w, next, next + la_w);
            if (rc != RMHIP_OK) break;
        }
        (void)hipStreamWaitEvent(side, panel_done, 0);
        {
            StreamScope scope(c, side);
            rc = update_columns(s, j, w, next + la_w, s.cols);      // S_j
            if (rc == RMHIP_OK && j > 0) rc = laswp(s, 0, j, j, j + w);  // finished left columns
        }
        side_done = new_event();
        (void)hipEventRecord(side_done, side);
    }
    if (side_done) (void)hipStreamWaitEvent(main_stream, side_done, 0);
    (void)hipStreamSynchronize(side);
    (void)hipStreamSynchronize(main_stream);
    for (hipEvent_t e : events)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(side);
    return rc;
}

// In-place LU of A (rows x cols, lda). perm_dev[rows] receives the row permutation as the
// reference reports it (perm[k] = original row now at position k, host_lu.rs:50,107).
// *info_host = number of pivots that hit the singular cut-off.
int lu_factor_device(Context* c, double* A, size_t rows, size_t cols, size_t lda, int* perm_dev, int* info_host,
                     std::vector<int>* ipiv_host) {
    const size_t kmin = rows < cols ? rows : cols;
    if (rows > 0x7fffffffULL || cols > 0x7fffffffULL) return fail(RMHIP_ERR_UNSUPPORTED, "lu: dimension exceeds 2^31");
    // one device block: ipiv[rows] | info | pos_of | row_at | prow | panel lists | cand_abs[2*MAXB] | cand_pos | cand_row
    const size_t n_int = rows + 4;
    const size_t isz = (rows * sizeof(int) + 15) & ~(size_t)15;
    const size_t max_panels = kmin / 16 + 2;  // base panels are >= 16 columns wide except possibly the last
    const size_t off_posof = (n_int * sizeof(int) + 15) & ~(size_t)15;
    const size_t off_rowat = off_posof + isz;
    const size_t off_prow = off_rowat + isz;
    const size_t off_plist = off_prow + isz;
    const size_t off_abs = off_plist + max_panels * PLIST * sizeof(int2);
    const size_t off_pos = off_abs + sizeof(double) * 2 * MAX_PANEL_BLOCKS;
    const size_t off_row = off_pos + sizeof(int) * 2 * MAX_PANEL_BLOCKS;
    const size_t total = off_row + sizeof(int) * 2 * MAX_PANEL_BLOCKS;
    char* blk = nullptr;
    RMHIP_HIP_CHECK(hipMalloc((void**)&blk, total));
    int* ipiv = (int*)blk;
    int* info = ipiv + rows;
    hipError_t e = hipMemsetAsync(blk, 0, total, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(blk);
        return fail(RMHIP_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    std::vector<size_t> panel_start;
    panel_start.reserve(max_panels);
    LuState s{c, A, rows, cols, lda, ipiv, info, (int*)(blk + off_posof), (int*)(blk + off_rowat), (int*)(blk + off_prow),
              (int2*)(blk + off_plist), &panel_start, (double*)(blk + off_abs), (int*)(blk + off_pos), (int*)(blk + off_row)};
    size_t nb = 512;
    if (const char* v = std::getenv("RMHIP_LU_NB")) nb = (size_t)std::atoll(v);
    nb = nb < 64 ? 64 : (nb / 64) * 64;
    // Look-ahead (second stream) is opt-in: measured on MI355X it does not yet beat the single-stream
    // recursion, because the side stream's dgemm blocks (2 x 218 VGPRs per SIMD) leave no room for the
    // 512-thread column kernels, which then wait for a dgemm block to retire.
    const char* la = std::getenv("RMHIP_LU_LOOKAHEAD");
    const bool blocked = kmin > nb && la && la[0] == '1';
    int rc = blocked ? getrf_blocked(s, kmin, nb) : getrf_rec(s, 0, kmin);
    if (rc == RMHIP_OK && cols > rows && !blocked) {  // wide: finish U's right block (the blocked driver covers it)
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
    (void)hipFree(blk);
    if (rc != RMHIP_OK) return rc;
    if (info_host) *info_host = h_ipiv[rows];
    if (ipiv_host) ipiv_host->assign(h_ipiv.begin(), h_ipiv.begin() + (long)kmin);
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

// Arbitrary row permutation of a block of columns, given as one (dst, src) list: every load of a
// column precedes every store (one block per column, list entries spread over the threads).
static constexpr int PERM_PER_THREAD = 8;  // up to 256*8 = 2048 moved rows per call
__global__ void __launch_bounds__(256) k_permute_rows(double* __restrict__ A, size_t lda, size_t ncols,
                                                      const int2* __restrict__ list, int len) {
    const size_t cc = blockIdx.x;
    if (cc >= ncols) return;
    double* col = A + cc * lda;
    double vals[PERM_PER_THREAD];
    int dst[PERM_PER_THREAD];
#pragma unroll
    for (int q = 0; q < PERM_PER_THREAD; ++q) {
        const int i = threadIdx.x + 256 * q;
        dst[q] = -1;
        if (i < len) {
            const int2 e = list[i];
            dst[q] = e.x;
            vals[q] = col[e.y];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PERM_PER_THREAD; ++q)
        if (dst[q] >= 0) col[dst[q]] = vals[q];
}

// Apply the sequential interchanges k <-> ipiv[k] (k = 0..npiv-1, rows relative to A) to ncols columns.
// The swaps are composed on the host into one permutation of the touched rows.
int lu_swap_rows_device(Context* c, double* A, size_t lda, size_t ncols, const std::vector<int>& ipiv) {
    if (ncols == 0 || ipiv.empty()) return RMHIP_OK;
    std::unordered_map<int, int> content;  // position -> original row now stored there
    auto get = [&](int p) {
        auto it = content.find(p);
        return it == content.end() ? p : it->second;
    };
    for (size_t k = 0; k < ipiv.size(); ++k) {
        const int p = ipiv[k];
        if (p == (int)k) continue;
        const int a = get((int)k), b = get(p);
        content[(int)k] = b;
        content[p] = a;
    }
    std::vector<int2> list;
    for (const auto& kv : content)
        if (kv.first != kv.second) list.push_back(make_int2(kv.first, kv.second));
    if (list.empty()) return RMHIP_OK;
    if (list.size() > (size_t)256 * PERM_PER_THREAD)
        return fail(RMHIP_ERR_UNSUPPORTED, "swap_rows: more than %d moved rows per call", 256 * PERM_PER_THREAD);
    int2* dlist = nullptr;
    RMHIP_HIP_CHECK(hipMalloc((void**)&dlist, sizeof(int2) * list.size()));
    hipError_t e = hipMemcpyAsync(dlist, list.data(), sizeof(int2) * list.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        for (size_t c0 = 0; c0 < ncols; c0 += 65535) {
            const size_t nc = (ncols - c0) < 65535 ? (ncols - c0) : 65535;
            hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)nc), dim3(256), 0, c->stream, A + c0 * lda, lda, nc, dlist,
                               (int)list.size());
        }
        e = hipGetLastError();
        c->tel.kernel_launches++;
    }
    (void)hipStreamSynchronize(c->stream);  // the host list must outlive the copy; dlist is freed next
    (void)hipFree(dlist);
    if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "swap_rows: %s", hipGetErrorString(e));
    return RMHIP_OK;
}

int trsm_lower_unit_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    return trsm_lower_rec(c, T, ldt, w, B, ldb, nc);
}
int trsm_upper_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    return trsm_upper_rec(c, T, ldt, w, B, ldb, nc);
}
int trsm_lower_nonunit_device(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    return trsm_lower_rec(c, T, ldt, w, B, ldb, nc, false);
}

// ---- transpose (AccelProvider::transpose, lib.rs; also `linsolve` TRANSA, linsolve.rs:698-705) -----
// 64x64 tiles through LDS (row stride 65 doubles: conflict-free both ways); a wave reads 512
// contiguous bytes of a source column and writes 512 contiguous bytes of a destination column.
static constexpr int TR_TILE = 64;
__global__ void __launch_bounds__(256) k_transpose(const double* __restrict__ src, size_t lds_, size_t rows, size_t cols,
                                                   double* __restrict__ dst, size_t ldd) {
    __shared__ double tile[TR_TILE][TR_TILE + 1];
    const size_t r0 = (size_t)blockIdx.x * TR_TILE, c0 = (size_t)blockIdx.y * TR_TILE;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
#pragma unroll 4
    for (int p = 0; p < TR_TILE / 4; ++p) {
        const int cc = grp + 4 * p;
        if (r0 + lane < rows && c0 + cc < cols) tile[cc][lane] = src[(r0 + lane) + (c0 + cc) * lds_];
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < TR_TILE / 4; ++p) {
        const int rr = grp + 4 * p;
        if (c0 + lane < cols && r0 + rr < rows) dst[(c0 + lane) + (r0 + rr) * ldd] = tile[lane][rr];
    }
}

int transpose_device(Context* c, const double* src, size_t lds_, size_t rows, size_t cols, double* dst, size_t ldd) {
    if (rows == 0 || cols == 0) return RMHIP_OK;
    const size_t gx = (rows + TR_TILE - 1) / TR_TILE, gy = (cols + TR_TILE - 1) / TR_TILE;
    if (gy > 65535) return fail(RMHIP_ERR_UNSUPPORTED, "transpose: more than %d columns", 65535 * TR_TILE);
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, src, lds_, rows, cols, dst, ldd);
    return launch_check(c);
}

// min |d_ii|, max |d_ii| and the number of exact zeros on the diagonal (linsolve.rs:776-786: a zero
// diagonal entry is the "singular to working precision" error; rcond = min/max, linalg.rs:232-238).
__global__ void __launch_bounds__(256) k_diag_stats(const double* __restrict__ A, size_t lda, size_t n, double* __restrict__ out3) {
    __shared__ double s_min[256], s_max[256], s_zero[256];
    double mn = __builtin_inf(), mx = 0.0, z = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const double a = fabs(A[i + i * lda]);
        mn = fmin(mn, a);  // f64::min / max: a NaN operand is ignored
        mx = fmax(mx, a);
        if (a == 0.0) z += 1.0;
    }
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    s_zero[threadIdx.x] = z;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + off]);
            s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + off]);
            s_zero[threadIdx.x] += s_zero[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out3[0] = s_min[0];
        out3[1] = s_max[0];
        out3[2] = s_zero[0];
    }
}

int diag_stats_device(Context* c, const double* A, size_t lda, size_t n, double* min_abs, double* max_abs, size_t* zeros) {
    double* d = nullptr;
    RMHIP_HIP_CHECK(hipMalloc((void**)&d, 3 * sizeof(double)));
    hipLaunchKernelGGL(k_diag_stats, dim3(1), dim3(256), 0, c->stream, A, lda, n, d);
    double h[3] = {0, 0, 0};
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    c->tel.kernel_launches++;
    if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "diag_stats: %s", hipGetErrorString(e));
    *min_abs = h[0];
    *max_abs = h[1];
    *zeros = (size_t)h[2];
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
