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
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"
#include <chrono>
#include <string>

namespace rmhip {

// Wave priority of the panel chain's kernels (s_setprio): they share CUs with the update stream's dgemm blocks, and every
// instruction they wait to issue is on the critical path.  Set once per process (lu_factor_device); RMHIP_LU_CHAIN_PRIO=0 turns it
// off.  Measured at n = 16384: 74.0-74.1 ms with, 74.6-75.2 without; with the panel block sharing its CU (no LDS padding) 75.1
// against 77.4 - the priority recovers most of what sharing costs, a CU of its own is still better.
// (XCC id, HW_ID cu / sh / se byte) of the CU this wave runs on, bit 31 set: never 0
__device__ __forceinline__ unsigned cu_key() {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return 0x80000000u | ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);
}
__device__ int d_chain_prio = 0;
__device__ __forceinline__ void chain_prio() {
    if (d_chain_prio) __builtin_amdgcn_s_setprio(3);
}


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
    // persistent panel kernel (k_lu_panel2): inter-block exchange area
    int* xerr;            // device: set when a bounded spin expired (blocks not co-resident)
    unsigned long long* xrec;   // device [2][PK_MAXB][2]: candidate records (16-byte granules, see k_lu_panel2)
    unsigned long long* xvals;  // device [2][PK_MAXB][BASE_W][2]: candidate rows' panel values (tagged half words)
    unsigned xbase;       // host: panel columns factored so far (exchange step counter)
    bool persistent;      // use k_lu_panel2 for base panels
    unsigned long long* xdbg;  // device [16] phase ticks (RMHIP_LU_PANEL_DEBUG=1) or nullptr
    long panel_pad_kb = -1;    // extra LDS a panel block asks for (-1: the default, see getrf_rec)
    int* panel_xcc = nullptr;  // device word: the XCD of the last one-XCD panel (-1 otherwise)
    // solve path (k_rp_top + k_rp_below): pivoting restricted to the panel's top block, multipliers checked against tau
    bool fast = false;
    double tau = 8.0;
    double* ucomp = nullptr;              // device [BASE_W][BASE_W]: the pivot rows of the panel in flight
    unsigned long long* growth = nullptr; // device: bits of the largest multiplier below the top block so far
    bool screened = false;                // the first panel's multipliers were checked on the host (early way out)
    // band split (look-ahead driver, solve path): rows below band_end of the panel in flight are nobody's dependency until the NEXT
    // panel's look-ahead update, so their share of the full-height kernels (k_rp_below, the in-panel dgemm) runs on `aux` and the main
    // stream - the critical chain of k_rp_top launches - only touches rows [j, band_end)
    hipStream_t aux = nullptr;
    size_t band_end = 0;                  // 0: no split
    size_t ev_used = 0;                   // events drawn from the context's pool by this factorisation
    unsigned ucomp_slot = 0;              // ring of compact U copies: k_rp_below on `aux` may still read panel p's while k_rp_top writes p + 1's
    hipEvent_t aux_tail = nullptr;        // last event recorded on aux (what the main stream's next look-ahead update has to wait for)
    double* linv = nullptr;               // solve path: inverted 16 x 16 diagonal blocks of L, block q at 256 q (k_trsm_lower_mfma)
    unsigned* yield_word = nullptr;       // two-level driver: the yield table (common.h CuAnnounce) through which chain kernels ask the update blocks on their CU to pause
    int yield_all = 0;                    // which chain kernels besides k_rp_top count themselves in: 1 main-stream dgemm, 2 k_rp_below_mfma, 4 k_trsm_lower_mfma
    unsigned long long* minv_max = nullptr;  // two-level driver: bits of the largest |entry| of the super-panels' inverted L11 blocks (guard, see getrf_super)
};
static constexpr unsigned kUcompSlots = 16;  // >= base panels per look-ahead panel (512 / 64) with room to spare
// one slot: the 64 x 64 compact U of k_rp_top, then the inverses of its four 16 x 16 diagonal blocks (row-major [k][i], zero below the
// diagonal) for k_rp_below_mfma
static constexpr size_t UCOMP_STRIDE = (size_t)BASE_W * BASE_W + 4 * 256;

static hipEvent_t lu_new_event(LuState& s) {
    Context* c = s.c;
    if (s.ev_used == c->lu_events.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        c->lu_events.push_back(e);
    }
    return c->lu_events[s.ev_used++];
}

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

// ---- base panel, persistent variant: ONE launch per panel ---------------------------------------------
// The per-column kernel above pays a launch gap, a cold instruction fetch and two or three dependent
// trips to memory per column (~6.9 us measured, 113 ms of the 219 ms at n = 16384).  Here the whole
// panel (<= 64 columns) is factored by one grid of co-resident workgroups (256 rows each) that meet once
// per column:
//   * per column: block arg-max -> wave 0 publishes the block's candidate (|a|, position, row) AND that
//     row's remaining panel values -> every block folds the P candidates and takes the winner's values
//     -> every thread eliminates.  One exchange per column.
//   * every cross-block access is a write-through store / L1-bypassing load (sc1: coherent across the
//     eight XCD L2s) of a self-describing granule; there is no arrival counter and no fence.
//   * spins are bounded: if the blocks are not co-resident (device shared with another context) the
//     kernel sets *xerr and the factorisation fails loudly instead of hanging.
// Pivot rule, tie-break, singular cut-off and the lazy-pivoting bookkeeping are those of k_lu_col.
static constexpr int PK_MAXB = 256;              // at most one block per CU
static constexpr int PK_SPIN_LIMIT = 400000;
typedef unsigned long long pk_u64;

// ---- wave reductions on the DPP network (a __shfl_down tree is ds_bpermute based: ~0.4 us per
// 3-value arg-max, measured; these are a dozen VALU instructions).  Result is wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ pk_u64 dpp_u64(pk_u64 ident, pk_u64 v) {
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)ident, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(ident >> 32), (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((pk_u64)(unsigned)hi << 32) | (pk_u64)(unsigned)lo;
}
__device__ __forceinline__ pk_u64 umax64(pk_u64 a, pk_u64 b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ pk_u64 wave_max_u64(pk_u64 v) {
    v = umax64(v, dpp_u64<0x111, 0xf>(0, v));  // row_shr:1
    v = umax64(v, dpp_u64<0x112, 0xf>(0, v));  // row_shr:2
    v = umax64(v, dpp_u64<0x114, 0xf>(0, v));  // row_shr:4
    v = umax64(v, dpp_u64<0x118, 0xf>(0, v));  // row_shr:8  -> lane 15 of every row holds the row maximum
    v = umax64(v, dpp_u64<0x142, 0xa>(0, v));  // row_bcast:15 into rows 1 and 3
    v = umax64(v, dpp_u64<0x143, 0xc>(0, v));  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((pk_u64)hi << 32) | lo;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
    v = umin32(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// arg-max over the wave by (key descending, pos ascending); key == 0 never wins.  Returns the
// winning lane (wave-uniform) or -1; *key_out / *pos_out receive the winning pair.
__device__ __forceinline__ int wave_argmax(pk_u64 key, unsigned pos, pk_u64* key_out, unsigned* pos_out) {
    const pk_u64 m = wave_max_u64(key);
    const unsigned pm = wave_min_u32((key == m && m != 0) ? pos : 0xffffffffu);
    *key_out = m;
    *pos_out = pm;
    if (m == 0) return -1;
    const pk_u64 hit = __ballot(key == m && pos == pm);
    return (int)__builtin_ctzll(hit);
}

static constexpr int PLIST = 2 * BASE_W;  // per base panel: BASE_W pivot rows brought to the top + <= BASE_W displaced rows

// Faster arg-max for the register panel: the keys are non-negative doubles, so the maximum is six v_max_f64 steps on
// DPP-shifted copies; the (key descending, pos ascending) winner is then the only lane that holds the maximum except
// on exact ties, and only those pay the second (minimum position) reduction.  The general routine above runs both
// reductions back to back: ~0.27 us of dependent VALU latency with a single wave on the SIMD (measured), three times
// per column.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_argmax_f64(double key, unsigned pos, double* key_out, unsigned* pos_out) {
    double m = key;
    m = fmax(m, dpp_f64<0x111, 0xf>(m));
    m = fmax(m, dpp_f64<0x112, 0xf>(m));
    m = fmax(m, dpp_f64<0x114, 0xf>(m));
    m = fmax(m, dpp_f64<0x118, 0xf>(m));
    m = fmax(m, dpp_f64<0x142, 0xa>(m));
    m = fmax(m, dpp_f64<0x143, 0xc>(m));
    const double mx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 63), __builtin_amdgcn_readlane(__double2loint(m), 63));
    *key_out = mx;
    if (!(mx > 0.0)) {
        *pos_out = 0xffffffffu;
        return -1;
    }
    pk_u64 hit = __ballot(key == mx);
    if (__popcll(hit) > 1) {  // exact tie: first position wins (host_lu.rs:38-47)
        const unsigned pm = wave_min_u32(key == mx ? pos : 0xffffffffu);
        hit = __ballot(key == mx && pos == pm);
    }
    const int wl = (int)__builtin_ctzll(hit);
    *pos_out = (unsigned)__builtin_amdgcn_readlane((int)pos, wl);
    return wl;
}
// the same over lanes 0..3 only (the four wave records of a block): two quad steps
__device__ __forceinline__ int quad_argmax_f64(double key, unsigned pos, double* key_out, unsigned* pos_out) {
    double m = key;
    m = fmax(m, dpp_f64<0xb1, 0xf>(m));  // quad_perm [1,0,3,2]
    m = fmax(m, dpp_f64<0x4e, 0xf>(m));  // quad_perm [2,3,0,1]
    const double mx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), 0), __builtin_amdgcn_readlane(__double2loint(m), 0));
    *key_out = mx;
    if (!(mx > 0.0)) {
        *pos_out = 0xffffffffu;
        return -1;
    }
    const int lane = threadIdx.x & 63;
    pk_u64 hit = __ballot(lane < 4 && key == mx);
    if (__popcll(hit) > 1) {
        unsigned pm = 0xffffffffu;
        for (int l = 0; l < 4; ++l)
            if ((hit >> l) & 1) {
                const unsigned pl = (unsigned)__builtin_amdgcn_readlane((int)pos, l);
                pm = pl < pm ? pl : pm;
            }
        hit = __ballot(lane < 4 && key == mx && pos == pm);
    }
    const int wl = (int)__builtin_ctzll(hit);
    *pos_out = (unsigned)__builtin_amdgcn_readlane((int)pos, wl);
    return wl;
}

// ---- base panel, persistent variant 2: rows in REGISTERS, one-hop exchange ---------------------------------
// Measured on this kernel's LDS-resident predecessor (round 1; n = 16384, in-kernel ticks): 4.0 us per column = block
// arg-max 0.45 + publish 0.40 + wait/fold 1.85 + winner's row 0.40 + elimination 0.87, nearly independent of the
// block count (4 blocks: 3.7 us).  The exchange itself is a memory-system hop (scripts/micro/xcd_exchange.hip: 0.7-1 us
// per dependent hop idle, ~1.5 under the update stream's traffic, same XCD or not), so this variant removes what
// surrounds it.  What a column costs locally is instructions per wave x waves per SIMD (the code is a serial chain of
// short phases), so the block is FOUR waves - one per SIMD - and each thread does a whole row:
//   * thread t owns row t of the block: its panel values sit in 64 REGISTERS (a[i] = column 4*jj + i; the window
//     is shifted by four every four columns, so inside the four unrolled column steps every register index is
//     static).  Elimination = one division + (63 - k) multiply-subtracts on registers with the pivot row read from LDS
//     (broadcast reads, immediate offsets).  A first version with four threads per row (16 waves) spent 2.4 us per
//     column in local work: every wave runs the whole per-column instruction stream, four to a SIMD.
//   * every wave's arg-max winner dumps its row to LDS before the block barrier, wave 0 publishes the block's
//     candidate from there (conflict-free read; the predecessor read S[lane][bt]: a 64-way bank conflict).
//   * ONE hop for <= 32 blocks: every wave polls eight blocks' candidate rows (16-byte granules: two tagged half-words,
//     one dwordx4 access, four in flight) while wave 0 folds the records, so the winner's row is already in LDS when
//     it is known.  More blocks (tall panels, off the critical path under look-ahead) keep the second dependent read.
//   * finished columns leave the register window straight to memory at the row's ORIGINAL position (coalesced, fire and
//     forget); the panel's own interchange is then one more column range for the k_laswp_lists call that moves the
//     neighbouring columns anyway (getrf_rec), and the row-move list is written by the threads that own the moves (no
//     k_build_plist launch).  An earlier version parked the finished columns in LDS (128 KiB) to write them at their
//     final positions itself: the block then needed a whole CU and had to wait for one to drain under look-ahead; with
//     18.5 KiB it starts beside the update stream's dgemm block.
// Pivot rule, tie-break, cut-off, lazy bookkeeping and the arithmetic (division, unfused multiply-subtract) are unchanged.
static constexpr int P2_ROWS = 256;
static constexpr int P2_THREADS = P2_ROWS;
static constexpr int P2_WAVES = P2_THREADS / 64;
static constexpr int P2_ONEHOP_MAXB = 32;
static constexpr int P2_RB = BASE_W + 16;  // row stride of the wave candidate buffer (the dump runs in groups of 16 slots and may overshoot by 12)
static constexpr size_t P2_LDS_DOUBLES = (size_t)P2_ONEHOP_MAXB * BASE_W + (size_t)P2_WAVES * P2_RB;  // 18.5 KiB
typedef unsigned int pk_v4u __attribute__((ext_vector_type(4)));

struct P2Args {
    double* A;
    size_t lda, rows;
    int j0, w, nblocks;
    int onehop_max;  // one-hop exchange up to this many blocks (<= P2_ONEHOP_MAXB)
    int bstride;     // 1, or 8: only workgroups with blockIdx.x % 8 == 0 take part - all on ONE XCD (workgroup b is
                     // dispatched to XCD b % 8), the exchange then stays inside that XCD's L2 (plain stores)
    unsigned seq0;
    int* xerr;
    pk_u64* xrec;   // [2][PK_MAXB] records, 16 bytes each: |a| bits | fresh, position | row slot << 32 | fresh
    pk_u64* xvals;  // [2][PK_MAXB][BASE_W] values, 16 bytes each: low half | tag << 32, high half | tag << 32 (tag = step + 1)
    int* prow_arr;
    int* ipiv;
    int* info;
    int2* plist;    // this panel's row-move list (PLIST entries)
    pk_u64* dbg;
    int* xcc_out;   // receives the XCD the panel sits on (one-XCD placement) or -1; read by the update stream's persistent dgemm
};

// 16-byte exchange granules: one write-through (sc1) store / one L1-bypassing load; each 8-byte half is self-describing,
// so nothing depends on the two halves landing together.  Records are rewritten by every block at every step, so one
// freshness bit per half (flipping between consecutive uses of a slot) identifies the step.  A VALUE granule is only
// written while its column is still in play (columns >= k), i.e. an even number of slot uses can pass between two
// writes: its halves carry the whole 32-bit step number instead (a bit would accept the previous panel's value).
// `local`: every reader sits on the writer's XCD - the store only has to reach that XCD's L2 (no write-through to memory)
__device__ __forceinline__ void st_granule(pk_u64* p, pk_u64 w0, pk_u64 w1, bool local) {
    const pk_v4u v = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
    if (local) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld_granule(const pk_u64* p, pk_u64& w0, pk_u64& w1) {
    pk_v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    w0 = ((pk_u64)v.y << 32) | v.x;
    w1 = ((pk_u64)v.w << 32) | v.z;
}
// four granules in flight; value = (low | high << 32) when both tags match, else `ok` is cleared
__device__ __forceinline__ void ld_values4(const pk_u64* p0, const pk_u64* p1, const pk_u64* p2, const pk_u64* p3, pk_u64 vtag,
                                           double (&out)[4], bool& ok) {
    pk_v4u v0, v1, v2, v3;
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
    const unsigned tag = (unsigned)(vtag >> 32);
    ok = v0.y == tag && v0.w == tag && v1.y == tag && v1.w == tag && v2.y == tag && v2.w == tag && v3.y == tag && v3.w == tag;
    out[0] = __hiloint2double((int)v0.z, (int)v0.x);
    out[1] = __hiloint2double((int)v1.z, (int)v1.x);
    out[2] = __hiloint2double((int)v2.z, (int)v2.x);
    out[3] = __hiloint2double((int)v3.z, (int)v3.x);
}

struct P2Lds {
    double* cand;    // [P2_ONEHOP_MAXB][BASE_W] candidate rows of the other blocks (slot 0 only beyond 32 blocks)
    double* rowbuf;  // [P2_WAVES][P2_RB] every wave's candidate row
    double* r_key;
    unsigned* r_pos;
    int* r_t;
    int* s_ctl;      // {pivot row, pivot position, skip | error << 1 | cand slot << 8, -}
};
struct P2Ticks {  // developer instrumentation (RMHIP_LU_PANEL_DEBUG=1): 100 MHz wall-clock ticks per phase, block 0 thread 0
    pk_u64 tk, acc[16];
};
#define P2_TICK(i)                                              \
    if (DBG && blockIdx.x == 0 && threadIdx.x == 0) {           \
        const pk_u64 now_ = wall_clock64();                     \
        ticks->acc[i] += now_ - ticks->tk;                      \
        ticks->tk = now_;                                       \
    }

// ---- the column loop is software pipelined: the exchange for column k+1 runs under the bulk of elimination k.
// Once the pivot row of column k is known, every thread updates ONLY column k+1 of its row, the block picks its
// candidate for k+1 and publishes it, and the remaining columns are eliminated while the records travel.  A
// candidate row is parked BEFORE its columns beyond k+1 saw pivot row k; wave 0 applies that update to the parked
// copy (one multiply-subtract per lane, the same two operations the owning thread performs later, so the published
// values are bit-identical to what the row will hold).
//
// p2_select<NS>: pick and publish the block's candidate for column kn (its values sit in window slot NS), then - after
// `rest` ran (the elimination the exchange hides) - collect every block's candidate and leave the winner in
// s_ctl / cand.  FIX: the parked row still needs pivot row kn-1 (factor in its slot NS-1) applied beyond column kn.
template <int NS, bool FIX, bool DBG, class Rest>
__device__ __forceinline__ bool p2_select(const P2Args& g, const P2Lds& L, double (&a)[BASE_W + 4], const int pos, const int jj, const int kn,
                                         const int fix_skip, const double* fix_row, P2Ticks* ticks, Rest rest) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int blk = blockIdx.x / g.bstride;
    const bool local = g.bstride > 1;
    const unsigned seq = g.seq0 + (unsigned)kn;
    const int par = (int)(seq & 1u);
    const pk_u64 fresh = (pk_u64)(((seq >> 1) & 1u) ^ 1u) << 63;
    const pk_u64 topbit = (pk_u64)1 << 63;
    const pk_u64 vtag = (pk_u64)(seq + 1u) << 32;  // value granules: exact step tag in the upper half of both words
    // ---- wave candidate
    double key = 0.0;
    if (pos >= 0) {
        const double av = fabs(a[NS]);
        if (av > 0.0) key = av;  // NaN or zero never wins (host_lu.rs: `abs > pivot_abs`)
    }
    double wk;
    unsigned wp;
    const int wl = wave_argmax_f64(key, (unsigned)pos, &wk, &wp);
    if (lane == wl) {  // the wave's winner parks its row (the live part of the register window)
        double* dst = L.rowbuf + wv * P2_RB + 4 * jj;
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            if (jj < 16 - 4 * G) {  // uniform (an `if`, not a `break`: a loop with an early exit is not unrolled and the window would go to scratch)
#pragma unroll
                for (int u = 0; u < 16; ++u) dst[16 * G + u] = a[16 * G + u];
            }
        }
    }
    if (lane == 0) {
        L.r_key[wv] = wk;
        L.r_pos[wv] = wp;
        L.r_t[wv] = wl < 0 ? 0 : wv * 64 + wl;
    }
    __syncthreads();
    P2_TICK(1)  // wave arg-max + park + barrier
    if (wv == 0) {
        // ---- block candidate = best of the wave records; publish its row and the record (nothing orders the stores:
        // every half word is self-describing and readers retry stale ones)
        const double k4 = lane < P2_WAVES ? L.r_key[lane] : 0.0;
        const unsigned p4 = lane < P2_WAVES ? L.r_pos[lane] : 0xffffffffu;
        double bkd;
        unsigned bp;
        const int bl = quad_argmax_f64(k4, p4, &bkd, &bp);
        const pk_u64 bk = bl < 0 ? 0 : (pk_u64)__double_as_longlong(bkd);
        const int bw = bl < 0 ? 0 : bl;
        const int bt = L.r_t[bw];
        const size_t slot = (size_t)par * PK_MAXB + blk;
        if (lane == 0) st_granule(g.xrec + slot * 2, bk | fresh, (pk_u64)bp | ((pk_u64)(unsigned)bt << 32) | fresh, local);
        if (lane >= kn && lane < g.w) {
            double v = L.rowbuf[bw * P2_RB + lane];
            if (FIX && !fix_skip && lane > kn) {
                const double fw = L.rowbuf[bw * P2_RB + kn - 1];  // the row's multiplier for pivot row kn-1
                const double prod = fw * fix_row[lane];
                v = v - prod;
            }
            const pk_u64 bits = (pk_u64)__double_as_longlong(v);
            st_granule(g.xvals + (slot * BASE_W + lane) * 2, (bits & 0xffffffffull) | vtag, (bits >> 32) | vtag, local);
        }
        P2_TICK(2)  // publish
    }
    rest();
    P2_TICK(6)  // elimination under the exchange
    if (FIX) __syncthreads();  // every wave is done with the previous pivot row in `cand`
    const bool onehop = g.nblocks <= g.onehop_max;
    int bad = 0;
    if (onehop && lane >= kn && lane < g.w) {
        // ---- wave wv fetches the candidate rows of blocks wv, wv + 4, ... (four loads in flight per round)
#pragma unroll
        for (int round = 0; round < P2_ONEHOP_MAXB / (4 * P2_WAVES); ++round) {
            const int b0 = wv + 4 * P2_WAVES * round;
            if (b0 >= g.nblocks) break;  // uniform
            const pk_u64* base = g.xvals + ((size_t)par * PK_MAXB * BASE_W + lane) * 2;
            const pk_u64* ptr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + P2_WAVES * u;
                ptr[u] = base + (size_t)(b < g.nblocks ? b : b0) * BASE_W * 2;
            }
            double vals[4];
            bool ok;
            int spins = 0;
            for (;;) {
                ld_values4(ptr[0], ptr[1], ptr[2], ptr[3], vtag, vals, ok);
                if (ok) break;
                if (++spins > PK_SPIN_LIMIT || __hip_atomic_load(g.xerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    bad = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + P2_WAVES * u;
                if (b < g.nblocks) L.cand[b * BASE_W + lane] = vals[u];
            }
        }
    }
    if (wv == 0) {
        // ---- fold every block's record as it arrives
        double gk = 0.0;
        unsigned gp = 0xffffffffu;
        int gb = -1, gt = 0;
        for (int b = lane; b < g.nblocks; b += 64) {
            pk_u64 wa, wb;
            int spins = 0;
            for (;;) {
                ld_granule(g.xrec + ((size_t)par * PK_MAXB + b) * 2, wa, wb);
                if ((((wa ^ fresh) | (wb ^ fresh)) >> 63) == 0) break;
                if (++spins > PK_SPIN_LIMIT || __hip_atomic_load(g.xerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    bad = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const double ck = __longlong_as_double((long long)(wa & ~topbit));  // |a| >= 0, never NaN (NaN never becomes a key)
            const unsigned cp = (unsigned)wb;
            if (ck > gk || (ck == gk && gk != 0.0 && cp < gp)) {
                gk = ck;
                gp = cp;
                gb = b;
                gt = (int)((wb >> 32) & 0xffffu);
            }
        }
        double mk;
        unsigned mp;
        const int wl2 = wave_argmax_f64(gk, gp, &mk, &mp);
        P2_TICK(3)  // wait + fold
        int grow = -1, sel = 0;
        if (wl2 >= 0) {
            const int wb_ = __builtin_amdgcn_readlane(gb, wl2);
            const int wt_ = __builtin_amdgcn_readlane(gt, wl2);
            grow = g.j0 + wb_ * P2_ROWS + wt_;
            sel = wb_;
            if (!onehop) {
                sel = 0;
                if (lane >= kn && lane < g.w) {
                    const pk_u64* src = g.xvals + (((size_t)par * PK_MAXB + wb_) * BASE_W + lane) * 2;
                    pk_u64 lo, hi;
                    int spins = 0;
                    for (;;) {
                        ld_granule(src, lo, hi);
                        if ((((lo ^ vtag) | (hi ^ vtag)) >> 32) == 0) break;
                        if (++spins > PK_SPIN_LIMIT) {
                            bad = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    L.cand[lane] = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
                }
            }
        }
        if (lane == 0) {
            L.s_ctl[0] = grow;
            L.s_ctl[1] = (int)mp;
            L.s_ctl[2] = (L.s_ctl[2] & 2) | ((mk <= LU_EPS || grow < 0) ? 1 : 0) | (sel << 8);
        }
        P2_TICK(4)  // winner's row (beyond 32 blocks)
    }
    if (__any(bad)) {
        if (lane == 0) {
            __hip_atomic_store(g.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(&L.s_ctl[3], 1);
        }
    }
    __syncthreads();
    P2_TICK(5)  // barrier
    return L.s_ctl[3] == 0;
}

// Rows of pivot-row multiply-subtracts on the register window, slots [FIRST, BASE_W + 4): hand-made pipeline (one wave
// per SIMD: nothing else fills the slots behind a dependent pair, and the scheduler - in register-pressure mode at > 200
// VGPRs - would emit every product right in front of the subtraction that consumes it and keep two LDS reads in
// flight): batches of eight columns, the pivot values fetched two batches ahead, eight independent products, then
// eight subtractions.  Dead slots (columns >= w, the stale tail of the shifted window) meet whatever the pivot buffer
// holds there; they are never stored.
template <int FIRST>
__device__ __forceinline__ void p2_update(double (&a)[BASE_W + 4], const double factor, const double* pr) {
    constexpr int NB = (BASE_W + 4 + 7) / 8;  // 9 batches cover slots 0..71 (a has 68: the last batch is clipped)
    double pb[3][8];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (8 * st + u < BASE_W + 4) pb[st][u] = pr[8 * st + u];
#pragma unroll
    for (int st = 0; st < NB; ++st) {
        if (8 * st + 7 < FIRST) continue;
        if (st + 2 < NB) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (8 * (st + 2) + u < BASE_W + 4) pb[(st + 2) % 3][u] = pr[8 * (st + 2) + u];
        }
        __builtin_amdgcn_sched_barrier(0);
        double prod[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) prod[u] = factor * pb[st % 3][u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (8 * st + u >= FIRST && 8 * st + u < BASE_W + 4) a[8 * st + u] = a[8 * st + u] - prod[u];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one column (k = 4*jj + KK, window slot KK): its pivot is in s_ctl / cand; retire / displace rows, form the
// multipliers, update column k+1, run the selection for k+1 with the bulk elimination under its exchange.
template <int KK, bool DBG>
__device__ __forceinline__ bool p2_column(const P2Args& g, const P2Lds& L, double (&a)[BASE_W + 4], int& pos, int& retk, int& rpiv,
                                         const int jj, const size_t r, P2Ticks* ticks) {
    const int k = 4 * jj + KK;
    const int kabs = g.j0 + k;
    const int prow = L.s_ctl[0], ppos = L.s_ctl[1], flags = L.s_ctl[2];
    const int skip = flags & 1;  // |pivot| <= 1e-12, or an all-zero / NaN-only column: no elimination (host_lu.rs:54-59)
    const double* pr = L.cand + (flags >> 8) * BASE_W + 4 * jj;  // pr[i] pairs with a[i]
    if (prow < 0) {
        // all-zero (or NaN-only) column: pivot_row stays k (host_lu.rs:38); its occupant retires as row k of U
        if (pos == kabs) {
            pos = -1;
            retk = k;
            rpiv = kabs | 0x40000000;  // bit 30: counts as a singular pivot
        }
    } else if ((int)r == prow && pos >= 0) {
        pos = -1;  // retires as row k of U
        retk = k;
        rpiv = ppos | (skip ? 0x40000000 : 0);
    } else if (pos == kabs) {
        pos = ppos;  // the old occupant of position k moves to the pivot's position
    }
    const bool act = pos >= 0;
    double factor = 0.0;
    if (!skip) factor = a[KK] / pr[KK];
    if (act) a[KK] = factor;  // 0.0 when the step is skipped (host_lu.rs:55-57)
    P2_TICK(10)  // bookkeeping + division
    const bool more = k + 1 < g.w;
    auto rest = [&]() {
        if (act && !skip) p2_update<KK + 2>(a, factor, pr);
    };
    if (!more) return true;
    if (act && !skip) {
        const double prod = factor * pr[KK + 1];
        a[KK + 1] = a[KK + 1] - prod;
    }
    return p2_select<KK + 1, true, DBG>(g, L, a, pos, jj, k + 1, skip, L.cand + (flags >> 8) * BASE_W, ticks, rest);
}

template <bool DBG>
__global__ void __launch_bounds__(P2_THREADS) k_lu_panel2(const P2Args g) {
    extern __shared__ double p2_lds[];
    __shared__ double r_key[P2_WAVES];
    __shared__ unsigned r_pos[P2_WAVES];
    __shared__ int r_t[P2_WAVES];
    __shared__ int s_ctl[8];
    __shared__ P2Ticks s_ticks;  // ticks live in LDS
    P2Lds L;
    L.cand = p2_lds;
    L.rowbuf = L.cand + (size_t)P2_ONEHOP_MAXB * BASE_W;
    L.r_key = r_key;
    L.r_pos = r_pos;
    L.r_t = r_t;
    L.s_ctl = s_ctl;
    P2Ticks* ticks = &s_ticks;
    const bool dbg_on = DBG && blockIdx.x == 0 && threadIdx.x == 0;
    if (dbg_on) {
        for (int i = 0; i < 16; ++i) s_ticks.acc[i] = 0;
        s_ticks.tk = wall_clock64();
    }
    if (blockIdx.x % g.bstride) return;  // one-XCD placement: the other seven XCDs' workgroups are placeholders
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.xcc_out) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        *g.xcc_out = g.bstride > 1 ? (int)(xcc & 0xf) : -1;
    }
    const int t = threadIdx.x;  // row slot
    const size_t r = (size_t)g.j0 + (size_t)(blockIdx.x / g.bstride) * P2_ROWS + t;
    const bool in_rows = r < g.rows;
    int pos = in_rows ? (int)r : -1, retk = -1, rpiv = 0;
    double a[BASE_W + 4];  // register window: a[i] = column 4*jj + i (four spare slots: column k+1 of the last step of a group)
#pragma unroll
    for (int c = 0; c < BASE_W; ++c) a[c] = (in_rows && c < g.w) ? g.A[r + (size_t)(g.j0 + c) * g.lda] : 0.0;
#pragma unroll
    for (int c = BASE_W; c < BASE_W + 4; ++c) a[c] = 0.0;
    if (t < 8) s_ctl[t] = 0;
    for (int i = t; i < P2_ONEHOP_MAXB * BASE_W + P2_WAVES * P2_RB; i += P2_THREADS) L.cand[i] = 0.0;  // cand and rowbuf are contiguous
    __syncthreads();
    P2_TICK(0)  // load
    // pivot of column 0: nothing to hide, nothing to fix up
    if (!p2_select<0, false, DBG>(g, L, a, pos, 0, 0, 1, L.cand, ticks, [] {})) return;
    const int ngroups = (g.w + 3) >> 2;
    for (int jj = 0; jj < ngroups; ++jj) {
        if (!p2_column<0, DBG>(g, L, a, pos, retk, rpiv, jj, r, ticks)) return;
        if (4 * jj + 1 < g.w && !p2_column<1, DBG>(g, L, a, pos, retk, rpiv, jj, r, ticks)) return;
        if (4 * jj + 2 < g.w && !p2_column<2, DBG>(g, L, a, pos, retk, rpiv, jj, r, ticks)) return;
        if (4 * jj + 3 < g.w && !p2_column<3, DBG>(g, L, a, pos, retk, rpiv, jj, r, ticks)) return;
        // columns 4jj .. 4jj+3 are final (multipliers, or the U values of a retired row): store them where the row was
        // loaded from - nobody reads the panel's columns before the kernel ends - and shift the register window by four
        if (in_rows) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * jj + i < g.w) g.A[r + (size_t)(g.j0 + 4 * jj + i) * g.lda] = a[i];
        }
#pragma unroll
        for (int i = 0; i < BASE_W; ++i) a[i] = a[i + 4];
    }
    P2_TICK(7)
    if (in_rows) {
        // the interchange record of the step this row retired at (LAPACK-style target position), its row move (applied to
        // the panel's own columns and to every other column by k_laswp_lists), and the singular-pivot count
        if (retk >= 0) {
            const int kabs = g.j0 + retk;
            g.ipiv[kabs] = rpiv & 0x3fffffff;
            g.prow_arr[kabs] = (int)r;
            g.plist[retk] = (int)r != kabs ? make_int2(kabs, (int)r) : make_int2(-1, -1);
            if (rpiv & 0x40000000) atomicAdd(g.info, 1);
        }
    }
    // ---- slots BASE_W + i of the row-move list: top rows that were displaced instead of retired
    if (blockIdx.x == 0 && t < BASE_W) {
        int2 e = make_int2(-1, -1);
        if (t < g.w && pos >= 0 && pos != (int)r) e = make_int2(pos, (int)r);
        g.plist[BASE_W + t] = e;
        if (t >= g.w) g.plist[t] = make_int2(-1, -1);
    }
    if (dbg_on) {
        s_ticks.acc[8] += wall_clock64() - s_ticks.tk;
        for (int i = 0; i < 16; ++i) g.dbg[i] += s_ticks.acc[i];
    }
}
#undef P2_TICK

// ---- solve path (`mldivide` / `linsolve` / `mrdivide`): panel with pivoting RESTRICTED to the top block ----------------------------
// The reference's contract for a solve is the solution to a residual tolerance (mldivide.rs:380-404, tests :662-696); the pivot
// sequence never leaves the provider there (it does for `lu`, which keeps k_lu_panel2's grid-wide first-maximum rule).  What the
// grid-wide rule costs is a grid of workgroups that meet once per column: 2.97-3.6 us x n columns, 59 of 98 ms at n = 16384, and
// 32-64 CUs held while they wait.  On the solve path a base panel is therefore factored like the LU step of the hybrid LU-QR
// algorithm (Faverge, Herrmann, Langou, Lowery, Robert, Dongarra, "Designing LU-QR hybrid solvers for performance and stability",
// IPDPS 2014): partial pivoting inside the DIAGONAL DOMAIN - the top RT_ROWS = 256 rows of the panel, ONE workgroup (k_rp_top), no
// exchange - and every row below becomes l = a U11^-1 by substitution (k_rp_below, all rows in parallel), with their "Max
// criterion" as the guard: the factorisation is accepted only if every multiplier satisfies |l| <= tau (threshold partial
// pivoting; tau = 8 by default - the sparse direct solvers' usual threshold u = 0.1 is tau = 10).  The largest |l| is accumulated
// in *growth; a violation, a pivot at the singular cut-off or a NaN makes lu_factor_device report RMHIP_LU_GROWTH and the caller
// refactors a fresh copy with the grid-wide rule (telemetry.solve_fallbacks "lu:pivot_growth", rmhip_lu_stats).  For matrices
// whose entries are not arranged to defeat it (random dense, diagonally dominant, SPD Gram matrices) the 256 candidates hold an
// entry within a small factor of the column maximum: max|l| 1.8-2.5 on U(-1,1) matrices up to n = 16384, residuals equal to
// partial pivoting's (tests/test_gpu_solvepath.py); a row-permuted diagonally dominant matrix fails at its first panel and costs
// one small read before the grid-wide rule takes over.
//
// Both kernels keep one matrix row per thread in a 64-register window and work in MICRO-PANELS of eight columns: inside a
// micro-panel only the eight window columns are eliminated step by step (the latency chain), everything to the right receives
// the eight pivots at once - eight chained FMAs per element with the operands fetched from LDS in bulk, instead of 64 sweeps that
// each wait for their own operands.  The loop over micro-panels is rolled and the window shifts by eight (static register
// indices; straight-line code for 64 steps would be executed once per launch from cold instruction caches).
static constexpr int RT_ROWS = 256;  // rows of the top block = threads of k_rp_top
static constexpr int RT_VS = 10;     // doubles per 8-value LDS record: 80 bytes put the 16-byte accesses of consecutive records on distinct banks
static constexpr int RT_TRAIL = BASE_W - 8;

// workgroup barrier that does not wait for this wave's global stores (on gfx9 `__syncthreads` waits for vmcnt too)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a[8 + c] += sum_i ng[i] * T[i * ldt + c] over the live trailing slots c of micro-panel m.  Per batch of eight columns the eight
// pivots are applied one after the other to eight INDEPENDENT accumulators (a column's eight FMAs are a dependent chain: done column
// by column, as first written, every column paid the LDS latency and the chain latency - 145 cycles per column, measured), and the
// operands of pivot i + 1 are fetched before the FMAs of pivot i.
__device__ __forceinline__ void rank8_update(double (&a)[BASE_W], const double (&ng)[8], const double* T, const int ldt, const int m) {
#pragma unroll
    for (int bch = 0; bch < RT_TRAIL / 8; ++bch) {
        if (m + 1 + bch < BASE_W / 8) {  // uniform: window slots beyond column 63 are dead
            double u[2][8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[0][q] = T[8 * bch + q];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i + 1 < 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) u[(i + 1) & 1][q] = T[(i + 1) * ldt + 8 * bch + q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) a[8 + 8 * bch + q] = __builtin_fma(ng[i], u[i & 1][q], a[8 + 8 * bch + q]);
            }
        }
    }
}

struct RtArgs {
    double* A;
    size_t lda, rows;  // rows: one past the last row of the top block (<= j0 + RT_ROWS)
    int j0, w;
    int* prow_arr;
    int* ipiv;
    int* info;
    int2* plist;    // this panel's row-move list (PLIST entries, the layout k_lu_panel2 writes)
    double* ucomp;  // [BASE_W][BASE_W] compact copy of the pivot rows for k_rp_below: ucomp[k * BASE_W + c] = U[k][c], 1 / u_kk on the diagonal
    int* xcc_out;   // receives the XCD this workgroup runs on (read by the update stream's persistent kernels), or nullptr
    double* linv;   // receives the inverses of L11's four unit-lower 16 x 16 diagonal blocks (4 x 256 doubles, [k][i]), or nullptr
    int uinv_on;    // also leave the inverses of U11's 16 x 16 diagonal blocks behind the compact copy (k_rp_below_mfma)
    unsigned* yield_word;  // two-level driver: receives this workgroup's CU key while it runs, 0 when it is done (update blocks on that CU pause), or nullptr
};
struct RtLds {
    unsigned* wmax; // [2][4] every wave's best key (high word of |a|, low byte = 255 - thread), by step parity
    double* vals;   // [2][4][RT_VS] the wave candidate's window values (+ its reciprocal in slot 8, its current position in the low word of slot 9)
    double* pw;     // [8][RT_VS] pivot j of the micro-panel: its window values (multipliers w.r.t. pivots < j, then u_jj ...)
    double* pt;     // [8][RT_TRAIL] pt[j][c] = pivot j's value in trailing slot c, as it was when the micro-panel began
};

struct RtTicks {  // developer instrumentation (RMHIP_LU_PANEL_DEBUG=1): 100 MHz ticks per phase, thread 0
    pk_u64 tk, acc[12];
};
#define RT_TICK(i)                              \
    if (DBG && threadIdx.x == 0) {              \
        const pk_u64 now_ = wall_clock64();     \
        ticks.acc[i] += now_ - ticks.tk;        \
        ticks.tk = now_;                        \
    }
// One column (k = 8 m + J, window slot J).  The pivot search runs on 32-bit keys: the high word of |a_k| with its low byte replaced
// by 255 - thread, so the maximum is ONE v_max_u32 per DPP step (a 64-bit arg-max is three instructions and two reductions),
// names its owner, and picks the largest candidate up to 2^-12 relative, ties to the lowest row - the solve path need not
// reproduce host_lu.rs's first-maximum rule, only keep the multipliers small (they stay below 1 + 2^-12 inside the block).
// Each wave's candidate leaves its window values, its reciprocal and its position in LDS; one barrier; every thread folds the four
// wave records itself.  (A first version let all 256 rows issue one LDS atomic max on the same word: 2.8 us per column - same-address
// LDS atomics serialise at ~25 cycles each.)
template <int J, bool DBG>
__device__ __forceinline__ void rt_step(const RtArgs& g, const RtLds& L, const int m, double (&a)[BASE_W], int& pos, int& retk, int& rpiv,
                                        double& myrinv, RtTicks& ticks) {
    const int k = 8 * m + J;
    if (k >= g.w) return;  // uniform
    const int kabs = g.j0 + k, par = J & 1, t = threadIdx.x, wv = t >> 6;
    unsigned key = 0;
    if (pos >= 0) {
        const double av = fabs(a[J]);
        if (av > 0.0) key = ((unsigned)__double2hiint(av) & ~0xffu) | (unsigned)(255 - t);  // NaN or zero never wins
    }
    // 1 / a[J]: hardware estimate + two Newton steps (the error squares twice: full precision from any estimate good to 2^-14); a
    // division is three times the instructions on this chain.  Zero / tiny / non-finite candidates give garbage nobody multiplies by
    // (such a pivot is skipped).
    double rinv = __builtin_amdgcn_rcp(a[J]);
    rinv = __builtin_fma(rinv, __builtin_fma(-a[J], rinv, 1.0), rinv);
    rinv = __builtin_fma(rinv, __builtin_fma(-a[J], rinv, 1.0), rinv);
    const unsigned wm = wave_max_u32(key);
    if (key != 0 && key == wm) {  // exactly one lane: the keys carry the thread number
        double* mine = L.vals + (par * 4 + wv) * RT_VS;
#pragma unroll
        for (int i = J; i < 8; ++i) mine[i] = a[i];
        mine[8] = rinv;
        mine[9] = __hiloint2double(0, pos);  // its position rides in the record's spare slot: one LDS round trip for the readers
    }
    if ((t & 63) == 0) L.wmax[par * 4 + wv] = wm;
    RT_TICK(1)  // candidate: wave maximum, reciprocal, the wave winner's record
    lds_barrier();
    RT_TICK(2)  // barrier
    const uint4 w4 = *reinterpret_cast<const uint4*>(L.wmax + par * 4);
    const unsigned b01 = w4.x > w4.y ? w4.x : w4.y, b23 = w4.z > w4.w ? w4.z : w4.w;
    const unsigned best = b01 > b23 ? b01 : b23;
    const bool none = best == 0;  // all-zero / NaN-only column: the row at position k retires (host_lu.rs:38)
    const int btid = 255 - (int)(best & 0xffu), bwv = btid >> 6;
    const double* pvp = L.vals + (par * 4 + bwv) * RT_VS;
    // the whole record of the winning wave in ONE LDS round trip, before any of the decisions below (read on demand - the pivot for
    // the singular test, then the position, then the window values inside the branch that uses them - they were three dependent
    // round trips of ~120 cycles on the per-column chain).  With no candidate at all the record is stale: nothing below uses it then.
    double pv[10];
#pragma unroll
    for (int i = J; i < 10; ++i) pv[i] = pvp[i];
    const int bpos = __double2loint(pv[9]);
    const bool skip = none || !(fabs(pv[J]) > LU_EPS);  // counted as a singular pivot; the solve path then refactors with the grid-wide rule
    if (none) {
        if (pos == kabs) {
            pos = -1;
            retk = k;
            rpiv = kabs | 0x40000000;
        }
    } else {
        if (t == btid) {
            pos = -1;  // retires as row k of U
            retk = k;
            rpiv = bpos | (skip ? 0x40000000 : 0);
            myrinv = rinv;
        } else if (pos == kabs) {
            pos = bpos;  // the old occupant of position k moves to the pivot's position
        }
    }
    if (pos >= 0) {
        if (skip) {
            a[J] = 0.0;  // host_lu.rs:55-57
        } else {
            const double f = a[J] * pv[8];
            a[J] = f;
#pragma unroll
            for (int i = J + 1; i < 8; ++i) a[i] = __builtin_fma(-f, pv[i], a[i]);
        }
    }
    RT_TICK(3)  // winner, bookkeeping, window elimination
}
template <int J, bool DBG>
__device__ __forceinline__ void rt_steps(const RtArgs& g, const RtLds& L, const int m, double (&a)[BASE_W], int& pos, int& retk, int& rpiv,
                                         double& myrinv, RtTicks& ticks) {
    if constexpr (J < 8) {
        rt_step<J, DBG>(g, L, m, a, pos, retk, rpiv, myrinv, ticks);
        rt_steps<J + 1, DBG>(g, L, m, a, pos, retk, rpiv, myrinv, ticks);
    }
}

template <bool DBG>
__global__ void __launch_bounds__(RT_ROWS) k_rp_top(const RtArgs g, pk_u64* dbg) {
    __shared__ __attribute__((aligned(16))) unsigned s_wmax[2 * 4];
    __shared__ __attribute__((aligned(16))) double s_vals[2 * 4 * RT_VS];
    __shared__ __attribute__((aligned(16))) double s_pw[8 * RT_VS];
    __shared__ __attribute__((aligned(16))) double s_pt[8 * RT_TRAIL];
    __shared__ int s_prow[BASE_W];  // physical row of pivot k (for the inverses of L11's diagonal blocks at the end)
    extern __shared__ double rt_pad[];  // only asked for: keeps update-stream dgemm blocks off this CU (getrf_rec)
    RtLds L;
    L.wmax = s_wmax;
    L.vals = s_vals;
    L.pw = s_pw;
    L.pt = s_pt;
    const int t = threadIdx.x;
    chain_prio();
    RtTicks ticks;
    if (DBG) {
        for (int i = 0; i < 12; ++i) ticks.acc[i] = 0;
        ticks.tk = wall_clock64();
    }
    if (t == 0 && g.xcc_out) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        *g.xcc_out = (int)(xcc & 0xf);
    }
    // Cooperative yield (two-level driver).  This workgroup's fp64 FMAs queue behind the matrix-core instructions of any dgemm wave on
    // the same SIMD - 57-63 us for a 64-column top block on a CU of its own, 120-250 us beside a block of the deep update (rocprofv3
    // timeline, round 5) - and it is a quarter of the critical chain.  It names its CU in a device word; the update streams' eight-wave
    // blocks read that word once per k tile and the one that finds its own CU there sleeps until the word changes (dgemm.hip, w8_tile).
    const CuAnnounce on_cu(g.yield_word);
    const size_t r = (size_t)g.j0 + t;
    const bool in_rows = r < g.rows;
    int pos = in_rows ? (int)r : -1, retk = -1, rpiv = 0;
    double a[BASE_W];  // register window: a[i] = column 8 m + i
#pragma unroll
    for (int c = 0; c < BASE_W; ++c) a[c] = (in_rows && c < g.w) ? g.A[r + (size_t)(g.j0 + c) * g.lda] : 0.0;
    double myrinv = 0.0;  // a pivot row keeps the reciprocal of its pivot: k_rp_below multiplies by it
    RT_TICK(0)  // load
    const int nmp = (g.w + 7) >> 3;
#pragma unroll 1
    for (int m = 0; m < nmp; ++m) {
        rt_steps<0, DBG>(g, L, m, a, pos, retk, rpiv, myrinv, ticks);
        // ---- the micro-panel's pivot rows park their window and their (not yet updated) trailing values
        if (retk >= 8 * m) {
            const int jr = retk - 8 * m;
#pragma unroll
            for (int i = 0; i < 8; ++i) s_pw[jr * RT_VS + i] = a[i];
#pragma unroll
            for (int bch = 0; bch < RT_TRAIL / 8; ++bch) {
                if (m + 1 + bch < BASE_W / 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) s_pt[jr * RT_TRAIL + 8 * bch + q] = a[8 + 8 * bch + q];
                }
            }
        }
        if (t < 8 && 8 * m + t >= g.w) {  // steps beyond a ragged panel's width have no pivot: their records must not hold an earlier micro-panel's
#pragma unroll
            for (int i = 0; i < 8; ++i) s_pw[t * RT_VS + i] = 0.0;
        }
        RT_TICK(4)  // park
        lds_barrier();
        RT_TICK(5)
        // ---- multipliers this row holds for pivots i of the micro-panel: valid while it was active at step i.  The trailing rows
        // in s_pt are RAW (pivot j's row still lacks pivots i < j of this micro-panel), so instead of fixing them up in a serial
        // pass the multipliers are transformed: a -= f U = f (L8^-1 R) = (f L8^-1) R with L8 the unit-lower triangle in s_pw;
        // g L8 = f is 28 operations per row.  A pivot row takes part with the multipliers it collected before it retired, which
        // is exactly its own row of L8^-1 R - its final U values.
        double gm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gm[i] = (retk < 0 || 8 * m + i < retk) ? a[i] : 0.0;
#pragma unroll
        for (int i = 6; i >= 0; --i) {
#pragma unroll
            for (int j = i + 1; j < 8; ++j) gm[i] = __builtin_fma(-gm[j], s_pw[j * RT_VS + i], gm[i]);
        }
        double ng[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ng[i] = -gm[i];
        RT_TICK(6)  // multipliers -> g
        rank8_update(a, ng, s_pt, RT_TRAIL, m);
        RT_TICK(7)  // rank-8 update
        // ---- columns 8 m .. 8 m + 7 are final: multipliers, or the U values of a retired row.  They go where the row was loaded from
        // (the interchange is a k_laswp_lists call, as for k_lu_panel2); pivot rows also leave the transposed copy for k_rp_below.
        if (in_rows) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * m + i < g.w) g.A[r + (size_t)(g.j0 + 8 * m + i) * g.lda] = a[i];
            if (retk >= 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (8 * m + i < g.w) g.ucomp[retk * BASE_W + 8 * m + i] = (8 * m + i == retk) ? myrinv : a[i];  // diagonal: 1 / u_kk
            }
        }
#pragma unroll
        for (int i = 0; i < BASE_W - 8; ++i) a[i] = a[i + 8];
#pragma unroll
        for (int i = BASE_W - 8; i < BASE_W; ++i) a[i] = 0.0;
        RT_TICK(8)  // store + shift
    }
    if (DBG && t == 0)
        for (int i = 0; i < 12; ++i) dbg[i] += ticks.acc[i];
    if (in_rows && retk >= 0) {
        // interchange record of the step this row retired at, its row move, and the singular-pivot count (as k_lu_panel2)
        const int kabs = g.j0 + retk;
        g.ipiv[kabs] = rpiv & 0x3fffffff;
        g.prow_arr[kabs] = (int)r;
        g.plist[retk] = (int)r != kabs ? make_int2(kabs, (int)r) : make_int2(-1, -1);
        if (rpiv & 0x40000000) atomicAdd(g.info, 1);
    }
    if (t < BASE_W) {  // top rows that were displaced instead of retired
        int2 e = make_int2(-1, -1);
        if (t < g.w && pos >= 0 && pos != (int)r) e = make_int2(pos, (int)r);
        g.plist[BASE_W + t] = e;
        if (t >= g.w) g.plist[t] = make_int2(-1, -1);
    }
    if (g.w == BASE_W && g.uinv_on && in_rows && retk >= 0) s_prow[retk] = (int)r;
    if (g.w == BASE_W && g.uinv_on) {
        // Inverses of U11's four 16 x 16 diagonal blocks for k_rp_below_mfma (the rows below then need matrix-core products only).
        // Thread (b, i) solves U_bb x = e_i by back substitution - every index static, x[m] = 0 beyond i - and writes column i.
        __syncthreads();  // the pivot rows' stores to ucomp (other threads of this workgroup) are visible
        if (t < 64) {
            const int b = t >> 4, i = t & 15;
            const double* ub = g.ucomp + (size_t)(16 * b) * BASE_W + 16 * b;  // ub[k * BASE_W + m] = U_bb[k][m], 1 / u_kk on the diagonal
            double x[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) x[m] = (m == i) ? ub[m * BASE_W + m] : 0.0;
#pragma unroll
            for (int k = 14; k >= 0; --k) {
                double sum = 0.0;
#pragma unroll
                for (int m = k + 1; m < 16; ++m) sum = __builtin_fma(ub[k * BASE_W + m], x[m], sum);
                if (k < i) x[k] = -sum * ub[k * BASE_W + k];
            }
            double* dst = g.ucomp + (size_t)BASE_W * BASE_W + 256 * b + i;
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[16 * k] = x[k];
        } else if (t < 128 && g.linv) {
            // the same for L11's unit-lower diagonal blocks (k_trsm_lower_mfma): thread (b, k) solves L_bb x = e_k forwards and writes
            // column k as 16 contiguous values ([k][i] storage).  The multipliers sit below the diagonal of the compact copy.
            // (the compact copy only holds a pivot row's multipliers inside its own micro-panel; the full rows are in A, where each
            // pivot row was loaded from - s_prow - and were stored by other threads of this workgroup: read past the L1)
            const int b = (t - 64) >> 4, k = (t - 64) & 15;
            double x[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) x[m] = (m == k) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                const unsigned long long* lrow = reinterpret_cast<const unsigned long long*>(g.A + (size_t)s_prow[16 * b + i] + (size_t)(g.j0 + 16 * b) * g.lda);
                double sum = 0.0;
#pragma unroll
                for (int m = 0; m < i; ++m) {
                    const double lim = __longlong_as_double((long long)__hip_atomic_load(lrow + (size_t)m * g.lda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    sum = __builtin_fma(lim, x[m], sum);
                }
                if (i > k) x[i] = -sum;
            }
            double* dst = g.linv + (size_t)256 * b + 16 * k;
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i] = x[i];
        }
    }
    on_cu.done();
}

// Rows below the top block on the matrix cores (round 5).  Beside an update block every fp64 VALU instruction of a chain kernel waits
// for the matrix-core instruction in flight on its SIMD - the two share the fp64 datapath (scripts/micro/chain_contention.hip: 2048
// dependent FMAs 12.8 us alone, 113 us beside MFMA waves even at wave priority 3, integer work unaffected) - so what a chain kernel
// pays for is its NUMBER of fp64 instructions: 2080 v_fma_f64 per row-wave in k_rp_below, 60-130 us in the solve against 20 alone.
// Here l = a U11^-1 is a blocked substitution over U11's 16 x 16 blocks with the diagonal blocks inverted (by k_rp_top, above):
//   for b = 0..3:  X_b = A_b inv(U_bb);   for c > b:  A_c -= X_b U_bc
// 80 v_mfma_f64_16x16x4 per 32-row wave and nothing else in fp64.  Roles are transposed (D[i][j]: i = panel column, j = matrix row) so
// that the accumulator layout of one product (register r of lane (lq, l15) = column 4 r + lq, row l15) IS the B operand of the next
// (k = 4 s + lq at step s = r): no shuffles, no LDS round trip, the A operands (inverses and negated U blocks, row-major [k][i]) come
// from LDS with conflict-free 64-lane reads.  Rounding differs from the substitution's (products with an inverted block: error
// ~ cond(U_bb) eps per block instead of eps per step); the multiplier bound is checked exactly as before.
static constexpr int RBM_JT = 2;  // 16-row tiles per wave (32 rows: 64 accumulator registers - the wave must fit beside an update block's two waves per SIMD)
__global__ void __launch_bounds__(64) k_rp_below_mfma(double* __restrict__ A, const size_t lda, const size_t rows, const size_t r0, const int j0,
                                                      const double* __restrict__ ut, pk_u64* __restrict__ growth, unsigned* yield_tab) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double Ub[10 * 256];  // block id(b, c), b <= c: diagonal = inv(U_bb), off-diagonal = -U_bc; [k][i] row-major
    const int t = threadIdx.x, l15 = t & 15, lq = t >> 4;
    chain_prio();
    const CuAnnounce on_cu(yield_tab);
    {
        constexpr int BB[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, CC[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
#pragma unroll
        for (int e0 = 0; e0 < 40; e0 += 8) {  // eight loads in flight, then their LDS writes
            double stage[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q, blk = e >> 2, within = (e & 3) * 64 + t, kk = within >> 4, ii = within & 15;
                stage[q] = BB[blk] == CC[blk] ? ut[BASE_W * BASE_W + 256 * BB[blk] + within] : -ut[(16 * BB[blk] + kk) * BASE_W + 16 * CC[blk] + ii];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) Ub[(e0 + q) * 64 + t] = stage[q];
        }
    }
    const size_t rbase = r0 + (size_t)blockIdx.x * (16 * RBM_JT);
    v4d acc[4][RBM_JT];  // acc[b][jt][r] = A[rbase + 16 jt + l15][j0 + 16 b + 4 r + lq]
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int jt = 0; jt < RBM_JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t row = rbase + 16 * jt + l15;
                acc[b][jt][r] = row < rows ? A[row + (size_t)(j0 + 16 * b + 4 * r + lq) * lda] : 0.0;
            }
    lds_barrier();
    pk_u64 mx = 0;
    constexpr int DIAG[4] = {0, 4, 7, 9};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double aop[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) aop[s4] = Ub[DIAG[b] * 256 + (4 * s4 + lq) * 16 + l15];
        v4d x[RBM_JT];
#pragma unroll
        for (int jt = 0; jt < RBM_JT; ++jt) {
            x[jt] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) x[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s4], acc[b][jt][s4], x[jt], 0, 0, 0);
        }
#pragma unroll
        for (int c = b + 1; c < 4; ++c) {
            double uop[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) uop[s4] = Ub[(DIAG[b] + (c - b)) * 256 + (4 * s4 + lq) * 16 + l15];
#pragma unroll
            for (int jt = 0; jt < RBM_JT; ++jt)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc[c][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(uop[s4], x[jt][s4], acc[c][jt], 0, 0, 0);
        }
#pragma unroll
        for (int jt = 0; jt < RBM_JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t row = rbase + 16 * jt + l15;
                const double v = x[jt][r];
                if (row < rows) A[row + (size_t)(j0 + 16 * b + 4 * r + lq) * lda] = v;
                const pk_u64 bits = (pk_u64)__double_as_longlong(v) & 0x7fffffffffffffffull;  // |v| as an integer: a NaN sorts above every number
                mx = bits > mx ? bits : mx;
            }
    }
    mx = wave_max_u64(mx);
    if (t == 0 && mx != 0) atomicMax(growth, mx > 0x7ff0000000000000ull ? 0x7ff8000000000000ull : mx);
    on_cu.done();
}

// Rows below the top block: l = a U11^-1, one thread per row, U (transposed, as k_rp_top left it) staged in LDS.  One wave per
// workgroup: every FMA takes an 8-byte operand from LDS, so a CU's LDS feeds about one wave at the rate its SIMD multiplies, and the
// rows should spread over as many CUs as there are.
template <int RB_THREADS, bool DBG>
__global__ void __launch_bounds__(RB_THREADS) k_rp_below(double* __restrict__ A, const size_t lda, const size_t rows, const size_t r0,
                                                         const int j0, const int w, const double* __restrict__ ut,
                                                         pk_u64* __restrict__ growth, const double /*tau: the host compares*/, pk_u64* dbg) {
    __shared__ __attribute__((aligned(16))) double U[BASE_W * BASE_W + BASE_W];  // U[k * BASE_W + c] = U11[k][c], 1 / u_kk on the diagonal (+ slack for dead slots)
    const int t = threadIdx.x;
    chain_prio();
    RtTicks ticks;
    if (DBG) {
        for (int i = 0; i < 12; ++i) ticks.acc[i] = 0;
        ticks.tk = wall_clock64();
    }
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* src = reinterpret_cast<const d2*>(ut);
        d2* dst = reinterpret_cast<d2*>(U);
        constexpr int PER = BASE_W * BASE_W / 2 / RB_THREADS;
        d2 stage[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) stage[u] = src[u * RB_THREADS + t];
#pragma unroll
        for (int u = 0; u < PER; ++u) dst[u * RB_THREADS + t] = stage[u];
        if (t < BASE_W) U[BASE_W * BASE_W + t] = 0.0;
    }
    RT_TICK(0)  // stage U
    const size_t r = r0 + (size_t)blockIdx.x * RB_THREADS + t;
    const bool live = r < rows;
    double a[BASE_W];
#pragma unroll
    for (int c = 0; c < BASE_W; ++c) a[c] = (live && c < w) ? A[r + (size_t)(j0 + c) * lda] : 0.0;
    lds_barrier();
    if (DBG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RT_TICK(1)  // load the rows
    double mx = 0.0;
    int bad = 0;
#pragma unroll 1
    for (int m = 0; m < BASE_W / 8; ++m) {
        const double* ud = U + (8 * m) * BASE_W + 8 * m;  // ud[kk * BASE_W + i] = U11[8 m + kk][8 m + i]
        // the 8 x 8 triangle's 36 operands first, so that the eight dependent steps below wait for nothing
        double tri[8][8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int i = kk; i < 8; ++i) tri[kk][i] = ud[kk * BASE_W + i];
        }
        double nx[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            // columns beyond w hold zeros here and whatever an earlier panel left in ut: their multiplier is 0 by definition
            const double x = (8 * m + kk < w) ? a[kk] * tri[kk][kk] : 0.0;  // the diagonal holds 1 / u_kk
            a[kk] = x;
            const double ax = fabs(x);
            bad |= (ax != ax);  // NaN: fmax below would drop it
            mx = fmax(mx, ax);
            nx[kk] = -x;
#pragma unroll
            for (int i = kk + 1; i < 8; ++i) a[i] = __builtin_fma(nx[kk], tri[kk][i], a[i]);
        }
        RT_TICK(2)  // 8 x 8 triangle
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * m + i < w) A[r + (size_t)(j0 + 8 * m + i) * lda] = a[i];
        }
        RT_TICK(3)  // stores
        rank8_update(a, nx, ud + 8, BASE_W, m);
        RT_TICK(4)  // rank-8 update
#pragma unroll
        for (int i = 0; i < BASE_W - 8; ++i) a[i] = a[i + 8];
#pragma unroll
        for (int i = BASE_W - 8; i < BASE_W; ++i) a[i] = 0.0;
    }
    if (!live) {
        mx = 0.0;
        bad = 0;
    }
    pk_u64 bits = bad ? 0x7ff8000000000000ull : (pk_u64)__double_as_longlong(mx);
    bits = wave_max_u64(bits);
    if ((t & 63) == 0 && bits != 0) atomicMax(growth, bits);
    if (DBG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RT_TICK(5)  // tail
    if (DBG && t == 0 && blockIdx.x == 0)
        for (int i = 0; i < 6; ++i) dbg[9 + i] += ticks.acc[i];
}

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
// counter != nullptr: the column groups are handed out by a counter and workgroups on XCD *avoid_xcc leave at once (the
// update stream in the LU's late phase, see k_dgemm_w8p) - the grid still has one workgroup per group.
__global__ void __launch_bounds__(2 * PLIST) k_laswp_lists(double* __restrict__ A, size_t lda, size_t c0, size_t c1,
                                                           const int2* __restrict__ lists, int p0, int p1, unsigned* counter,
                                                           const int* avoid_xcc) {
    __shared__ unsigned s_group;
    __shared__ int s_avoid;
    const int i = threadIdx.x & (PLIST - 1);
    const int half = threadIdx.x / PLIST;  // 0 or 1
    unsigned group = blockIdx.x;
    if (!counter) chain_prio();
    if (counter) {
        if (avoid_xcc && gridDim.x >= 16) {  // (a smaller grid may sit on that XCD entirely)
            // one read per workgroup (the panel kernel may be writing it right now: see k_dgemm_w8p)
            if (threadIdx.x == 0) s_avoid = __hip_atomic_load(avoid_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            if ((int)(xcc & 0xf) == s_avoid) return;
        }
    }
  for (;;) {
    if (counter) {
        if (threadIdx.x == 0) s_group = atomicAdd(counter, 1u);
        __syncthreads();
        group = s_group;
        if (group >= gridDim.x) return;
    }
    const size_t cbase = c0 + (size_t)group * LASWP_COLS;
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
    if (!counter) return;
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
static constexpr int TRSM_SW = TRSM_W + 1;  // LDS row stride (doubles) of a 65..128-wide solve; 65 for <= 64

// value of lane `lane` (wave-uniform index) in every lane: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// TRSM_NC = right-hand sides a wave solves together: 4 when there are more columns than waves on the
// chip (independent dependency chains interleave), 1 otherwise (dead column slots would cost issue cycles).
// TRSM_THREADS: 256 (one wave per SIMD: the substitution is VALU-issue bound) for few columns, 512 for many.
template <int MODE, int TRSM_NC, int TRSM_THREADS>
__global__ void __launch_bounds__(TRSM_THREADS) k_trsm_fused(const double* __restrict__ T, size_t ldt, int w,
                                                             double* __restrict__ B, size_t ldb, size_t ncols, unsigned* counter,
                                                             const int* avoid_xcc) {
    extern __shared__ double Ts[];  // [w][sw]
    if (!counter) chain_prio();
    // counter != nullptr (update stream of the LU's late phase, see k_dgemm_w8p): column groups are handed out by a counter, wave
    // by wave, and workgroups on XCD *avoid_xcc leave before they stage anything
    if (counter && avoid_xcc && gridDim.x >= 16) {
        __shared__ int s_avoid;  // one read per workgroup (see k_dgemm_w8p)
        if (threadIdx.x == 0) s_avoid = __hip_atomic_load(avoid_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((int)(xcc & 0xf) == s_avoid) return;
    }
    const int wr = w > 64 ? TRSM_W : 64;  // staged rows per column (64-wide solves keep a 33 KiB footprint)
    const int sw = wr + 1;                // LDS row stride (doubles)
    // stage the needed triangle: eight independent loads in flight per thread before the first LDS write
    // (a rolled load->write loop serialises on memory latency: 32 round trips, ~30 us per call, measured)
    for (int base = 0; base < w * wr; base += 8 * TRSM_THREADS) {
        double stage[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            const int r = idx & (wr - 1), k = idx / wr;
            const bool need = k < w && r < w && (MODE == 0 ? r > k : (MODE == 1 ? r >= k : r <= k));
            stage[u] = need ? T[r + (size_t)k * ldt] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            const int r = idx & (wr - 1), k = idx / wr;
            if (k < w) Ts[k * sw + r] = stage[u];
        }
    }
    __syncthreads();
    const int i = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * TRSM_THREADS + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * TRSM_THREADS) >> 6;
    const bool two = w > 64;
    double d0 = 1.0, d1 = 1.0;
    if (MODE != 0) {
        if (i < w) d0 = Ts[i * sw + i];
        if (64 + i < w) d1 = Ts[(64 + i) * sw + 64 + i];
    }
    // x / d sits on the dependency chain of every step (an fp64 division is ~30 dependent instructions).
    // When every diagonal entry has a finite, normal reciprocal the chain uses x * (1/d) instead (<= 1 ulp
    // from the quotient, far inside the solve's error bound); otherwise the exact division is kept.
    const double r0 = 1.0 / d0, r1 = 1.0 / d1;
    const bool recip_ok = MODE != 0 && !__any(!(fabs(r0) < 1.0e300 && fabs(r0) > 1.0e-300 && fabs(r1) < 1.0e300 && fabs(r1) > 1.0e-300));
    auto scale = [=](double x, double d, double r) { return recip_ok ? x * r : x / d; };
    for (size_t it = 0;; ++it) {
        size_t c0;
        if (counter) {
            unsigned ch = 0;
            if (i == 0) ch = atomicAdd(counter, 1u);
            c0 = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)ch) * TRSM_NC;
        } else {
            c0 = (wave + it * nwaves) * TRSM_NC;
        }
        if (c0 >= ncols) break;
        double x0[TRSM_NC], x1[TRSM_NC];
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j) {
            const bool live = c0 + j < ncols;
            x0[j] = (live && i < w) ? B[(c0 + j) * ldb + i] : 0.0;
            x1[j] = (live && 64 + i < w) ? B[(c0 + j) * ldb + 64 + i] : 0.0;
        }
        // selects, not branches and not zero multipliers: 0 * inf must not poison finished lanes.
        // Four steps per trip with their LDS operands fetched up front (a rolled loop exposes the LDS
        // latency in every step: ~0.2 us per step, measured).
        if (MODE != 2) {
            const int k0end = w < 64 ? w : 64;
#pragma unroll 1
            for (int kb = 0; kb < k0end; kb += 4) {
                double l0[4], l1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u < w ? kb + u : w - 1;
                    l0[u] = Ts[k * sw + i];
                    l1[u] = two ? Ts[k * sw + 64 + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u;
                    if (k < k0end) {
#pragma unroll
                        for (int j = 0; j < TRSM_NC; ++j) {
                            const double xk = bcast_lane(MODE == 0 ? x0[j] : scale(x0[j], d0, r0), k);  // final x_k
                            const double u0 = x0[j] - l0[u] * xk;
                            x0[j] = (MODE != 0 && i == k) ? xk : (i > k ? u0 : x0[j]);
                            x1[j] = two ? x1[j] - l1[u] * xk : x1[j];
                        }
                    }
                }
            }
#pragma unroll 1
            for (int kb = 64; kb < w; kb += 4) {
                double l1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u < w ? kb + u : w - 1;
                    l1[u] = two ? Ts[k * sw + 64 + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u;
                    if (k < w) {
#pragma unroll
                        for (int j = 0; j < TRSM_NC; ++j) {
                            const double xk = bcast_lane(MODE == 0 ? x1[j] : scale(x1[j], d1, r1), k - 64);
                            const double u1 = x1[j] - l1[u] * xk;
                            x1[j] = (MODE != 0 && 64 + i == k) ? xk : (64 + i > k ? u1 : x1[j]);
                        }
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int kb = w - 1; kb >= 64; kb -= 4) {
                double u0[4], u1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u >= 64 ? kb - u : 64;
                    u0[u] = Ts[k * sw + i];
                    u1[u] = two ? Ts[k * sw + 64 + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u;
                    if (k >= 64) {
#pragma unroll
                        for (int j = 0; j < TRSM_NC; ++j) {
                            const double xk = bcast_lane(scale(x1[j], d1, r1), k - 64);
                            const double v1 = x1[j] - u1[u] * xk;
                            x1[j] = (64 + i == k) ? xk : (64 + i < k ? v1 : x1[j]);
                            x0[j] -= u0[u] * xk;
                        }
                    }
                }
            }
#pragma unroll 1
            for (int kb = (w < 64 ? w : 64) - 1; kb >= 0; kb -= 4) {
                double u0[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u >= 0 ? kb - u : 0;
                    u0[u] = Ts[k * sw + i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u;
                    if (k >= 0) {
#pragma unroll
                        for (int j = 0; j < TRSM_NC; ++j) {
                            const double xk = bcast_lane(scale(x0[j], d0, r0), k);
                            const double v0 = x0[j] - u0[u] * xk;
                            x0[j] = (i == k) ? xk : (i < k ? v0 : x0[j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j) {
            if (c0 + j < ncols) {
                if (i < w) B[(c0 + j) * ldb + i] = x0[j];
                if (64 + i < w) B[(c0 + j) * ldb + 64 + i] = x1[j];
            }
        }
    }
}

// Unit-lower solve of 65..128 columns of T in TWO passes with half the LDS (64 x 129 doubles = 66 KiB instead of 132): pass 1
// stages T[:, 0:64] and runs steps 0..63 for every right-hand side of the block (x[0:64] final, x[64:128] partially updated, both
// written back), pass 2 stages the lower-right triangle into the same buffer and runs steps 64..w-1.  Per element the same
// operations in the same order as k_trsm_fused<0>: identical results.  At 66 KiB the block fits beside an update-stream dgemm
// block (84 KiB) instead of waiting for a whole CU - and keeping the dgemm off that CU while it runs.
template <int TRSM_NC, int TRSM_THREADS>
__global__ void __launch_bounds__(TRSM_THREADS) k_trsm_lower_2p(const double* __restrict__ T, size_t ldt, int w, double* __restrict__ B,
                                                                size_t ldb, size_t ncols) {
    extern __shared__ double Ts[];  // [64][sw]
    constexpr int sw = TRSM_W + 1;
    chain_prio();
    const int i = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * TRSM_THREADS + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * TRSM_THREADS) >> 6;
    // ---- pass 1: columns 0..63 of T, rows 0..w-1
    for (int base = 0; base < 64 * TRSM_W; base += 8 * TRSM_THREADS) {
        double stage[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            const int r = idx & (TRSM_W - 1), k = idx / TRSM_W;
            stage[u] = (k < 64 && r < w && r > k) ? T[r + (size_t)k * ldt] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            const int r = idx & (TRSM_W - 1), k = idx / TRSM_W;
            if (k < 64) Ts[k * sw + r] = stage[u];
        }
    }
    __syncthreads();
    for (size_t it = 0;; ++it) {
        const size_t c0 = (wave + it * nwaves) * TRSM_NC;
        if (c0 >= ncols) break;
        double x0[TRSM_NC], x1[TRSM_NC];
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j) {
            const bool live = c0 + j < ncols;
            x0[j] = live ? B[(c0 + j) * ldb + i] : 0.0;
            x1[j] = (live && 64 + i < w) ? B[(c0 + j) * ldb + 64 + i] : 0.0;
        }
#pragma unroll 1
        for (int kb = 0; kb < 64; kb += 4) {
            double l0[4], l1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                l0[u] = Ts[(kb + u) * sw + i];
                l1[u] = Ts[(kb + u) * sw + 64 + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u;
#pragma unroll
                for (int j = 0; j < TRSM_NC; ++j) {
                    const double xk = bcast_lane(x0[j], k);  // final x_k
                    const double u0 = x0[j] - l0[u] * xk;
                    x0[j] = i > k ? u0 : x0[j];
                    x1[j] = x1[j] - l1[u] * xk;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j) {
            if (c0 + j < ncols) {
                B[(c0 + j) * ldb + i] = x0[j];
                if (64 + i < w) B[(c0 + j) * ldb + 64 + i] = x1[j];
            }
        }
    }
    __syncthreads();  // every wave is done with the first half of T (and its partial x[64:] is on its way to memory)
    // ---- pass 2: the lower-right triangle, Ts[k - 64][r - 64]
    for (int base = 0; base < 64 * 64; base += 8 * TRSM_THREADS) {
        double stage[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            const int r = 64 + (idx & 63), k = 64 + idx / 64;
            stage[u] = (idx < 64 * 64 && k < w && r < w && r > k) ? T[r + (size_t)k * ldt] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * TRSM_THREADS + (int)threadIdx.x;
            if (idx < 64 * 64) Ts[(idx / 64) * sw + (idx & 63)] = stage[u];
        }
    }
    __syncthreads();
    for (size_t it = 0;; ++it) {
        const size_t c0 = (wave + it * nwaves) * TRSM_NC;
        if (c0 >= ncols) break;
        double x1[TRSM_NC];
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j) x1[j] = (c0 + j < ncols && 64 + i < w) ? B[(c0 + j) * ldb + 64 + i] : 0.0;  // this wave's own pass-1 stores
#pragma unroll 1
        for (int kb = 64; kb < w; kb += 4) {
            double l1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u < w ? kb + u : w - 1;
                l1[u] = Ts[(k - 64) * sw + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u;
                if (k < w) {
#pragma unroll
                    for (int j = 0; j < TRSM_NC; ++j) {
                        const double xk = bcast_lane(x1[j], k - 64);
                        const double u1 = x1[j] - l1[u] * xk;
                        x1[j] = 64 + i > k ? u1 : x1[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TRSM_NC; ++j)
            if (c0 + j < ncols && 64 + i < w) B[(c0 + j) * ldb + 64 + i] = x1[j];
    }
}

// Unit-lower triangular solve on the matrix cores (solve path, round 5): B <- L^-1 B for a 64- or 128-wide L whose 16 x 16 diagonal
// blocks k_rp_top has inverted (Context::lu_linv).  k_trsm_fused / k_trsm_lower_2p are chains of fp64 VALU FMAs fed by v_readlane:
// 10-15 us alone, 50-80 us beside the update streams' blocks (every fp64 instruction waits for the MFMA in flight on its SIMD), and
// 31 of them in a row are the W-wide solve of a super-panel boundary.  Here one wave owns 16 right-hand sides: right-looking over
// L's blocks, X_c = inv(L_cc) B_c, then B_b -= L_bc X_c for b > c - all v_mfma_f64_16x16x4 (D[i][j]: i = row, j = right-hand side), the
// accumulator layout of X_c (register r of lane (lq, l15) = row 4 r + lq) being the B operand of the updates (k = 4 s + lq at step
// s = r).  L's blocks come straight from memory as A operands (lane l15 = row: 128-byte segments; every wave reads the same 72 KiB:
// L2 hits), no LDS.  Rounding: products with an inverted 16 x 16 block instead of 16 substitution steps.
template <int NB>  // 16 x 16 blocks of L: 4 (w = 64) or 8 (w = 128)
__global__ void __launch_bounds__(64) k_trsm_lower_mfma(const double* __restrict__ T, const size_t ldt, const double* __restrict__ linv,
                                                        double* __restrict__ B, const size_t ldb, const size_t nc, unsigned* yield_tab) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int W = 16 * NB;
    constexpr int CS = W + 2;  // LDS column stride: 2 (mod 32) doubles puts the 32 lanes of a half-wave read on 64 distinct banks
    __shared__ __attribute__((aligned(16))) double tile[16 * CS];  // this wave's 16 right-hand sides, [column][row]
    const int t = threadIdx.x, l15 = t & 15, lq = t >> 4;
    chain_prio();
    const CuAnnounce on_cu(yield_tab);
    const size_t col0 = (size_t)blockIdx.x * 16;
    // coalesced load: one right-hand side per step, W rows as W / 2 lanes x 16 bytes (the rows of a column are contiguous in memory)
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        if (2 * t < W) {
            v2d v = v2d{0.0, 0.0};
            if (col0 + cc < nc) v = *(const v2d*)(B + (col0 + cc) * ldb + 2 * t);
            *(v2d*)(tile + cc * CS + 2 * t) = v;
        }
    }
    lds_barrier();
    v4d acc[NB];  // acc[b][r] = B[16 b + 4 r + lq][col0 + l15]
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[b][r] = tile[l15 * CS + 16 * b + 4 * r + lq];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        double iop[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) iop[s4] = linv[256 * c + (4 * s4 + lq) * 16 + l15];  // inv(L_cc)[i = l15][k = 4 s + lq]
        v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x = __builtin_amdgcn_mfma_f64_16x16x4f64(iop[s4], acc[c][s4], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[l15 * CS + 16 * c + 4 * r + lq] = x[r];
#pragma unroll
        for (int b = c + 1; b < NB; ++b) {
            double lop[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) lop[s4] = -T[(size_t)(16 * b + l15) + (size_t)(16 * c + 4 * s4 + lq) * ldt];  // -L_bc[i = l15][k = 4 s + lq]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[s4], x[s4], acc[b], 0, 0, 0);
        }
    }
    lds_barrier();
#pragma unroll
    for (int cc = 0; cc < 16; ++cc)
        if (2 * t < W && col0 + cc < nc) *(v2d*)(B + (col0 + cc) * ldb + 2 * t) = *(const v2d*)(tile + cc * CS + 2 * t);
    on_cu.done();
}

static int launch_check(Context* c);
template <int MODE, int NC, int TRSM_THREADS>
static int launch_trsm_fused_nc(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    // inside the look-ahead LU a 65..128-wide unit-lower solve takes the two-pass kernel (66 KiB: it shares CUs with the update
    // stream's dgemm blocks; RMHIP_LU_TRSM_2P=0: the 132 KiB one-pass kernel everywhere)
    static const int two_pass = std::getenv("RMHIP_LU_TRSM_2P") ? std::atoi(std::getenv("RMHIP_LU_TRSM_2P")) : 1;
    if (MODE == 0 && w > 64 && c->in_lookahead && two_pass) {
        const size_t lds2 = (size_t)64 * TRSM_SW * sizeof(double);
        c->ensure_max_lds((const void*)k_trsm_lower_2p<NC, TRSM_THREADS>, lds2);
        const size_t per_block2 = (size_t)(TRSM_THREADS / 64) * NC;
        size_t want2 = (nc + per_block2 - 1) / per_block2;
        const size_t cap2 = (size_t)c->num_cus * (two_pass > 1 ? (size_t)two_pass : 1);
        if (want2 < 1) want2 = 1;
        hipLaunchKernelGGL((k_trsm_lower_2p<NC, TRSM_THREADS>), dim3((unsigned)(want2 < cap2 ? want2 : cap2)), dim3(TRSM_THREADS), lds2,
                           c->stream, T, ldt, (int)w, B, ldb, nc);
        return launch_check(c);
    }
    const size_t lds_bytes = w * (w > 64 ? (size_t)TRSM_SW : (size_t)65) * sizeof(double);
    c->ensure_max_lds((const void*)k_trsm_fused<MODE, NC, TRSM_THREADS>, TRSM_W * TRSM_SW * sizeof(double));
    const size_t per_block = (size_t)(TRSM_THREADS / 64) * NC;  // columns one block solves per pass
    size_t want = (nc + per_block - 1) / per_block;
    const size_t cap = (size_t)c->num_cus * (w <= 64 ? 2 : 1);
    if (want < 1) want = 1;
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    // (counter-driven column groups + leaving the panels' XCD, as k_laswp_lists does in the LU's late phase, measured WORSE for
    // this kernel: 104.7 vs 103.5 ms at n = 16384 - the kernel arguments stay, the driver passes none)
    unsigned* counter = nullptr;
    hipLaunchKernelGGL((k_trsm_fused<MODE, NC, TRSM_THREADS>), dim3(grid), dim3(TRSM_THREADS), lds_bytes, c->stream, T, ldt, (int)w, B, ldb,
                       nc, counter, (const int*)nullptr);
    return launch_check(c);
}
template <int MODE>
static int launch_trsm_fused(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    if (MODE == 0 && c->lu_linv && (w == 64 || w == 128) && ldt == c->lu_work_ld && T >= c->lu_work && ((uintptr_t)B & 15) == 0 && ldb % 2 == 0) {
        // a diagonal block of the solve-path factorisation in flight: the matrix-core solve with k_rp_top's inverted 16 x 16 blocks
        const size_t off = (size_t)(T - c->lu_work);
        const size_t j = off / (ldt + 1);
        bool have = j * (ldt + 1) == off && j % 16 == 0 && c->lu_linv_ok && (j + w) / 16 <= c->lu_linv_ok->size();
        for (size_t q = j / 16; have && q < (j + w) / 16; ++q) have = (*c->lu_linv_ok)[q] != 0;
        if (have) {
            const double* linv = c->lu_linv + (j / 16) * 256;
            const unsigned grid = (unsigned)((nc + 15) / 16);
            unsigned* const ytab = (c->in_lookahead && c->gemm_lds_pad == 0 && c->lu_yield_trsm) ? c->gemm_announce_tab : nullptr;  // main stream of the two-level driver
            if (w == 64) hipLaunchKernelGGL(k_trsm_lower_mfma<4>, dim3(grid), dim3(64), 0, c->stream, T, ldt, linv, B, ldb, nc, ytab);
            else hipLaunchKernelGGL(k_trsm_lower_mfma<8>, dim3(grid), dim3(64), 0, c->stream, T, ldt, linv, B, ldb, nc, ytab);
            return launch_check(c);
        }
    }
    const size_t waves_on_chip = (size_t)c->num_cus * 4;  // one wave per SIMD
    return nc > 2 * waves_on_chip ? launch_trsm_fused_nc<MODE, 4, 512>(c, T, ldt, w, B, ldb, nc)
                                  : launch_trsm_fused_nc<MODE, 1, 256>(c, T, ldt, w, B, ldb, nc);
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
    static const bool log_shapes = std::getenv("RMHIP_LU_GEMM_LOG") != nullptr;  // developer aid: every update's shape and stream, in launch order (stderr)
    if (log_shapes) std::fprintf(stderr, "[lu_dgemm] stream %p m %zu n %zu k %zu pad %zu\n", (void*)c->stream, m, n, k, c->gemm_lds_pad);
    return launch_dgemm(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

static int launch_check(Context* c) {
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// Base size of the recursion: TRSM_W, or 64 when Context::trsm_base says so (under look-ahead a 64-wide solve's
// 33 KiB of LDS fits beside an update dgemm block, a 128-wide one needs a whole CU and waits for one to drain).
static size_t trsm_base(const Context* c) { return c->trsm_base == 64 ? 64 : (size_t)TRSM_W; }

// split point of a triangular solve wider than the base: whole base blocks on the left
static size_t trsm_split(size_t w, size_t base) {
    size_t h = ((w / 2 + base - 1) / base) * base;
    if (h >= w) h = w / 2;
    return h;
}

// B[w x nc] <- L^-1 B with L = unit-lower part of T[w x w]; recursive halving, dgemm in between.
static int trsm_lower_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc,
                          bool unit = true) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= trsm_base(c)) {
        if (lu_skip_mask() & 4) return RMHIP_OK;
        return unit ? launch_trsm_fused<0>(c, T, ldt, w, B, ldb, nc) : launch_trsm_fused<1>(c, T, ldt, w, B, ldb, nc);
    }
    const size_t h = trsm_split(w, trsm_base(c));
    RMHIP_TRY(trsm_lower_rec(c, T, ldt, h, B, ldb, nc, unit));
    RMHIP_TRY(lu_dgemm(c, w - h, nc, h, -1.0, T + h, ldt, B, ldb, 1.0, B + h, ldb));
    return trsm_lower_rec(c, T + h + h * ldt, ldt, w - h, B + h, ldb, nc, unit);
}

static int trsm_upper_rec(Context* c, const double* T, size_t ldt, size_t w, double* B, size_t ldb, size_t nc) {
    if (w == 0 || nc == 0) return RMHIP_OK;
    if (w <= trsm_base(c)) {
        if (lu_skip_mask() & 4) return RMHIP_OK;
        return launch_trsm_fused<2>(c, T, ldt, w, B, ldb, nc);
    }
    const size_t h = trsm_split(w, trsm_base(c));
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
    Context* c = s.c;
    unsigned* counter = nullptr;
    if (c->gemm_tile_counters && c->gemm_counter_next < c->gemm_counter_cap) counter = c->gemm_tile_counters + c->gemm_counter_next++;
    hipLaunchKernelGGL(k_laswp_lists, dim3((unsigned)((ncols + LASWP_COLS - 1) / LASWP_COLS)), dim3(2 * PLIST), 0,
                       s.c->stream, s.A, s.lda, c0, c1, s.plist, p0, p1, counter, counter ? c->gemm_avoid_xcc : nullptr);
    return launch_check(s.c);
}

// Most blocks a base panel may have and still sit on ONE XCD, one per CU (getrf_rec's placement rule; 0: never;
// RMHIP_LU_ONE_XCD=0 disables).
// (Tried: twice as many - tall panels, two blocks per CU in a 256-register build of the kernel - so that the update stream's
// persistent dgemm has the other seven XCDs to itself in the first half too: 124 ms against 102 at n = 16384.  A panel block
// that needs a whole SIMD's registers waits for every small kernel of the other streams to leave its CU; the panels then
// average 450 us instead of 210 and the main stream becomes the critical path from the start.)
static size_t one_xcd_block_limit(const Context* c) {
    static int one_xcd = -1;
    if (one_xcd < 0) {
        const char* v = std::getenv("RMHIP_LU_ONE_XCD");
        one_xcd = (v && *v == '0') ? 0 : 1;
    }
    return (one_xcd && c->one_xcd_ok) ? (size_t)c->num_cus / 8 : 0;
}

// retarget the context's stream (and the LDS pad / triangular-solve base that go with it) for a scope
struct StreamScope {
    Context* c;
    hipStream_t saved;
    size_t saved_pad;
    int saved_base;
    bool saved_prio;
    StreamScope(Context* ctx, hipStream_t s, size_t gemm_lds_pad, int base = 128)
        : c(ctx), saved(ctx->stream), saved_pad(ctx->gemm_lds_pad), saved_base(ctx->trsm_base), saved_prio(ctx->gemm_chain_prio) {
        c->stream = s;
        c->gemm_lds_pad = gemm_lds_pad;
        c->trsm_base = base;  // (128: the update stream owns whole CUs between its dgemm blocks anyway)
        c->gemm_chain_prio = false;  // only the main stream's dgemm launches raise their wave priority
    }
    ~StreamScope() {
        c->stream = saved;
        c->gemm_lds_pad = saved_pad;
        c->trsm_base = saved_base;
        c->gemm_chain_prio = saved_prio;
    }
};

// Factor columns [j0, j0+w) over rows [j0, rows); swaps are applied inside that column range only.
// own_swaps_by_caller: a base panel leaves its own columns un-interchanged (k_lu_panel2 stores every row where it was
// loaded from); the recursion step above it then widens the k_laswp_lists call that moves the sibling's columns anyway
// to this panel's columns, instead of a launch of its own.  *deferred reports that this happened.
static int getrf_rec(LuState& s, size_t j0, size_t w, bool own_swaps_by_caller = false, bool* deferred = nullptr) {
    if (deferred) *deferred = false;
    if (w == 0 || j0 >= s.rows) return RMHIP_OK;
    if (w <= (size_t)BASE_W) {
        const size_t c1 = j0 + w;  // j0 + w <= min(rows, cols) always holds (see lu_factor_device)
        const size_t nbp = (s.rows - j0 + P2_ROWS - 1) / P2_ROWS;
        if (s.fast && !(lu_skip_mask() & 1)) {
            // solve path: ONE workgroup factors the top P2_ROWS rows with partial pivoting (no exchange), k_rp_below turns every row
            // below into multipliers and records the largest one
            const size_t pid = s.panel_start->size();
            RtArgs g;
            g.A = s.A;
            g.lda = s.lda;
            g.rows = (s.rows - j0) > (size_t)RT_ROWS ? j0 + RT_ROWS : s.rows;
            g.j0 = (int)j0;
            g.w = (int)w;
            g.prow_arr = s.prow;
            g.ipiv = s.ipiv;
            g.info = s.info;
            g.plist = s.plist + pid * PLIST;
            double* const ucomp = s.ucomp + (size_t)(s.ucomp_slot++ % kUcompSlots) * UCOMP_STRIDE;
            g.ucomp = ucomp;
            g.xcc_out = s.panel_xcc;
            g.yield_word = s.yield_word;
            static const int rb_mfma = std::getenv("RMHIP_LU_RB_MFMA") ? std::atoi(std::getenv("RMHIP_LU_RB_MFMA")) : 1;
            const bool below_mfma = rb_mfma && w == (size_t)BASE_W && !s.xdbg;
            g.uinv_on = below_mfma ? 1 : 0;
            g.linv = (below_mfma && s.linv) ? s.linv + (j0 / 16) * 256 : nullptr;
            if (g.linv && s.c->lu_linv_ok)
                for (size_t q = j0 / 16; q < (j0 + BASE_W) / 16 && q < s.c->lu_linv_ok->size(); ++q) (*s.c->lu_linv_ok)[q] = 1;
            // like k_lu_panel2 the block ASKS for more LDS than it uses (48.6 KiB static) so that it does not share its CU with an
            // update-stream dgemm block: every column step would run slower beside one (RMHIP_LU_PANEL_PAD_KB; the phase-dependent
            // values of getrf_blocked apply)
            static long pad_kb_top = -1;
            if (pad_kb_top < 0) {
                const char* v = std::getenv("RMHIP_LU_PANEL_PAD_KB");
                pad_kb_top = v ? std::atol(v) : 96;  // (96: the workgroup has its CU to itself; 32: 78.3 -> 77.3 ms at n = 16384)
                if (pad_kb_top > 96) pad_kb_top = 96;
            }
            const size_t pad_bytes = (size_t)(s.panel_pad_kb >= 0 ? (s.panel_pad_kb > 96 ? 96 : s.panel_pad_kb) : pad_kb_top) * 1024;
            if (pad_bytes) {
                s.c->ensure_max_lds((const void*)k_rp_top<false>, 96 * 1024);
                s.c->ensure_max_lds((const void*)k_rp_top<true>, 96 * 1024);
            }
            if (s.xdbg) hipLaunchKernelGGL(k_rp_top<true>, dim3(1), dim3(RT_ROWS), pad_bytes, s.c->stream, g, (pk_u64*)s.xdbg);
            else hipLaunchKernelGGL(k_rp_top<false>, dim3(1), dim3(RT_ROWS), pad_bytes, s.c->stream, g, (pk_u64*)nullptr);
            RMHIP_TRY(launch_check(s.c));
            if (g.rows < s.rows) {
                static const int rb_threads = std::getenv("RMHIP_LU_RB_THREADS") ? std::atoi(std::getenv("RMHIP_LU_RB_THREADS")) : 64;  // developer knob (A/B)
                const size_t rbt = rb_threads == 256 ? 256 : (rb_threads == 128 ? 128 : 64);
                void (*kern)(double*, size_t, size_t, size_t, int, int, const double*, pk_u64*, double, pk_u64*) =
                    s.xdbg ? (rbt == 256 ? k_rp_below<256, true> : (rbt == 128 ? k_rp_below<128, true> : k_rp_below<64, true>))
                           : (rbt == 256 ? k_rp_below<256, false> : (rbt == 128 ? k_rp_below<128, false> : k_rp_below<64, false>));
                auto below = [&](hipStream_t st, size_t r0, size_t r1) {  // rows [r0, r1)
                    if (r1 <= r0) return;
                    if (below_mfma) {
                        hipLaunchKernelGGL(k_rp_below_mfma, dim3((unsigned)((r1 - r0 + 16 * RBM_JT - 1) / (16 * RBM_JT))), dim3(64), 0, st, s.A, s.lda, r1, r0, (int)j0,
                                           (const double*)ucomp, (pk_u64*)s.growth, (s.yield_all & 2) ? s.yield_word : (unsigned*)nullptr);
                        s.c->tel.kernel_launches++;
                        return;
                    }
                    hipLaunchKernelGGL(kern, dim3((unsigned)((r1 - r0 + rbt - 1) / rbt)), dim3((unsigned)rbt), 0, st, s.A, s.lda, r1, r0, (int)j0, (int)w,
                                       (const double*)ucomp, (pk_u64*)s.growth, s.tau, (pk_u64*)s.xdbg);
                    s.c->tel.kernel_launches++;
                };
                const size_t split = (s.aux && s.band_end > (size_t)g.rows && s.band_end < s.rows) ? s.band_end : s.rows;
                below(s.c->stream, g.rows, split);
                if (split < s.rows) {  // the rest of the rows: off the critical chain
                    hipEvent_t top_done = lu_new_event(s);
                    (void)hipEventRecord(top_done, s.c->stream);  // (recorded behind the band's k_rp_below: same stream order as the top block)
                    (void)hipStreamWaitEvent(s.aux, top_done, 0);
                    below(s.aux, split, s.rows);
                    s.aux_tail = lu_new_event(s);
                    (void)hipEventRecord(s.aux_tail, s.aux);
                }
                RMHIP_HIP_CHECK(hipGetLastError());
            }
            s.xbase += (unsigned)w;
            s.panel_start->push_back(j0);
            if (!s.screened) {
                // Early way out: a matrix whose large entries lie outside the top block (a row-permuted diagonally dominant one, say)
                // fails at its first panel; one small read here instead of a whole factorisation of wasted work.
                s.screened = true;
                unsigned long long bits = 0;
                if (s.aux) RMHIP_HIP_CHECK(hipStreamSynchronize(s.aux));  // its rows of the first panel count too
                RMHIP_HIP_CHECK(hipMemcpyAsync(&bits, s.growth, sizeof(bits), hipMemcpyDeviceToHost, s.c->stream));
                RMHIP_HIP_CHECK(hipStreamSynchronize(s.c->stream));
                double gmax;
                std::memcpy(&gmax, &bits, sizeof(gmax));
                if (!(gmax <= s.tau)) {
                    s.c->lu_last_growth = gmax;
                    return RMHIP_LU_GROWTH;
                }
            }
            if (own_swaps_by_caller && deferred) {
                *deferred = true;
                return RMHIP_OK;
            }
            return laswp(s, j0, c1, j0, c1);  // the panel's own interchange
        }
        if (s.persistent && nbp <= (size_t)PK_MAXB && nbp <= (size_t)s.c->num_cus && !(lu_skip_mask() & 1)) {
            // one launch factors the panel, interchanges its own columns and leaves the row-move list for the others
            const size_t pid = s.panel_start->size();
            P2Args g;
            g.A = s.A;
            g.lda = s.lda;
            g.rows = s.rows;
            g.j0 = (int)j0;
            g.w = (int)w;
            g.nblocks = (int)nbp;
            static int onehop_max = -1;  // developer knob
            if (onehop_max < 0) {
                const char* v = std::getenv("RMHIP_LU_ONEHOP");
                onehop_max = v ? std::atoi(v) : P2_ONEHOP_MAXB;
                if (onehop_max > P2_ONEHOP_MAXB) onehop_max = P2_ONEHOP_MAXB;
            }
            g.onehop_max = onehop_max;
            // One-XCD placement when every block finds a CU of its own on one XCD (<= num_cus / 8 blocks): the exchange hop
            // drops from a write-through + a miss in another XCD's L2 to two accesses of one L2 (scripts/micro/
            // xcd_exchange.hip: 2.1-2.5 -> 1.45 us per step).  RMHIP_LU_ONE_XCD=0 disables.
            g.bstride = nbp <= one_xcd_block_limit(s.c) ? 8 : 1;
            if (g.bstride > 1) s.c->lu_used_one_xcd = true;
            g.seq0 = s.xbase;
            g.xerr = s.xerr;
            g.xrec = s.xrec;
            g.xvals = s.xvals;
            g.prow_arr = s.prow;
            g.ipiv = s.ipiv;
            g.info = s.info;
            g.plist = s.plist + pid * PLIST;
            g.dbg = s.xdbg;
            g.xcc_out = s.panel_xcc;
            // The block needs 18.5 KiB of LDS but ASKS for 82.5: it then does not fit beside an update-stream dgemm block
            // (84 KiB) and waits for a CU of its own.  Sharing a CU costs more than the wait: with the panel waves on the
            // same SIMDs as a dgemm wave every column step slows down, and the column chain is the critical path
            // (n = 16384: 135 ms sharing, 123 ms exclusive; RMHIP_LU_PANEL_PAD_KB).
            static long pad_kb = -1;
            if (pad_kb < 0) {
                const char* v = std::getenv("RMHIP_LU_PANEL_PAD_KB");
                pad_kb = v ? std::atol(v) : 64;
                if (pad_kb > 128) pad_kb = 128;
            }
            const size_t lds_bytes = P2_LDS_DOUBLES * sizeof(double) + (size_t)(s.panel_pad_kb >= 0 ? s.panel_pad_kb : pad_kb) * 1024;
            if (lds_bytes > 65536) {
                s.c->ensure_max_lds((const void*)k_lu_panel2<false>, lds_bytes);
                s.c->ensure_max_lds((const void*)k_lu_panel2<true>, lds_bytes);
            }
            const unsigned grid = (unsigned)nbp * (unsigned)g.bstride;
            if (s.xdbg) hipLaunchKernelGGL((k_lu_panel2<true>), dim3(grid), dim3(P2_THREADS), lds_bytes, s.c->stream, g);
            else hipLaunchKernelGGL((k_lu_panel2<false>), dim3(grid), dim3(P2_THREADS), lds_bytes, s.c->stream, g);
            RMHIP_TRY(launch_check(s.c));
            s.xbase += (unsigned)w;
            s.panel_start->push_back(j0);
            if (own_swaps_by_caller && deferred) {
                *deferred = true;
                return RMHIP_OK;
            }
            return laswp(s, j0, c1, j0, c1);  // the panel's own interchange
        }
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
    bool left_deferred = false, right_deferred = false;
    RMHIP_TRY(getrf_rec(s, j0, h, true, &left_deferred));
    RMHIP_TRY(laswp(s, left_deferred ? j0 : j0 + h, j0 + w, j0, j0 + hk));
    double* A11 = s.A + j0 + j0 * s.lda;
    double* A12 = s.A + j0 + (j0 + h) * s.lda;
    RMHIP_TRY(trsm_lower_rec(s.c, A11, s.lda, hk, A12, s.lda, w - h));
    if (j0 + h < s.rows) {
        double* A21 = s.A + (j0 + h) + j0 * s.lda;
        double* A22 = s.A + (j0 + h) + (j0 + h) * s.lda;
        const size_t r0 = j0 + h;
        const size_t split = (s.fast && s.aux && s.band_end > r0 && s.band_end < s.rows) ? s.band_end : s.rows;
        RMHIP_TRY(lu_dgemm(s.c, split - r0, w - h, h, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda));
        if (split < s.rows) {
            // rows below the band: their multipliers come from aux's own k_rp_below launches (stream order), the U block row from the
            // triangular solve the main stream just enqueued
            hipEvent_t u_ready = lu_new_event(s);
            (void)hipEventRecord(u_ready, s.c->stream);
            (void)hipStreamWaitEvent(s.aux, u_ready, 0);
            {
                StreamScope scope(s.c, s.aux, s.c->gemm_lds_pad, s.c->trsm_base);
                RMHIP_TRY(lu_dgemm(s.c, s.rows - split, w - h, h, -1.0, A21 + (split - r0), s.lda, A12, s.lda, 1.0, A22 + (split - r0), s.lda));
            }
            s.aux_tail = lu_new_event(s);
            (void)hipEventRecord(s.aux_tail, s.aux);
        }
        RMHIP_TRY(getrf_rec(s, j0 + h, w - h, true, &right_deferred));
        const size_t k1 = (j0 + w <= s.rows) ? (j0 + w) : s.rows;
        RMHIP_TRY(laswp(s, j0, right_deferred ? j0 + w : j0 + h, j0 + h, k1));
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

// columns [c0, c1) receive the row interchanges of panel [j, j+w) and the U block row ...
static int prep_columns(LuState& s, size_t j, size_t w, size_t c0, size_t c1) {
    if (c1 <= c0) return RMHIP_OK;
    RMHIP_TRY(laswp(s, c0, c1, j, j + w));
    double* A11 = s.A + j + j * s.lda;
    double* A12 = s.A + j + c0 * s.lda;
    return trsm_lower_rec(s.c, A11, s.lda, w, A12, s.lda, c1 - c0);
}
// ... and the Schur update
static int gemm_columns(LuState& s, size_t j, size_t w, size_t c0, size_t c1) {
    if (c1 <= c0 || j + w >= s.rows) return RMHIP_OK;
    double* A12 = s.A + j + c0 * s.lda;
    double* A21 = s.A + (j + w) + j * s.lda;
    double* A22 = s.A + (j + w) + c0 * s.lda;
    return lu_dgemm(s.c, s.rows - j - w, c1 - c0, w, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda);
}
static int update_columns(LuState& s, size_t j, size_t w, size_t c0, size_t c1) {
    RMHIP_TRY(prep_columns(s, j, w, c0, c1));
    return gemm_columns(s, j, w, c0, c1);
}
// Incremental form of prep_columns(S0, W, c0, c1) for a super-panel [S0, S0 + W) whose inner panels finish one after the other: once
// panel [j, j + w) is factored (and its interchanges have reached the super-panel's left columns), columns [c0, c1) right of the
// super-panel receive its interchanges and ITS block row of U - the rows still lack the inner panels before it:
//   A[j : j+w, c] -= L[j : j+w, S0 : j] U[S0 : j, c];   A[j : j+w, c] <- L_jj^-1 A[j : j+w, c]
// After the last inner panel the columns hold what the W-wide solve would have produced (same operations, grouped by panel instead
// of by the solve's recursive halving) and only the deep rank-W update is left for the boundary.
static int iprep_columns(LuState& s, size_t S0, size_t j, size_t w, size_t c0, size_t c1) {
    if (c1 <= c0) return RMHIP_OK;
    RMHIP_TRY(laswp(s, c0, c1, j, j + w));
    double* A12 = s.A + j + c0 * s.lda;
    if (j > S0) RMHIP_TRY(lu_dgemm(s.c, w, c1 - c0, j - S0, -1.0, s.A + j + S0 * s.lda, s.lda, s.A + S0 + c0 * s.lda, s.lda, 1.0, A12, s.lda));
    return trsm_lower_rec(s.c, s.A + j + j * s.lda, s.lda, w, A12, s.lda, c1 - c0);
}

static int getrf_blocked(LuState& s, size_t kmin, size_t nb) {
    Context* c = s.c;
    hipStream_t main_stream = c->stream;
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    // (a high-priority stand-in for the context stream makes no difference: 100.7 vs 100.8 ms)
    // the update stream and the events live in the context (created once: a stream and ~200 events per factorisation cost
    // about a millisecond of host time)
    if (!c->lu_side_stream) RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&c->lu_side_stream, hipStreamNonBlocking, prio_low));
    hipStream_t side = c->lu_side_stream;
    // (Tried: the update stream on a CU-masked stream - hipExtStreamCreateWithCUMask, mask bit i = CU i/8 of XCD i%8 - that
    // leaves 32 or 64 CUs to the main stream, two unpadded dgemm blocks per CU on the rest: 140-169 ms against 123 at
    // n = 16384; the main stream's updates crawl on the reserved CUs.  Second attempt, round 2: only while the panels sit on
    // XCD 0, a mask that bars the update stream from XCD 0 except one CU - a mask that empties an XCD is ignored as a whole -
    // with the usual padded blocks: 370 ms against 112.  Work submitted through a CU-masked queue is slow here for
    // reasons beyond the CU count.)
    auto new_event = [&]() { return lu_new_event(s); };
    // solve path: the band split (LuState::aux).  Off by default: measured 75.0 / 42.6 / 21.7 ms with it against 74.4 / 42.0 / 20.8
    // without (n = 16384 / 12288 / 8192) - the update stream is the busy one for the first 60 ms and the extra stream only takes CUs
    // from it (profiles/r03_mldivide_timeline.txt).  RMHIP_LU_BAND=1 turns it on for A/B runs.
    static const int band_on = std::getenv("RMHIP_LU_BAND") ? std::atoi(std::getenv("RMHIP_LU_BAND")) : 0;
    static const long band_extra = std::getenv("RMHIP_LU_BAND_ROWS") ? std::atol(std::getenv("RMHIP_LU_BAND_ROWS")) : 320;  // rows of a top block (256) + one base panel
    if (s.fast && band_on) {
        if (!c->lu_aux_stream) RMHIP_HIP_CHECK(hipStreamCreateWithFlags(&c->lu_aux_stream, hipStreamNonBlocking));
        s.aux = c->lu_aux_stream;
    }
    // update-stream dgemm blocks ask for 84 KiB of LDS (73.7 needed): one per CU, leaving 76 KiB for a panel
    // block (66 KiB) or a main-stream dgemm block (73.7 KiB)
    size_t side_pad = 84 * 1024 - 73728;
    if (const char* v = std::getenv("RMHIP_LU_LA_PAD")) side_pad = (size_t)std::atoll(v);
    // main-stream triangular solves keep the 128-wide base: a 64-wide one (33 KiB of LDS) would fit beside an update
    // dgemm block instead of waiting for a CU to drain, but the extra launches cost more (140.7 vs 129.6 ms at
    // n = 16384; RMHIP_LU_LA_TRSM=64 selects it)
    std::shared_ptr<Allocation> late_ctl;  // tile counters of the persistent kernels (outlives the guard below)
    // the context fields this driver borrows go back on EVERY way out (an allocation or stream-creation failure below returns early),
    // and nothing of this factorisation is still running when they do
    struct Restore {
        Context* c;
        LuState* s;
        int trsm_base;
        ~Restore() {
            if (c->lu_prep_stream) (void)hipStreamSynchronize(c->lu_prep_stream);
            if (c->lu_side_stream) (void)hipStreamSynchronize(c->lu_side_stream);
            if (c->lu_aux_stream) (void)hipStreamSynchronize(c->lu_aux_stream);
            s->aux = nullptr;
            s->band_end = 0;
            c->gemm_tile_counters = nullptr;
            c->gemm_avoid_xcc = nullptr;
            c->gemm_counter_cap = 0;
            s->panel_xcc = nullptr;
            s->panel_pad_kb = -1;
            c->in_lookahead = false;
            c->trsm_base = trsm_base;
        }
    } restore{c, &s, c->trsm_base};
    if (const char* v = std::getenv("RMHIP_LU_LA_TRSM")) c->trsm_base = std::atoi(v) == 64 ? 64 : 128;
    c->in_lookahead = true;
    int rc = RMHIP_OK;
    // Late phase (the next panel fits one XCD, getrf_rec places it there): the update stream's dgemm runs as a persistent
    // kernel whose workgroups leave the panel's XCD at once (k_dgemm_w8p), and the panel blocks drop their LDS pad so that
    // such a placeholder can always be scheduled beside them.  The panel chain is the critical path there and everything
    // the update stream does disturbs it (without ANY update work in that phase the solve takes 95.7 instead of 108.4 ms):
    // this keeps the panel's CUs free (no drain before a panel or a 132 KiB triangular solve starts) and its L2 quiet.
    // RMHIP_LU_LATE_XCD=0 disables.
    // (solve path: off by default - its panel is one workgroup and k_rp_below's workgroups fit beside a dgemm block, so there is no XCD to
    // keep free; n = 16384: 78.8 -> 76.4 ms without the persistent form)
    static const int late_xcd_env = std::getenv("RMHIP_LU_LATE_XCD") ? std::atoi(std::getenv("RMHIP_LU_LATE_XCD")) : -1;  // 2: persistent dgemm in every phase (A/B)
    const int late_xcd_on = late_xcd_env >= 0 ? late_xcd_env : (s.fast ? 0 : 1);
    constexpr size_t kCounters = 4096;
    unsigned* late_counters = nullptr;
    if (late_xcd_on && c->one_xcd_ok) {
        RMHIP_TRY(c->alloc_device(kCounters / 2 + 8, &late_ctl));
        late_counters = (unsigned*)late_ctl->ptr;
        s.panel_xcc = (int*)(late_counters + kCounters);
        RMHIP_HIP_CHECK(hipMemsetAsync(late_counters, 0, kCounters * sizeof(unsigned), main_stream));
        RMHIP_HIP_CHECK(hipMemsetAsync(s.panel_xcc, 0xff, sizeof(int), main_stream));  // -1: no XCD to avoid yet
        c->gemm_counter_next = 0;
        c->gemm_counter_cap = kCounters;
        c->gemm_avoid_xcc = s.panel_xcc;
    }
    // Split update (RMHIP_LU_SPLIT=0 disables): on the update stream the interchanges and triangular solves are 14 % of the
    // throughput-bound first half, and the matrix cores idle meanwhile.  While more than split_rows rows remain (6144: into the
    // beginning of the late phase; 8192 measured 0.5 ms slower at n = 16384 and 8192) the trailing
    // columns are cut at a fixed column csplit into A | B; a third stream prepares (interchange + solve) one part while the
    // update stream's dgemm runs on the other:
    //   prep:    wait(P_j, dgemm_{j-1}(A)) -> prep_j(A) -> wait(dgemm_{j-1}(B)) -> prep_j(B)
    //   update:  wait(prep_j(A)) -> dgemm_j(A) -> wait(prep_j(B)) -> dgemm_j(B) -> interchanges of the finished left columns
    // csplit stays put (so A_j lies inside A_{j-1}) until A has shrunk below a fifth of the range, then it moves to the
    // middle again (that one step waits for all of step j-1).  The main stream waits for dgemm_{j-1}(A) only - the next
    // panel's columns are its first ones - in either mode.  Same kernels on the same columns: bit-identical factors.
    static const int split_on = std::getenv("RMHIP_LU_SPLIT") ? std::atoi(std::getenv("RMHIP_LU_SPLIT")) : 1;
    static const long split_rows_env = std::getenv("RMHIP_LU_SPLIT_ROWS") ? std::atol(std::getenv("RMHIP_LU_SPLIT_ROWS")) : 6144;
    static const int prep_base = std::getenv("RMHIP_LU_PREP_TRSM") ? std::atoi(std::getenv("RMHIP_LU_PREP_TRSM")) : 128;
    hipStream_t prep = nullptr;
    if (split_on) {
        if (!c->lu_prep_stream) RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&c->lu_prep_stream, hipStreamNonBlocking, prio_low));
        prep = c->lu_prep_stream;
    }
    hipEvent_t ev_a = nullptr, ev_b = nullptr;  // dgemm of the previous step finished on [.., a_end) / everything of that step finished
    size_t a_end = s.cols;                      // right edge of the previous step's part A
    size_t csplit = 0;
    {
        hipEvent_t e0 = new_event();  // side starts after whatever main already has queued (the copy of A)
        (void)hipEventRecord(e0, main_stream);
        (void)hipStreamWaitEvent(side, e0, 0);
        if (prep) (void)hipStreamWaitEvent(prep, e0, 0);
    }
    // Panel width by phase.  While the trailing matrix is large the update stream is the bottleneck and the main stream
    // idles a third of the time: wider panels there (fewer, deeper rank-k updates: the dgemm runs 53 instead of 47
    // TFLOP/s at k = 512 with one block per CU).  Once the panel chain is the critical path (about the last 8192
    // columns) the narrower panel wins.  n = 16384: 128.3 -> 125.2 ms (interleaved, scripts/lu_env_ab.sh); giving the
    // main stream a share of the trailing columns as well (its dgemm blocks fit beside the update stream's) measured
    // nothing (127.3 vs 127.9).
    size_t nb_early = 512, early_rows = 10240;  // (10240 since the update stream's eight-wave dgemm: 113.7 -> 112.7 ms; 8192 before)
    if (const char* v = std::getenv("RMHIP_LU_NB_EARLY")) nb_early = (size_t)std::atoll(v);
    if (const char* v = std::getenv("RMHIP_LU_EARLY_ROWS")) early_rows = (size_t)std::atoll(v);
    nb_early = nb_early < 64 ? 64 : (nb_early / 64) * 64;
    if (nb_early < nb) nb_early = nb;
    // third tier (RMHIP_LU_NB_LATE / RMHIP_LU_LATE_ROWS): once the panel chain is the critical path, a narrower panel
    // moves more of each update from the main stream (look-ahead columns) to the idle update stream
    // (n = 16384, interleaved: 123.1 / 124.2 ms without, 120.8 / 120.6 with 128 below 6144 rows; 64 below 3072: 122.9 / 121.6)
    size_t nb_late = 128, late_rows = 8192;  // (8192 since the two-pass solve and the third stream: 98.4 vs 99.1 ms at n = 16384; 6144 before)
    if (const char* v = std::getenv("RMHIP_LU_NB_LATE")) nb_late = (size_t)std::atoll(v);
    if (const char* v = std::getenv("RMHIP_LU_LATE_ROWS")) late_rows = (size_t)std::atoll(v);
    nb_late = nb_late < 64 ? 64 : (nb_late / 64) * 64;
    if (nb_late > nb) nb_late = nb;
    // the very first panel is narrower (the update stream has nothing to do until it is factored: 128 columns instead of
    // 512 start it 1.5 ms earlier; n = 16384: 108.9 -> 107.9 ms); the second one realigns to multiples of the wide width
    // (RMHIP_LU_NB_FIRST, 0 = off)
    size_t nb_first = 128;
    if (const char* v = std::getenv("RMHIP_LU_NB_FIRST")) nb_first = ((size_t)std::atoll(v) / 64) * 64;
    auto width_at = [&](size_t j) {
        if (nb_first && nb_first < nb_early && early_rows && kmin > early_rows + nb_early) {
            if (j == 0) return nb_first;
            if (j == nb_first) return nb_early - nb_first;
        }
        if (early_rows && kmin - j > early_rows) return nb_early;
        if (late_rows && kmin - j <= late_rows) return nb_late;
        return nb;
    };
    // While the trailing matrix is large the machine is throughput bound on the update dgemm (one block per CU: ~50 TFLOP/s
    // in place, 56 alone) and the main stream has slack, so the panel blocks drop their LDS pad there and start beside a
    // dgemm block instead of waiting for a CU to drain (n = 16384: 119.9 -> 118.8 ms).  Letting the update stream drop ITS
    // pad there as well (two dgemm blocks per CU) starves the main stream: a retiring block's slot goes to the next block
    // of the same kernel, stream priority or not (138 ms).  Developer knobs: RMHIP_LU_EARLY_SIDE_PAD (bytes),
    // RMHIP_LU_EARLY_PANEL_PAD_KB.
    // (Also tried: the interchanges of the finished left columns on a third low-priority stream - they only depend on their panel and
    // on the previous update - instead of behind the update: 109.25 vs 108.6 ms, 8192: 40.4 vs 39.7 ms.)
    // (Also tried: the first quarter / half of the trailing columns on a second update stream, so that its dgemm covers the
    // row interchange and triangular solve of the rest - 125-130 ms against 119.)
    // (With the eight-wave update kernel trimmed to 128 VGPRs and the panel to 256 the two do share a CU again - and the solve
    // takes 119 ms instead of 108: a panel wave beside TWO MFMA waves per SIMD crawls.  Exclusive CUs for the panel it is.)
    size_t early_side_pad = side_pad;
    long early_panel_pad = 0;
    if (const char* v = std::getenv("RMHIP_LU_EARLY_SIDE_PAD")) early_side_pad = (size_t)std::atoll(v);
    if (const char* v = std::getenv("RMHIP_LU_EARLY_PANEL_PAD_KB")) early_panel_pad = std::atol(v);
    // (Tried: DEFERRED update of the far columns.  While the chain is left of column far_c every step updates only [next panel,
    // far_c) and the columns right of it collect their panels - applied later, range by range, with one interchange pass
    // (F may see later interchanges early: Pi (F - L U) = Pi F - (Pi L) U), one triangular solve and one rank-(b - a) update,
    // scheduled by a cost model into the time the update stream has to spare and flushed before the chain gets there.
    // Bit-identical pivots, correct factors - and slower at every far_c: 113.3 ms at 10240, 111.4 at 12288, 109.6 at 13312
    // against 108.5 without.  The update stream's idle time in the second half is not free: whatever runs there slows the
    // panel chain - exchange latency under load, CUs that must drain before a panel or a triangular solve starts.)
    for (size_t j = 0; j < kmin && rc == RMHIP_OK;) {
        const size_t nbj = width_at(j);
        const bool early = early_rows && kmin - j > early_rows;
        // this panel on one XCD?  (the condition getrf_rec applies to its first base panel)
        const bool late_xcd = late_counters && (s.rows - j + P2_ROWS - 1) / P2_ROWS <= one_xcd_block_limit(c);
        static const long late_panel_pad = std::getenv("RMHIP_LU_LATE_PANEL_PAD_KB") ? std::atol(std::getenv("RMHIP_LU_LATE_PANEL_PAD_KB")) : 0;
        s.panel_pad_kb = late_xcd ? late_panel_pad : (early ? early_panel_pad : -1);
        const size_t w = (kmin - j) < nbj ? (kmin - j) : nbj;
        if (s.aux) {
            const size_t be = j + w + (size_t)band_extra;
            s.band_end = be < s.rows ? be : 0;  // nothing below the band: no split
            s.aux_tail = nullptr;
        }
        rc = getrf_rec(s, j, w);  // P_j on main (and, below the band, on aux)
        if (rc != RMHIP_OK) break;
        hipEvent_t panel_done = new_event();
        (void)hipEventRecord(panel_done, main_stream);
        hipEvent_t aux_done = s.aux_tail;  // aux's share of P_j (nullptr: it had none)
        const size_t panel_band_end = s.band_end;
        const size_t next = j + w;
        size_t la_w = 0;
        if (next < kmin) {  // there is a next panel: the main stream updates its columns right away
            const size_t nbn = width_at(next);
            la_w = (kmin - next) < nbn ? (kmin - next) : nbn;
        }
        const size_t t0 = next + la_w;  // trailing columns [t0, cols)
        const bool next_late = late_counters && next < s.rows && (s.rows - next + P2_ROWS - 1) / P2_ROWS <= one_xcd_block_limit(c);
        bool split = prep && kmin - j > (size_t)split_rows_env && s.cols > t0 && s.cols - t0 >= 2048;
        bool moved = false;
        if (split) {
            if (csplit < t0 + 256 || csplit >= s.cols || (csplit - t0) * 5 < (s.cols - t0)) {
                csplit = t0 + (((s.cols - t0) / 2 + 127) / 128) * 128;
                moved = true;
            }
            if (csplit >= s.cols) split = false;
        }
        // (Tried: while the update stream is the bottleneck, hand it the next panel's columns as well - first in line, prepared by
        // the third stream - so that the main stream only factors panels: 101.9-103.9 ms against 101.3, worse the longer it is kept up.)
        if (la_w) {
            if (ev_a) (void)hipStreamWaitEvent(main_stream, ev_a, 0);
            if (ev_b && next + la_w > a_end) (void)hipStreamWaitEvent(main_stream, ev_b, 0);
            if (s.aux && panel_band_end) {
                // Band split of the look-ahead update: the next panel's chain needs rows [next, next + la_w + band) of its columns;
                // of those, rows beyond THIS panel's band got their multipliers on aux (one wait per look-ahead panel).  Everything below
                // goes to aux, behind its share of P_j.
                rc = prep_columns(s, j, w, next, next + la_w);  // interchange + U block row: top rows only
                if (rc != RMHIP_OK) break;
                hipEvent_t u_ready = new_event();
                (void)hipEventRecord(u_ready, main_stream);
                size_t be_next = next + la_w + (size_t)band_extra;
                if (be_next > s.rows) be_next = s.rows;
                const size_t r0 = j + w;
                double* A12 = s.A + j + next * s.lda;
                double* A21 = s.A + r0 + j * s.lda;
                double* A22 = s.A + r0 + next * s.lda;
                if (aux_done) (void)hipStreamWaitEvent(main_stream, aux_done, 0);
                if (be_next > r0) rc = lu_dgemm(c, be_next - r0, la_w, w, -1.0, A21, s.lda, A12, s.lda, 1.0, A22, s.lda);
                if (rc != RMHIP_OK) break;
                if (be_next < s.rows) {
                    (void)hipStreamWaitEvent(s.aux, u_ready, 0);
                    if (ev_a) (void)hipStreamWaitEvent(s.aux, ev_a, 0);
                    if (ev_b && next + la_w > a_end) (void)hipStreamWaitEvent(s.aux, ev_b, 0);
                    {
                        StreamScope scope(c, s.aux, c->gemm_lds_pad, c->trsm_base);
                        rc = lu_dgemm(c, s.rows - be_next, la_w, w, -1.0, A21 + (be_next - r0), s.lda, A12, s.lda, 1.0, A22 + (be_next - r0), s.lda);
                    }
                    if (rc != RMHIP_OK) break;
                    s.aux_tail = new_event();
                    (void)hipEventRecord(s.aux_tail, s.aux);
                }
            } else {
                rc = update_columns(s, j, w, next, next + la_w);
                if (rc != RMHIP_OK) break;
            }
        }
        if (split) {
            c->gemm_tile_counters = next_late ? late_counters : nullptr;  // persistent, XCD-avoiding kernels if the next panel sits on one XCD
            hipEvent_t ra = new_event(), rb = new_event();
            (void)hipStreamWaitEvent(prep, panel_done, 0);
            if (aux_done) (void)hipStreamWaitEvent(side, aux_done, 0);  // the Schur update reads every row of L(P_j)
            {
                StreamScope scope(c, prep, 0, prep_base);  // unpadded small blocks: they run beside the update stream's dgemm
                if (ev_a) (void)hipStreamWaitEvent(prep, ev_a, 0);
                if (ev_b && (moved || csplit > a_end)) (void)hipStreamWaitEvent(prep, ev_b, 0);
                rc = prep_columns(s, j, w, t0, csplit);
                (void)hipEventRecord(ra, prep);
                if (ev_b) (void)hipStreamWaitEvent(prep, ev_b, 0);
                if (rc == RMHIP_OK) rc = prep_columns(s, j, w, csplit, s.cols);
                (void)hipEventRecord(rb, prep);
            }
            StreamScope scope(c, side, early ? early_side_pad : side_pad);
            (void)hipStreamWaitEvent(side, ra, 0);
            if (rc == RMHIP_OK) rc = gemm_columns(s, j, w, t0, csplit);
            ev_a = new_event();
            (void)hipEventRecord(ev_a, side);
            (void)hipStreamWaitEvent(side, rb, 0);
            if (rc == RMHIP_OK) rc = gemm_columns(s, j, w, csplit, s.cols);
            if (rc == RMHIP_OK && j > 0) rc = laswp(s, 0, j, j, j + w);  // finished left columns
            c->gemm_tile_counters = nullptr;
            a_end = csplit;
        } else {
            (void)hipStreamWaitEvent(side, panel_done, 0);
            if (aux_done) (void)hipStreamWaitEvent(side, aux_done, 0);
            StreamScope scope(c, side, early ? early_side_pad : side_pad);
            // S_j overlaps panel j+1: persistent, XCD-avoiding dgemm if that panel sits on one XCD
            c->gemm_tile_counters = (next_late || (late_xcd_on == 2 && late_counters)) ? late_counters : nullptr;
            // (Attribution, n = 16384 at 100.9 ms: dropping the update stream's work of this phase altogether - knob below, results
            // are garbage - gives 100.3: with the XCD partition it no longer disturbs the chain.  But the phase has no room to
            // spare either: repeating its dgemm into a scratch copy k more times, on this stream before or behind the event the main
            // stream waits for or on the third stream, costs 6 ms per repeat (12 ms of kernel time each); repeats only below 6144 /
            // 4096 remaining rows cost what the work takes at 55-75 TFLOP/s.  Each step's update just fits the time of the next
            // panel, so deferring first-half work into this phase buys (1/43 - 1/60 TFLOP/s) per flop at best.)
            static const int late_skip = std::getenv("RMHIP_LU_LATE_SKIP") ? std::atoi(std::getenv("RMHIP_LU_LATE_SKIP")) : 0;
            if (!(next_late && late_skip)) rc = update_columns(s, j, w, t0, s.cols);      // S_j
            ev_a = new_event();
            (void)hipEventRecord(ev_a, side);
            if (rc == RMHIP_OK && j > 0) rc = laswp(s, 0, j, j, j + w);  // finished left columns
            c->gemm_tile_counters = nullptr;
            a_end = s.cols;
        }
        ev_b = new_event();
        (void)hipEventRecord(ev_b, side);
        j = next;
    }
    s.panel_pad_kb = -1;
    if (ev_b) (void)hipStreamWaitEvent(main_stream, ev_b, 0);
    if (prep) (void)hipStreamSynchronize(prep);
    (void)hipStreamSynchronize(side);
    if (s.aux) (void)hipStreamSynchronize(s.aux);
    (void)hipStreamSynchronize(main_stream);
    return rc;
}

// ---- two-level blocked driver (solve path, round 5) --------------------------------------------------------------------------
// The one-level driver above applies every panel of nb <= 512 columns to the WHOLE trailing matrix at once: 120 in-place rank-k
// updates that read and write all of C for 128..512 steps of k each - the C tile's load and store, the first operand tiles and a
// k loop whose operands are touched once cost a third of the matrix pipe's time (50 TFLOP/s at k = 512, 34 at k = 128 against 72
// for a deep product; profiles/r04_lu_attribution.txt).  Here the columns are grouped into SUPER-PANELS of W columns (2048 while the
// trailing matrix is large).  Inside a super-panel the panels of nb columns are factored with look-ahead exactly as above, but their
// updates stop at the super-panel's right edge; the columns beyond it receive the whole super-panel at once - one interchange pass,
// one W-wide triangular solve, ONE rank-W update (k = W: the C traffic of 2048 / nb updates paid once) - on a third stream:
//   main:  P_j -> LA_j (next panel's columns; at a super-panel boundary: the rank-W update of the next super-panel's first panel)
//   mid:   the other columns of the super-panel in flight (rank-nb; at a boundary: rank-W of the next super-panel's other columns)
//          and the interchanges of the super-panel's own left columns (L21 of the super-panel is final when it completes)
//   far:   at a boundary, everything right of the next super-panel: interchange + W-wide solve + rank-W update, the columns of the
//          super-panel after next first (event), then the rest; last the interchanges of the columns left of the super-panel
// A super-panel of ONE panel is the one-level scheme (far = the update stream above), which is what the plan ends with once the
// panel chain is the critical path.  Same kernels, same per-element operation order inside a panel; the trailing updates sum in
// super-panel-sized groups (results agree with the one-level driver to rounding, pivots are identical).
// ---- round 6: the W-wide unit-lower solves of the two-level driver as products with the explicit inverse ----------------------------
// At a super-panel boundary every column right of it needs U12 = L11^-1 A12 with L11 the W x W unit-lower block of the super-panel
// (W = 1024 / 2048) before the deep rank-W update.  As a recursive solve that is 31 launches of 128-wide solves and few-tile products
// at low fill: 17 of the far stream's 58 ms, 1.2 of the 2 ms the main stream spends at every boundary, and the mid stream's 3 ms
// before every second inner panel (round 5's timeline).  Instead the mid stream builds M = L11^-1 row block by row block while the
// super-panel's inner panels are being factored (row block p as soon as panel p's interchanges have reached the columns left of it:
// M_pp = L_pp^-1 by the 256-wide solve on an identity, M_p,0:p = -M_pp (L_p,0:p M_0:p,0:p), two small products), and the boundary
// computes T = M A12 as ONE deep product per column chunk (twice the flops of the substitution, all of them at the deep-product rate),
// updates with T as the B operand and copies T into A12 afterwards.
// Error: |T - U12| <= c W eps |M| |L11| |U12|-like, i.e. the condition of L11 enters where substitution is backward stable.  The
// solve path already bounds every multiplier by tau = 8; here the largest |M_ij| of every super-panel is recorded as well (k_absmax_word)
// and a value above RMHIP_LU_MINV_MAX (default 1e6) sends the factorisation to the caller's fall-back like a multiplier above tau.
// RMHIP_LU_MINV=0 restores the recursive solves.
__global__ void __launch_bounds__(256) k_set_identity(double* __restrict__ M, size_t ld, unsigned w) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)w * w) return;
    const unsigned r = (unsigned)(idx % w), cidx = (unsigned)(idx / w);
    M[r + (size_t)cidx * ld] = r == cidx ? 1.0 : 0.0;
}
__global__ void __launch_bounds__(256) k_absmax_word(const double* __restrict__ M, size_t ld, unsigned rows, unsigned cols, pk_u64* __restrict__ word) {
    pk_u64 mx = 0;
    const size_t total = (size_t)rows * cols;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const pk_u64 b = (pk_u64)__double_as_longlong(M[idx % rows + (idx / rows) * ld]) & 0x7fffffffffffffffull;  // |x| (NaN sorts above Inf)
        mx = b > mx ? b : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const pk_u64 o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx != 0) atomicMax(word, mx > 0x7ff0000000000000ull ? 0x7ff8000000000000ull : mx);
}

struct SuperPanel {
    size_t s0, s1, nb;
};

static std::vector<std::pair<size_t, size_t>> parse_super_seq(const char* v) {
    std::vector<std::pair<size_t, size_t>> out;  // (W, nb)
    while (v && *v) {
        char* end = nullptr;
        const size_t W = (size_t)std::strtoull(v, &end, 10);
        size_t nb = W;
        if (end && *end == ':') nb = (size_t)std::strtoull(end + 1, &end, 10);
        if (W >= 64) out.emplace_back((W / 64) * 64, nb < 64 ? 64 : (nb / 64) * 64);
        v = end;
        while (v && (*v == ',' || *v == '/' || *v == ' ')) ++v;
        if (!end) break;
    }
    return out;
}

static int getrf_super(LuState& s, size_t kmin) {
    Context* c = s.c;
    hipStream_t main_stream = c->stream;
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    // (mid at the high priority - four hardware queues, below the cliff described next: 68.4-68.9 against 67.8-68.2 ms)
    if (!c->lu_mid_stream) RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&c->lu_mid_stream, hipStreamNonBlocking, (prio_low + prio_high) / 2));
    // Both update streams at the DEFAULT priority: HIP keeps default-priority streams inside a pool of four hardware queues, a stream of
    // another priority gets a queue of its own - and with five queues in use by the process (null stream, main, mid, a low-priority far
    // and any fourth stream of the driver's) every solve took 100-120 ms instead of 72: the same cliff hit the prep-stream, band-split
    // and seat-holder experiments of round 5.  With far at the default priority a fourth stream costs 2 ms, not 47 (docs/EXPERIMENTS.md R5).
    if (!c->lu_far_stream) RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&c->lu_far_stream, hipStreamNonBlocking, (prio_low + prio_high) / 2));
    hipStream_t mid = c->lu_mid_stream, far = c->lu_far_stream;
    std::shared_ptr<Allocation> yield_ctl;  // the yield word (outlives the guard below)
    struct Restore {
        Context* c;
        LuState* s;
        int trsm_base;
        bool clear_yield = false;
        ~Restore() {
            s->yield_word = nullptr;
            c->gemm_chain_prio = false;
            if (c->lu_mid_stream) (void)hipStreamSynchronize(c->lu_mid_stream);
            if (c->lu_far_stream) (void)hipStreamSynchronize(c->lu_far_stream);
            c->gemm_yield_word = nullptr;
            c->gemm_announce = nullptr;
            c->gemm_announce_tab = nullptr;
            c->lu_yield_trsm = false;
            s->panel_pad_kb = -1;
            c->in_lookahead = false;
            c->trsm_base = trsm_base;
        }
    } restore{c, &s, c->trsm_base};
    if (const char* v = std::getenv("RMHIP_LU_LA_TRSM")) c->trsm_base = std::atoi(v) == 64 ? 64 : 128;
    c->in_lookahead = true;
    static const int gemm_prio_on = std::getenv("RMHIP_LU_GEMM_PRIO") ? std::atoi(std::getenv("RMHIP_LU_GEMM_PRIO")) : 1;
    c->gemm_chain_prio = gemm_prio_on != 0;
    // ---- the plan: super-panels (W, nb) while more than super_rows rows remain, single panels afterwards
    // (defaults by order, interleaved runs of scripts/lu_super_ab.sh: n = 16384 69.7-69.9 ms with 256 / 1024 / 2048-column super-panels of
    // 256-column panels down to 4096 remaining rows against 71.6-72.7 with the plan below; 12288 41.0-41.5 against 40.9, 8192 22.1 against 20.8:
    // the shorter plan stays below 14336)
    static const std::vector<std::pair<size_t, size_t>> seq_env = parse_super_seq(std::getenv("RMHIP_LU_SUPER_SEQ"));
    const bool large = kmin >= 14336;
    const std::vector<std::pair<size_t, size_t>> seq =
        !seq_env.empty() ? seq_env
                         : (large ? std::vector<std::pair<size_t, size_t>>{{256, 256}, {1024, 256}, {2048, 256}}
                                  : std::vector<std::pair<size_t, size_t>>{{512, 512}, {1024, 512}, {2048, 512}});
    static const std::pair<size_t, size_t> late = [] {
        auto q = parse_super_seq(std::getenv("RMHIP_LU_SUPER_LATE"));
        return q.empty() ? std::pair<size_t, size_t>{128, 128} : q[0];
    }();
    static const long super_rows_env = std::getenv("RMHIP_LU_SUPER_ROWS") ? std::atol(std::getenv("RMHIP_LU_SUPER_ROWS")) : -1;
    const size_t super_rows = super_rows_env >= 0 ? (size_t)super_rows_env : (large ? 4096 : 6144);
    std::vector<SuperPanel> plan;
    for (size_t s0 = 0, i = 0; s0 < kmin;) {
        const size_t rem = kmin - s0;
        std::pair<size_t, size_t> e = late;
        if (rem > super_rows) {
            e = seq[i < seq.size() ? i : seq.size() - 1];
            ++i;
            if (e.first > rem - super_rows && rem - super_rows >= e.second) e.first = ((rem - super_rows + e.second - 1) / e.second) * e.second;  // do not overshoot the switch point by much
        }
        const size_t s1 = s0 + e.first < kmin ? s0 + e.first : kmin;
        plan.push_back({s0, s1, e.second < e.first ? e.second : e.first});
        s0 = s1;
    }
    // LDS asked for by the update streams' dgemm blocks (see getrf_blocked: 84 KiB = one eight-wave block per CU with room beside it)
    const size_t pad_default = 84 * 1024 - 73728;
    static const long mid_pad_env = std::getenv("RMHIP_LU_MID_PAD") ? std::atol(std::getenv("RMHIP_LU_MID_PAD")) : -1;
    static const long far_pad_env = std::getenv("RMHIP_LU_FAR_PAD") ? std::atol(std::getenv("RMHIP_LU_FAR_PAD")) : -1;
    static const long super_panel_pad = std::getenv("RMHIP_LU_SUPER_PANEL_PAD_KB") ? std::atol(std::getenv("RMHIP_LU_SUPER_PANEL_PAD_KB")) : 0;
    const size_t mid_pad = mid_pad_env >= 0 ? (size_t)mid_pad_env : pad_default;
    const size_t far_pad = far_pad_env >= 0 ? (size_t)far_pad_env : pad_default;
    // cooperative yield of the update blocks on k_rp_top's CU (RMHIP_LU_YIELD=0 disables)
    static const int yield_on = std::getenv("RMHIP_LU_YIELD") ? std::atoi(std::getenv("RMHIP_LU_YIELD")) : 1;
    static const long top_pad_env = std::getenv("RMHIP_LU_TOP_PAD_KB") ? std::atol(std::getenv("RMHIP_LU_TOP_PAD_KB")) : (yield_on ? 0 : -1);
    if (yield_on) {
        RMHIP_TRY(c->alloc_device(kYieldSlots / 2, &yield_ctl));
        RMHIP_HIP_CHECK(hipMemsetAsync(yield_ctl->ptr, 0, sizeof(unsigned) * kYieldSlots, main_stream));
        s.yield_word = (unsigned*)yield_ctl->ptr;
        c->gemm_yield_word = s.yield_word;
        restore.clear_yield = true;
        // round 6: other chain kernels count themselves in as well (RMHIP_LU_YIELD_ALL bit mask: 1 the main stream's dgemm blocks,
        // 2 k_rp_below_mfma, 4 k_trsm_lower_mfma; 0 = k_rp_top alone as in round 5)
        static const int yield_all = std::getenv("RMHIP_LU_YIELD_ALL") ? std::atoi(std::getenv("RMHIP_LU_YIELD_ALL")) : 0;
        s.yield_all = yield_all;
        c->gemm_announce = (yield_all & 1) ? s.yield_word : nullptr;
        c->gemm_announce_tab = s.yield_word;
        c->lu_yield_trsm = (yield_all & 4) != 0;
    }
    // ---- explicit inverses of the super-panels' L11 blocks (see the note above k_set_identity)
    const int minv_env = std::getenv("RMHIP_LU_MINV") ? std::atoi(std::getenv("RMHIP_LU_MINV")) : 0;  // (read per call: the tests run both forms; default off, see docs/EXPERIMENTS.md R6 1)
    static const int minv_who = std::getenv("RMHIP_LU_MINV_WHO") ? std::atoi(std::getenv("RMHIP_LU_MINV_WHO")) : 7;   // dev: 1 main, 2 mid, 4 far use the inverse
    static const int minv_ext_far = std::getenv("RMHIP_LU_MINV_EXT") ? std::atoi(std::getenv("RMHIP_LU_MINV_EXT")) : 0;  // dev: the inverse is built on the far stream
    static const size_t far_chunk = std::getenv("RMHIP_LU_MINV_CHUNK") ? (size_t)std::atol(std::getenv("RMHIP_LU_MINV_CHUNK")) / 128 * 128 : 4096;
    size_t Wmax = 0, nbmax = 0;
    for (const SuperPanel& sp : plan)
        if (sp.s1 - sp.s0 > sp.nb) {
            Wmax = sp.s1 - sp.s0 > Wmax ? sp.s1 - sp.s0 : Wmax;
            nbmax = sp.nb > nbmax ? sp.nb : nbmax;
        }
    std::shared_ptr<Allocation> minv_mem[2], mtmp_mem, t_main_mem, t_mid_mem, t_far_mem, split_ws[3];  // (split-K partials: one per stream)
    const size_t kSplitWsElems = (size_t)8 << 20;  // at most 64 output tiles x 8 slices
    const bool minv_on = minv_env != 0 && Wmax > 0 && s.minv_max != nullptr && far_chunk >= 128;
    if (minv_on) {
        const size_t far_cols = far_chunk > Wmax ? far_chunk : Wmax;
        RMHIP_TRY(c->alloc_device(Wmax * Wmax, &minv_mem[0]));
        RMHIP_TRY(c->alloc_device(Wmax * Wmax, &minv_mem[1]));
        RMHIP_TRY(c->alloc_device(nbmax * Wmax, &mtmp_mem));
        RMHIP_TRY(c->alloc_device(Wmax * 512, &t_main_mem));
        RMHIP_TRY(c->alloc_device(Wmax * Wmax, &t_mid_mem));
        RMHIP_TRY(c->alloc_device(Wmax * far_cols, &t_far_mem));
        for (auto& w : split_ws) RMHIP_TRY(c->alloc_device(kSplitWsElems, &w));
    }
    hipEvent_t ev_minv = nullptr;                      // the inverse of the super-panel in flight is complete (mid stream)
    hipEvent_t ev_minv_read[2] = {nullptr, nullptr};   // the far stream's last product with inverse buffer 0 / 1
    auto new_event = [&]() { return lu_new_event(s); };
    auto record = [&](hipStream_t st) {
        hipEvent_t e = new_event();
        (void)hipEventRecord(e, st);
        return e;
    };
    {
        hipEvent_t e0 = record(main_stream);  // the update streams start after whatever main already has queued (the copy of A)
        (void)hipStreamWaitEvent(mid, e0, 0);
        (void)hipStreamWaitEvent(far, e0, 0);
        if (s.aux) (void)hipStreamWaitEvent(s.aux, e0, 0);
    }
    hipEvent_t ev_mid = nullptr;       // everything the mid stream was given so far
    hipEvent_t ev_far_next = nullptr;  // far finished the columns of the super-panel after the one in flight
    hipEvent_t ev_far_all = nullptr;
    hipEvent_t ev_iprep = nullptr;     // the mid stream's last incremental block row of U for the next super-panel's columns
    // incremental block rows of U for the next super-panel's columns (iprep_columns): 12288 40.4-40.8 -> 39.3-39.9 ms with the 512-column
    // plan; with the large-order plan the far stream delivers those columns too late - the mid stream stalls behind the wait and
    // 16384 goes 69.4 -> 73 ms (docs/EXPERIMENTS.md R5 22)
    static const int iprep_env = std::getenv("RMHIP_LU_IPREP") ? std::atoi(std::getenv("RMHIP_LU_IPREP")) : -1;
    const bool iprep = iprep_env >= 0 ? iprep_env != 0 : !large;
    static const int iprep_split = std::getenv("RMHIP_LU_IPREP_SPLIT") ? std::atoi(std::getenv("RMHIP_LU_IPREP_SPLIT")) : 1;
    int rc = RMHIP_OK;
    const bool verbose = std::getenv("RMHIP_LU_VERBOSE") != nullptr;
    // developer aid (RMHIP_LU_TIMELINE=1): timed events on the main stream around every super-panel boundary, printed after the factorisation
    const bool tl_on = std::getenv("RMHIP_LU_TIMELINE") != nullptr;
    std::vector<std::pair<std::string, hipEvent_t>> tl;
    auto tl_mark = [&](const std::string& what) {
        if (!tl_on) return;
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, main_stream);
        tl.emplace_back(what, e);
    };
    tl_mark("start");
    const auto host_t0 = std::chrono::steady_clock::now();
    auto host_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count(); };
    // columns [c0, c1) right of the complete super-panel [S0, S1) (M = its inverted L11, leading dimension W): its interchanges (unless the
    // caller applied them), T = M A12, the rank-W update with T as the B operand, then T into A12 (U's block row) - on the stream in scope
    auto minv_update_sp = [&](size_t S0, size_t S1, size_t W, const double* M, double* T, size_t c0, size_t c1, bool swaps) -> int {
        if (c1 <= c0) return RMHIP_OK;
        struct WsScope {
            Context* c;
            ~WsScope() {
                c->gemm_split_ws = nullptr;
                c->gemm_split_ws_elems = 0;
            }
        } ws_scope{c};
        c->gemm_split_ws = split_ws[c->stream == main_stream ? 0 : (c->stream == mid ? 1 : 2)]->ptr;
        c->gemm_split_ws_elems = kSplitWsElems;
        if (swaps) RMHIP_TRY(laswp(s, c0, c1, S0, S1));
        const size_t nc = c1 - c0;
        double* A12 = s.A + S0 + c0 * s.lda;
        const size_t keep = c->gemm_split_min_k;
        c->gemm_split_min_k = 1024;  // few output tiles and k = W: split the inner dimension (one block alone walks 16 k per us)
        int r;
        const size_t h = (W / 2 / 128) * 128;
        if (nc >= 1024 && h >= 256) {  // M is lower triangular: the upper half of T only needs the leading h x h block
            r = lu_dgemm(c, h, nc, h, 1.0, M, W, A12, s.lda, 0.0, T, W);
            if (r == RMHIP_OK) r = lu_dgemm(c, W - h, nc, W, 1.0, M + h, W, A12, s.lda, 0.0, T + h, W);
        } else {
            r = lu_dgemm(c, W, nc, W, 1.0, M, W, A12, s.lda, 0.0, T, W);
        }
        c->gemm_split_min_k = keep;
        RMHIP_TRY(r);
        if (S1 < s.rows) RMHIP_TRY(lu_dgemm(c, s.rows - S1, nc, W, -1.0, s.A + S1 + S0 * s.lda, s.lda, T, W, 1.0, s.A + S1 + c0 * s.lda, s.lda));
        RMHIP_HIP_CHECK(hipMemcpy2DAsync(A12, s.lda * sizeof(double), T, W * sizeof(double), W * sizeof(double), nc, hipMemcpyDeviceToDevice, c->stream));
        return RMHIP_OK;
    };
    // The mid stream's share of a boundary beyond the next super-panel's first two inner panels is issued LATER - behind that
    // super-panel's first inner update of its second panel's columns (see the boundary code): the main stream's look-ahead after the
    // second inner panel then waits for two small launches instead of the deep update of 1280 columns (round 5: 1-4 ms per super-panel)
    std::function<int()> mid_deferred;
    for (size_t J = 0; J < plan.size() && rc == RMHIP_OK; ++J) {
        const size_t S0 = plan[J].s0, S1 = plan[J].s1, nbJ = plan[J].nb, W = S1 - S0;
        if (verbose && (W > nbJ || J + 1 == plan.size())) std::fprintf(stderr, "[lu] host %.2f ms: super-panel %zu [%zu, %zu) nb %zu queued so far %llu launches\n", host_ms(), J, S0, S1, nbJ, (unsigned long long)c->tel.kernel_launches);
        const size_t S1n = J + 1 < plan.size() ? plan[J + 1].s1 : S1;
        const size_t S1nn = J + 2 < plan.size() ? plan[J + 2].s1 : S1n;
        const bool multi = W > nbJ;
        const bool use_minv = minv_on && multi && !iprep && W <= Wmax;
        double* const minv = use_minv ? minv_mem[J & 1]->ptr : nullptr;
        const size_t ldm = W;
        // row block [j - S0, j - S0 + w) of M = L11^-1, on the stream in scope (mid), once panel [j, j + w)'s interchanges are in columns [S0, j)
        auto minv_extend = [&](size_t j, size_t w) -> int {
            const size_t off = j - S0;
            double* Mpp = minv + off + off * ldm;
            if (off == 0) {
                if (ev_minv_read[J & 1]) (void)hipStreamWaitEvent(c->stream, ev_minv_read[J & 1], 0);  // far may still read this buffer (boundary J - 2)
                RMHIP_HIP_CHECK(hipMemsetAsync(minv, 0, sizeof(double) * W * W, c->stream));
            }
            hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((w * w + 255) / 256)), dim3(256), 0, c->stream, Mpp, ldm, (unsigned)w);
            RMHIP_TRY(launch_check(c));
            RMHIP_TRY(trsm_lower_rec(c, s.A + j + j * s.lda, s.lda, w, Mpp, ldm, w));
            if (off) {
                RMHIP_TRY(lu_dgemm(c, w, off, off, 1.0, s.A + j + S0 * s.lda, s.lda, minv, ldm, 0.0, mtmp_mem->ptr, w));
                RMHIP_TRY(lu_dgemm(c, w, off, w, -1.0, Mpp, ldm, mtmp_mem->ptr, w, 0.0, minv + off, ldm));
            }
            return RMHIP_OK;
        };
        auto minv_update = [&, S0, S1, W, minv](double* T, size_t c0, size_t c1, bool swaps) -> int {
            return minv_update_sp(S0, S1, W, minv, T, c0, c1, swaps);
        };
        s.panel_pad_kb = top_pad_env >= 0 ? top_pad_env : ((kmin - S0 > super_rows) ? super_panel_pad : -1);
        for (size_t j = S0; j < S1 && rc == RMHIP_OK;) {
            const size_t w = (S1 - j) < nbJ ? (S1 - j) : nbJ;
            rc = getrf_rec(s, j, w);  // P_j on main
            if (rc != RMHIP_OK) break;
            if (tl_on && W > nbJ) tl_mark("J" + std::to_string(J) + " panel " + std::to_string(j) + " done");
            hipEvent_t panel_done = record(main_stream);
            const size_t next = j + w;
            const bool boundary = next == S1;
            size_t la_w = 0;
            if (next < kmin) {
                if (!boundary) la_w = (S1 - next) < nbJ ? (S1 - next) : nbJ;
                else la_w = (plan[J + 1].s1 - next) < plan[J + 1].nb ? (plan[J + 1].s1 - next) : plan[J + 1].nb;
            }
            const size_t t0 = next + la_w;
            if (!boundary) {
                if (ev_mid) (void)hipStreamWaitEvent(main_stream, ev_mid, 0);
                rc = update_columns(s, j, w, next, t0);  // LA_j on main
                if (rc != RMHIP_OK) break;
                (void)hipStreamWaitEvent(mid, panel_done, 0);
                {
                    StreamScope scope(c, mid, mid_pad);
                    if (mid_deferred) {
                        // first inner panel after a boundary: the next panel's columns first (what the main stream's next look-ahead
                        // needs), then the boundary's remaining columns, then this panel's update of those
                        const size_t ts = t0 + nbJ < S1 ? t0 + nbJ : S1;
                        rc = update_columns(s, j, w, t0, ts);
                        ev_mid = record(mid);
                        if (rc == RMHIP_OK) rc = mid_deferred();
                        mid_deferred = nullptr;
                        if (rc == RMHIP_OK) rc = update_columns(s, j, w, ts, S1);
                    } else {
                        rc = update_columns(s, j, w, t0, S1);
                        if (rc == RMHIP_OK && j > S0) rc = laswp(s, S0, j, j, j + w);  // the super-panel's own left columns
                        ev_mid = record(mid);
                    }
                    if (rc == RMHIP_OK && use_minv && !minv_ext_far) rc = minv_extend(j, w);  // (behind the event: the main stream does not wait for it)
                }
                if (rc == RMHIP_OK && use_minv && minv_ext_far) {
                    (void)hipStreamWaitEvent(far, ev_mid, 0);  // (this panel's interchanges are in the columns left of it)
                    StreamScope scope(c, far, far_pad);
                    rc = minv_extend(j, w);
                }
                if (rc != RMHIP_OK) break;
                if (iprep && S1 < S1n) {
                    // the next super-panel's columns receive this panel's block row of U now (not at the boundary): they are complete up
                    // to the previous super-panel once far's first update of boundary J - 1 is in
                    // (on a stream of their own instead: 76-79 ms at n = 16384)
                    if (ev_far_next) (void)hipStreamWaitEvent(mid, ev_far_next, 0);
                    {
                        StreamScope scope(c, mid, mid_pad);
                        rc = iprep_columns(s, S0, j, w, S1, S1n);
                    }
                    if (rc != RMHIP_OK) break;
                    ev_iprep = record(mid);
                }
            } else {
                // the super-panel is complete: its last panel's interchanges reach its left columns first - everything below reads L21
                // of the whole super-panel
                hipEvent_t ev_left = panel_done;
                if (mid_deferred) {  // (a super-panel of a single panel: nobody consumed the previous boundary's second piece)
                    StreamScope scope(c, mid, mid_pad);
                    rc = mid_deferred();
                    mid_deferred = nullptr;
                    if (rc != RMHIP_OK) break;
                    ev_mid = record(mid);
                }
                if (multi && j > S0) {
                    (void)hipStreamWaitEvent(mid, panel_done, 0);
                    {
                        StreamScope scope(c, mid, mid_pad);
                        rc = laswp(s, S0, j, j, j + w);
                        if (rc == RMHIP_OK) {
                            ev_left = record(mid);
                            ev_mid = ev_left;
                        }
                        if (rc == RMHIP_OK && use_minv && !minv_ext_far) {  // the last row block of the inverse, then its largest entry for the guard
                            rc = minv_extend(j, w);
                            if (rc == RMHIP_OK) {
                                hipLaunchKernelGGL(k_absmax_word, dim3(256), dim3(256), 0, mid, (const double*)minv, ldm, (unsigned)W, (unsigned)W, (pk_u64*)s.minv_max);
                                rc = launch_check(c);
                            }
                            ev_minv = record(mid);
                        }
                    }
                    if (rc == RMHIP_OK && use_minv && minv_ext_far) {
                        (void)hipStreamWaitEvent(far, ev_left, 0);
                        StreamScope scope(c, far, far_pad);
                        rc = minv_extend(j, w);
                        if (rc == RMHIP_OK) {
                            hipLaunchKernelGGL(k_absmax_word, dim3(256), dim3(256), 0, far, (const double*)minv, ldm, (unsigned)W, (unsigned)W, (pk_u64*)s.minv_max);
                            rc = launch_check(c);
                        }
                        ev_minv = record(far);
                    }
                    if (rc != RMHIP_OK) break;
                }
                if (la_w) {
                    tl_mark("J" + std::to_string(J) + " chain done");
                    if (ev_mid) (void)hipStreamWaitEvent(main_stream, ev_mid, 0);
                    tl_mark("J" + std::to_string(J) + " mid ready");
                    if (ev_far_next) (void)hipStreamWaitEvent(main_stream, ev_far_next, 0);
                    tl_mark("J" + std::to_string(J) + " far ready");
                    if (use_minv && (minv_who & 1)) {
                        (void)hipStreamWaitEvent(main_stream, ev_minv, 0);
                        rc = minv_update(t_main_mem->ptr, next, t0, true);
                    } else if (iprep && multi) {
                        // the columns hold every block row of U but the last panel's: that one, then the deep update alone
                        if (ev_iprep) (void)hipStreamWaitEvent(main_stream, ev_iprep, 0);
                        rc = iprep_columns(s, S0, j, w, next, t0);
                        if (rc == RMHIP_OK) rc = gemm_columns(s, S0, W, next, t0);
                    } else {
                        rc = update_columns(s, S0, W, next, t0);  // rank-W look-ahead update on main
                    }
                    if (rc != RMHIP_OK) break;
                    tl_mark("J" + std::to_string(J) + " LA done");
                }
                if (t0 < S1n) {
                    (void)hipStreamWaitEvent(mid, panel_done, 0);
                    if (ev_far_next) (void)hipStreamWaitEvent(mid, ev_far_next, 0);
                    {
                        StreamScope scope(c, mid, mid_pad);
                        if (use_minv && (minv_who & 2)) {
                            (void)hipStreamWaitEvent(mid, ev_minv, 0);
                            // The second inner panel's columns first: the main stream's next look-ahead
                            // update waits for these only
                            static const int defer_on = std::getenv("RMHIP_LU_MINV_DEFER") ? std::atoi(std::getenv("RMHIP_LU_MINV_DEFER")) : 1;
                            const size_t nbn = plan[J + 1].nb, t1 = t0 + 2 * nbn < S1n ? t0 + 2 * nbn : S1n;
                            rc = minv_update(t_mid_mem->ptr, t0, t1, true);  // the second and third inner panels' columns now
                            if (rc == RMHIP_OK) ev_mid = record(mid);
                            if (rc == RMHIP_OK && t1 < S1n) {
                                double* const Tm = t_mid_mem->ptr;
                                const double* const Mc = minv;
                                const size_t S0c = S0, S1c = S1, Wc = W, c1c = S1n;
                                auto rest = [&minv_update_sp, S0c, S1c, Wc, Mc, Tm, t1, c1c]() { return minv_update_sp(S0c, S1c, Wc, Mc, Tm, t1, c1c, true); };
                                if (defer_on && plan[J + 1].s1 - plan[J + 1].s0 > plan[J + 1].nb) mid_deferred = rest;  // the rest behind the next panel's first update
                                else rc = rest();
                            }
                        } else if (iprep && multi) {
                            rc = iprep_columns(s, S0, j, w, t0, S1n);
                            // the second panel's columns first: the main stream's next look-ahead update waits for these only
                            const size_t nbn = plan[J + 1].nb, t1 = (iprep_split && t0 + nbn < S1n) ? t0 + nbn : S1n;
                            if (rc == RMHIP_OK) rc = gemm_columns(s, S0, W, t0, t1);
                            if (rc == RMHIP_OK && t1 < S1n) {
                                ev_mid = record(mid);
                                rc = gemm_columns(s, S0, W, t1, S1n);
                            } else if (rc == RMHIP_OK) {
                                ev_mid = record(mid);
                            }
                        } else {
                            rc = update_columns(s, S0, W, t0, S1n);
                            if (rc == RMHIP_OK) ev_mid = record(mid);
                        }
                    }
                    if (rc != RMHIP_OK) break;
                    ev_iprep = nullptr;
                }
                // far: everything right of the next super-panel - interchange, W-wide solve, rank-W update (the next super-panel's columns
                // first: event) - then the interchanges of the columns left of this super-panel.
                // (Tried: the interchange + solve on a stream of their own, pipelined against the updates over two column parts as in
                // getrf_blocked - 100 ms against 73 with four HIP streams active, 71.2 against 73.0 when GPU_MAX_HW_QUEUES=2 makes them
                // share hardware queues; on the mid stream 75.5.  Deferring far updates under a per-boundary flop budget - identical
                // factors - 84.5 / 75.4 / 73.3 ms at budgets of 1 / 2 / 3 chain times against 73.4 eager.  docs/EXPERIMENTS.md.)
                (void)hipStreamWaitEvent(far, ev_left, 0);
                {
                    StreamScope scope(c, far, far_pad);
                    ev_far_next = nullptr;
                    if (S1n < s.cols) {
                        const size_t cn = S1nn < s.cols ? S1nn : s.cols;
                        // (tried: interchange + solve + update of the next super-panel's columns first, then the rest's - a second
                        // 31-launch solve per boundary on this stream: 69.4 -> 73.8 ms; the far stream is the bottleneck of the first phase)
                        if (use_minv && (minv_who & 4)) {
                            // one interchange pass, then per column chunk T = M A12, the deep update, T -> A12; the next super-panel's
                            // columns are the first chunk (the event no longer waits for the solve of ALL columns)
                            (void)hipStreamWaitEvent(far, ev_minv, 0);
                            rc = laswp(s, S1n, s.cols, S0, S1);
                            if (rc == RMHIP_OK && cn > S1n) {
                                rc = minv_update(t_far_mem->ptr, S1n, cn, false);
                                ev_far_next = record(far);
                            }
                            for (size_t q0 = cn; q0 < s.cols && rc == RMHIP_OK; q0 += far_chunk)
                                rc = minv_update(t_far_mem->ptr, q0, q0 + far_chunk < s.cols ? q0 + far_chunk : s.cols, false);
                            ev_minv_read[J & 1] = record(far);
                        } else {
                        rc = prep_columns(s, S0, W, S1n, s.cols);
                        if (rc == RMHIP_OK && cn > S1n) {
                            rc = gemm_columns(s, S0, W, S1n, cn);
                            ev_far_next = record(far);
                        }
                        if (rc == RMHIP_OK) rc = gemm_columns(s, S0, W, cn, s.cols);
                        }
                    }
                    if (rc == RMHIP_OK && S0 > 0) rc = laswp(s, 0, S0, S0, S1);  // columns left of the super-panel: all of its interchanges at once
                }
                ev_far_all = record(far);
            }
            j = next;
        }
    }
    s.panel_pad_kb = -1;
    if (ev_far_all) (void)hipStreamWaitEvent(main_stream, ev_far_all, 0);
    if (ev_mid) (void)hipStreamWaitEvent(main_stream, ev_mid, 0);
    if (tl_on) {
        (void)hipStreamSynchronize(main_stream);
        for (size_t i = 1; i < tl.size(); ++i) {
            float ms0 = 0.f, ms1 = 0.f;
            (void)hipEventElapsedTime(&ms0, tl[0].second, tl[i].second);
            (void)hipEventElapsedTime(&ms1, tl[i - 1].second, tl[i].second);
            std::fprintf(stderr, "[lu timeline] %8.2f ms (+%6.2f) %s\n", ms0, ms1, tl[i].first.c_str());
        }
        for (auto& kv : tl) (void)hipEventDestroy(kv.second);
    }
    if (verbose) std::fprintf(stderr, "[lu] host %.2f ms: everything queued (%llu launches)\n", host_ms(), (unsigned long long)c->tel.kernel_launches);
    (void)hipStreamSynchronize(mid);
    (void)hipStreamSynchronize(far);
    (void)hipStreamSynchronize(main_stream);
    return rc;
}

// In-place LU of A (rows x cols, lda). perm_dev[rows] receives the row permutation as the
// reference reports it (perm[k] = original row now at position k, host_lu.rs:50,107).
// *info_host = number of pivots that hit the singular cut-off.
// mode 0: the reference's pivot sequence (host_lu.rs:37-59: grid-wide first maximum) - `lu`, and the fallback of the solve path.
// mode 1: solve path (pivots unobservable): pivoting restricted to each panel's top block, multipliers checked against tau
//         (k_rp_below); RMHIP_LU_GROWTH when the check fails - the matrix is clobbered, the caller refactors a fresh copy in mode 0.
__global__ void __launch_bounds__(256) k_ipiv_to_f64(const int* __restrict__ ipiv, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)ipiv[i];
}

// deferred form (deferred_guard != nullptr): the factorisation's status words folded into the caller's running guard value on the
// device - NaN once a pivot hit the singular cut-off, else the larger of the old value and this panel's largest multiplier
__global__ void k_lu_fold_status(const int* __restrict__ info, const unsigned long long* __restrict__ growth_bits, double* __restrict__ guard) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double g = __longlong_as_double((long long)*growth_bits);
    const double cur = *guard;
    double out = (cur != cur || g != g) ? __longlong_as_double(0x7ff8000000000000LL) : (g > cur ? g : cur);
    if (*info > 0) out = __longlong_as_double(0x7ff8000000000000LL);
    *guard = out;
}

int lu_factor_device(Context* c, double* A, size_t rows, size_t cols, size_t lda, int* perm_dev, int* info_host,
                     std::vector<int>* ipiv_host, int mode, double* ipiv_dev_f64, double* deferred_guard) {
    const size_t kmin = rows < cols ? rows : cols;
    if (rows > 0x7fffffffULL || cols > 0x7fffffffULL) return fail(RMHIP_ERR_UNSUPPORTED, "lu: dimension exceeds 2^31");
    c->lu_used_one_xcd = false;
    {
        static std::once_flag prio_once;  // (one device per process: rmhip_init binds the context to its device)
        std::call_once(prio_once, [] {
            const int p = std::getenv("RMHIP_LU_CHAIN_PRIO") ? std::atoi(std::getenv("RMHIP_LU_CHAIN_PRIO")) : 1;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(d_chain_prio), &p, sizeof(int));
        });
    }
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
    const size_t off_xvals = (off_row + sizeof(int) * 2 * MAX_PANEL_BLOCKS + 15) & ~(size_t)15;
    const size_t off_xa = off_xvals + sizeof(unsigned long long) * 2 * PK_MAXB * BASE_W * 2;
    const size_t off_xb = off_xa + sizeof(unsigned long long) * 2 * PK_MAXB;
    const size_t off_xctl = off_xb + sizeof(unsigned long long) * 2 * PK_MAXB;
    const size_t off_ucomp = off_xctl + 64 + 16 * sizeof(unsigned long long);
    const size_t off_linv = off_ucomp + sizeof(double) * UCOMP_STRIDE * kUcompSlots;
    const size_t total = off_linv + sizeof(double) * 256 * (rows / 16 + 8);
    std::shared_ptr<Allocation> blk_mem;  // pooled: a hipMalloc / hipFree pair costs two device synchronisations per factorisation
    RMHIP_TRY(c->alloc_device(total / sizeof(double) + 2, &blk_mem));
    char* blk = (char*)blk_mem->ptr;
    int* ipiv = (int*)blk;
    int* info = ipiv + rows;
    hipError_t e = hipMemsetAsync(blk, 0, total, c->stream);
    if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    std::vector<size_t> panel_start;
    panel_start.reserve(max_panels);
    LuState s{c, A, rows, cols, lda, ipiv, info, (int*)(blk + off_posof), (int*)(blk + off_rowat), (int*)(blk + off_prow),
              (int2*)(blk + off_plist), &panel_start, (double*)(blk + off_abs), (int*)(blk + off_pos), (int*)(blk + off_row),
              (int*)(blk + off_xctl + 16), (unsigned long long*)(blk + off_xa), (unsigned long long*)(blk + off_xvals), 0u, true,
              nullptr};
    {
        // persistent panels need all their blocks co-resident (18.5 KiB of LDS and 270 VGPRs per 4-wave block: at most one
        // per CU beside a dgemm block, so up to num_cus blocks)
        const char* pm = std::getenv("RMHIP_LU_PANEL");  // "columns" selects the one-launch-per-column kernels
        if ((pm && pm[0] == 'c') || c->lu_conservative) s.persistent = false;
        const char* dbgenv = std::getenv("RMHIP_LU_PANEL_DEBUG");
        if (dbgenv && dbgenv[0] == '1') s.xdbg = (unsigned long long*)(blk + off_xctl + 64);
        // solve path: RMHIP_LU_FAST=0 keeps the grid-wide pivot rule everywhere; RMHIP_LU_TAU sets the multiplier bound (read per call: tests)
        const char* fe = std::getenv("RMHIP_LU_FAST");
        s.fast = mode == 1 && !(fe && fe[0] == '0') && s.persistent;
        c->lu_last_fast = s.fast;  // the caller counts solve-path factorisations on what actually ran (rmhip_lu_stats)
        if (const char* tv = std::getenv("RMHIP_LU_TAU")) s.tau = std::atof(tv);
        c->lu_tau = s.tau;
        s.ucomp = (double*)(blk + off_ucomp);
        s.growth = (unsigned long long*)(blk + off_xctl + 32);
        s.minv_max = (unsigned long long*)(blk + off_xctl + 40);
        if (deferred_guard) s.screened = true;  // no early read of the first panel's multipliers: the caller's guard sees them at the end
        // a driver that runs updates on a stream of its own beside this factorisation (csrc/sharded.cpp) lends its yield table: k_rp_top
        // counts itself in and that stream's eight-wave blocks on its CU pause (the two-level driver below installs its own)
        if (c->ext_yield_tab) s.yield_word = c->ext_yield_tab;
        // matrix-core triangular solves with k_rp_top's inverted diagonal blocks: every base panel must be 64 columns wide
        static const int trsm_mfma = std::getenv("RMHIP_LU_TRSM_MFMA") ? std::atoi(std::getenv("RMHIP_LU_TRSM_MFMA")) : 1;
        if (s.fast && trsm_mfma && !s.xdbg) s.linv = (double*)(blk + off_linv);
    }
    std::vector<unsigned char> linv_ok(rows / 16 + 8, 0);
    struct LinvScope {  // the solves find the inverses through the context while this factorisation runs
        Context* c;
        ~LinvScope() {
            c->lu_linv = nullptr;
            c->lu_linv_ok = nullptr;
            c->lu_work = nullptr;
            c->lu_work_ld = 0;
        }
    } linv_scope{c};
    if (s.linv) {
        c->lu_linv = s.linv;
        c->lu_linv_ok = &linv_ok;
        c->lu_work = A;
        c->lu_work_ld = lda;
    }
    // Panel width of the look-ahead driver and its threshold, from an interleaved sweep (scripts/lu_knobs.py, ms for
    // x = A\b): n = 6144: 38.7 without look-ahead, 33.6 with nb 128, 34.9 with 256, 36.8 with 512; 8192: 51.2 (nb 512)
    // / 48.4 (256) / 47.2 (128); 10240: 67.3 / 63.4 / 62.2; 12288: 84.7 / 80.6 / 81.2; 16384: 130.8 / 130.7 / -;
    // 5120: 31.6 without, 27.9 with; 4096: 22.2 without, 22.6-24 with.
    size_t nb = kmin < 12288 ? 128 : 256;
    if (const char* v = std::getenv("RMHIP_LU_NB")) nb = (size_t)std::atoll(v);
    nb = nb < 64 ? 64 : (nb / 64) * 64;
    // Look-ahead (second stream): default from kmin = 5120; RMHIP_LU_LOOKAHEAD=1 forces it for every kmin > nb, =0
    // disables it.  It needs the persistent panels: with one launch per column the panel kernels wait behind the
    // update's dgemm blocks.
    // Solve path (one-workgroup panels: nothing has to be co-resident, and the chain is all there is at these sizes): from 1152 -
    // n = 1280 3.30 -> 2.52 ms, 1536 3.63 -> 2.95, 2048 4.30 -> 3.91, 3072 8.06 -> 5.98, 4096 9.92 -> 8.17 (5120: 16.9 without,
    // 10.6 with; 1024: 2.03 / 2.00).
    const char* la = std::getenv("RMHIP_LU_LOOKAHEAD");
    bool blocked = kmin >= (s.fast ? 1152u : 5120u) && kmin > nb;
    if (la) blocked = kmin > nb && la[0] == '1';
    if (!s.persistent) blocked = false;
    // Under look-ahead the panels keep 256-row blocks (132 KiB of LDS: a whole CU).  The update stream's dgemm runs one
    // block per CU (84 KiB) at low priority, so a CU is empty whenever its dgemm block retires and the waiting panel
    // block (main stream, higher priority) takes it: all blocks are resident after about one dgemm-block time
    // (~55 us) and the bounded spins absorb that.  128-row blocks (66 KiB) fit BESIDE a dgemm block and start at
    // once, but twice as many blocks make every exchange slower: measured 145.8 ms against 129.0 ms at n = 16384
    // (RMHIP_LU_PANEL_ROWS=128 selects them).
    // two-level driver (super-panels) on the solve path from kmin = 8192 (RMHIP_LU_SUPER=0: the one-level driver; RMHIP_LU_SUPER_MIN)
    static const int super_on = std::getenv("RMHIP_LU_SUPER") ? std::atoi(std::getenv("RMHIP_LU_SUPER")) : 1;
    static const size_t super_min = std::getenv("RMHIP_LU_SUPER_MIN") ? (size_t)std::atoll(std::getenv("RMHIP_LU_SUPER_MIN")) : 8192;
    const bool super = blocked && s.fast && super_on && kmin >= super_min;
    int rc = super ? getrf_super(s, kmin) : (blocked ? getrf_blocked(s, kmin, nb) : getrf_rec(s, 0, kmin));
    if (rc == RMHIP_OK && cols > rows && !blocked) {  // wide: finish U's right block (the blocked driver covers it)
        rc = laswp(s, rows, cols, 0, rows);
        if (rc == RMHIP_OK) rc = trsm_lower_rec(c, A, lda, rows, A + rows * lda, lda, cols - rows);
    }
    if (rc == RMHIP_OK && deferred_guard && s.fast && !s.xdbg) {
        // Deferred checks (a panel of a larger factorisation whose driver has a guard of its own - csrc/sharded.cpp): nothing is read
        // back; the interchange vector goes out as a device tensor, the status words into the caller's guard value.  Everything above
        // and these two launches are queued on the context's stream: the workspace block goes back to the pool stream-ordered.
        if (ipiv_dev_f64 && kmin) {
            hipLaunchKernelGGL(k_ipiv_to_f64, dim3((unsigned)((kmin + 255) / 256)), dim3(256), 0, c->stream, (const int*)ipiv, ipiv_dev_f64, (int)kmin);
            RMHIP_TRY(launch_check(c));
        }
        hipLaunchKernelGGL(k_lu_fold_status, dim3(1), dim3(64), 0, c->stream, (const int*)info, (const unsigned long long*)s.growth, deferred_guard);
        RMHIP_TRY(launch_check(c));
        if (info_host) *info_host = 0;
        return RMHIP_OK;
    }
    std::vector<int> h_ipiv(rows + 1, 0);
    if (rc == RMHIP_OK) {
        int h_xerr = 0;
        unsigned long long h_growth = 0, h_minv = 0;
        e = hipMemcpyAsync(h_ipiv.data(), ipiv, sizeof(int) * (rows + 1), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_xerr, s.xerr, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && s.fast) e = hipMemcpyAsync(&h_growth, s.growth, sizeof(h_growth), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && s.fast) e = hipMemcpyAsync(&h_minv, s.minv_max, sizeof(h_minv), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "lu: reading pivots: %s", hipGetErrorString(e));
        if (e == hipSuccess && s.fast) {
            double gmax;
            std::memcpy(&gmax, &h_growth, sizeof(gmax));
            c->lu_last_growth = gmax;
            // a multiplier beyond tau, a NaN, or a pivot at the singular cut-off inside a top block (a larger entry may lie below it)
            double minv_abs;
            std::memcpy(&minv_abs, &h_minv, sizeof(minv_abs));
            c->lu_last_minv = minv_abs;
            static const double minv_limit = std::getenv("RMHIP_LU_MINV_MAX") ? std::atof(std::getenv("RMHIP_LU_MINV_MAX")) : 1.0e6;
            if (!(minv_abs <= minv_limit)) {  // an ill-conditioned L11: the products with its inverse are not to be trusted (also NaN)
                if (std::getenv("RMHIP_LU_VERBOSE"))
                    std::fprintf(stderr, "[lu] solve path: largest entry of an inverted L11 block %.3g (limit %.3g): refactoring with the grid-wide rule\n", minv_abs, minv_limit);
                return RMHIP_LU_GROWTH;
            }
            if (!(gmax <= s.tau) || h_ipiv[rows] > 0 || std::getenv("RMHIP_LU_TEST_GROWTH")) {
                if (std::getenv("RMHIP_LU_VERBOSE"))
                    std::fprintf(stderr, "[lu] solve path: max multiplier %.3g (tau %.3g), %d small pivot(s): refactoring with the grid-wide rule\n", gmax,
                                 s.tau, h_ipiv[rows]);
                return RMHIP_LU_GROWTH;
            }
        }
        if (!h_xerr && s.persistent && std::getenv("RMHIP_LU_TEST_RETRY")) h_xerr = 1;  // test hook for the retry path
        if (e == hipSuccess && h_xerr) {
            c->lu_exchange_timeouts++;
            // bounded spins expired: the panel workgroups were not co-resident (device shared with another
            // context?).  The matrix is clobbered; callers holding the original refactor it conservatively.
            // A factorisation that placed panels on one XCD first gives that up (if the workgroup -> XCD assignment is not
            // the expected round robin the plain stores of the exchange are never seen); only a failure of the spread
            // placement makes the context conservative.
            if (std::getenv("RMHIP_LU_VERBOSE"))
                std::fprintf(stderr, "[lu] panel exchange timed out (one-XCD placement used: %d, allowed: %d)\n", (int)c->lu_used_one_xcd,
                             (int)c->one_xcd_ok);
            if (c->lu_used_one_xcd && c->one_xcd_ok && !std::getenv("RMHIP_LU_TEST_RETRY")) c->one_xcd_ok = false;
            else c->lu_conservative = true;
            (void)fail(RMHIP_ERR_HIP, "lu: panel workgroups were not co-resident (device shared?)");
            rc = RMHIP_LU_RETRY;
        }
        if (s.xdbg) {
            unsigned long long h[16];
            if (hipMemcpy(h, s.xdbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                static const char* names_grid[16] = {"load", "argmax+park+sync", "publish", "wait+fold", "winner vals", "sync",
                                                     "elim under exch", "park+shift", "write back", "-", "bookkeep+div", "-",
                                                     "-", "-", "-", "-"};
                static const char* names_top[16] = {"load", "candidate", "barrier", "winner+window", "park", "barrier", "g solve", "rank-8",
                                                    "store+shift", "below: stage U", "below: load", "below: triangle", "below: stores", "below: rank-8",
                                                    "below: tail", "-"};
                const char* const* names = s.fast ? names_top : names_grid;
                for (int i = 0; i < 15; ++i)
                    std::fprintf(stderr, "[lu panel] %-12s %10.1f us total  %7.3f us/column\n", names[i], h[i] * 0.01,
                                 kmin ? h[i] * 0.01 / (double)kmin : 0.0);
            }
        }
    } else {
        (void)hipStreamSynchronize(c->stream);
    }
    if (rc != RMHIP_OK) return rc;
    if (info_host) *info_host = h_ipiv[rows];
    if (ipiv_host) ipiv_host->assign(h_ipiv.begin(), h_ipiv.begin() + (long)kmin);
    if (ipiv_dev_f64 && kmin) {  // the interchange targets as a device tensor (rmhip_blk_lu): no trip through the host
        hipLaunchKernelGGL(k_ipiv_to_f64, dim3((unsigned)((kmin + 255) / 256)), dim3(256), 0, c->stream, (const int*)ipiv, ipiv_dev_f64, (int)kmin);
        RMHIP_TRY(launch_check(c));
    }
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

// The same from a DEVICE interchange vector (doubles, as rmhip_blk_lu returns them), nothing crossing to the host: one workgroup
// composes the swaps k <-> ipiv[k] in an LDS map of the view's rows (thread 0 walks them in order - at most a few hundred), every thread
// then reports the positions whose content changed as (dst, src) moves; k_permute_rows_dev applies the list.  Entries outside
// [0, nrows) are ignored and counted in *bad (the caller's guard reads it with the factorisation's other status words).
__global__ void __launch_bounds__(256) k_compose_swaps(const double* __restrict__ ipiv, int npiv, int nrows, int2* __restrict__ list,
                                                      int* __restrict__ len_bad, int cap) {
    extern __shared__ int s_map[];  // [nrows] map, then [npiv] interchange targets (-1: out of range)
    __shared__ int s_len, s_bad;
    int* s_piv = s_map + nrows;
    for (int i = threadIdx.x; i < nrows; i += 256) s_map[i] = i;
    if (threadIdx.x == 0) s_len = s_bad = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < npiv; k += 256) {
        const double pv = ipiv[k];
        const bool ok = pv >= 0.0 && pv < (double)nrows && k < nrows;
        s_piv[k] = ok ? (int)pv : -1;
        if (!ok) atomicAdd(&s_bad, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < npiv; ++k) {
            const int p = s_piv[k];
            if (p < 0 || p == k) continue;
            const int a = s_map[k], b = s_map[p];
            s_map[k] = b;
            s_map[p] = a;
        }
        len_bad[1] = s_bad;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nrows; i += 256) {
        const int src = s_map[i];
        if (src != i) {
            const int slot = atomicAdd(&s_len, 1);
            if (slot < cap) list[slot] = make_int2(i, src);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) len_bad[0] = s_len < cap ? s_len : cap;
}
__global__ void __launch_bounds__(256) k_permute_rows_dev(double* __restrict__ A, size_t lda, size_t ncols, const int2* __restrict__ list,
                                                         const int* __restrict__ len_ptr) {
    const size_t cc = blockIdx.x;
    if (cc >= ncols) return;
    const int len = *len_ptr;
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
// (rows up to 36864: the map must fit the CU's LDS; the caller falls back to the host composition beyond)
static constexpr size_t kSwapDevMaxRows = 36864;
int lu_swap_rows_from_device(Context* c, double* A, size_t lda, size_t nrows, size_t ncols, const double* ipiv_dev, size_t npiv) {
    if (ncols == 0 || npiv == 0 || nrows == 0) return RMHIP_OK;
    if (nrows > kSwapDevMaxRows || npiv > (size_t)128 * PERM_PER_THREAD) return RMHIP_ERR_UNSUPPORTED;  // (not an error: the caller takes the host path)
    const int cap = 256 * PERM_PER_THREAD;
    std::shared_ptr<Allocation> ws;  // [cap] moves + (len, bad); stream-ordered reuse: everything below runs on c->stream
    RMHIP_TRY(c->alloc_device((size_t)cap + 2, &ws));
    int2* list = (int2*)ws->ptr;
    int* len_bad = (int*)(list + cap);
    const size_t lds = (nrows + npiv) * sizeof(int);
    if (lds > 65536) c->ensure_max_lds((const void*)k_compose_swaps, (kSwapDevMaxRows + 1024) * sizeof(int));
    hipLaunchKernelGGL(k_compose_swaps, dim3(1), dim3(256), lds, c->stream, ipiv_dev, (int)npiv, (int)nrows, list, len_bad, cap);
    RMHIP_TRY(launch_check(c));
    for (size_t c0 = 0; c0 < ncols; c0 += 65535) {
        const size_t nc = (ncols - c0) < 65535 ? (ncols - c0) : 65535;
        hipLaunchKernelGGL(k_permute_rows_dev, dim3((unsigned)nc), dim3(256), 0, c->stream, A + c0 * lda, lda, nc, (const int2*)list, (const int*)len_bad);
        RMHIP_TRY(launch_check(c));
    }
    return RMHIP_OK;
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
template <class T>
__global__ void __launch_bounds__(256) k_transpose(const T* __restrict__ src, size_t lds_, size_t rows, size_t cols,
                                                   T* __restrict__ dst, size_t ldd) {
    __shared__ T tile[TR_TILE][TR_TILE + 1];
    const size_t r0 = (size_t)blockIdx.x * TR_TILE, c0 = (size_t)blockIdx.y * TR_TILE;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    // all 16 loads first, on indices clamped into the matrix (a guarded loop waits for each load before it issues the next: the same
    // tile in tensor_ops.hip went from 268 to 217 us at 8192^2 with this), the stores stay guarded
    const size_t rl = r0 + lane < rows ? r0 + lane : rows - 1;
    T v[TR_TILE / 4];
#pragma unroll
    for (int p = 0; p < TR_TILE / 4; ++p) {
        const size_t cc = c0 + grp + 4 * p < cols ? c0 + grp + 4 * p : cols - 1;
        v[p] = src[rl + cc * lds_];
    }
#pragma unroll
    for (int p = 0; p < TR_TILE / 4; ++p) tile[grp + 4 * p][lane] = v[p];
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
    hipLaunchKernelGGL(k_transpose<double>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, src, lds_, rows, cols, dst, ldd);
    return launch_check(c);
}

int transpose_device_f32(Context* c, const float* src, size_t lds_, size_t rows, size_t cols, float* dst, size_t ldd) {
    if (rows == 0 || cols == 0) return RMHIP_OK;
    const size_t gx = (rows + TR_TILE - 1) / TR_TILE, gy = (cols + TR_TILE - 1) / TR_TILE;
    if (gy > 65535) return fail(RMHIP_ERR_UNSUPPORTED, "transpose: more than %d columns", 65535 * TR_TILE);
    hipLaunchKernelGGL(k_transpose<float>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, src, lds_, rows, cols, dst, ldd);
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
    std::shared_ptr<Allocation> mem;  // from the context's pool (a hipMalloc / hipFree pair per call cost more than the solve of a small system)
    RMHIP_TRY(c->alloc_device(4, &mem));
    double* d = mem->ptr;
    hipLaunchKernelGGL(k_diag_stats, dim3(1), dim3(256), 0, c->stream, A, lda, n, d);
    double h[3] = {0, 0, 0};
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
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

// ---- few right-hand sides (x = A\b with a vector b): blocked substitution with a GEMV-style update ----
// The recursive solves above would run their off-diagonal updates as MFMA dgemms with a 128-wide tile for
// one or two columns; at n = 16384 that substitution cost ~25 ms for 2 n^2 flop (measured).  Here every
// 128-row diagonal block is one k_trsm_fused launch followed by one k_rhs_update launch that streams the
// block column below (lower) / above (upper) it once: X[rest, :] -= T[rest, blk] * X[blk, :].
static constexpr int RU_MAX_RHS = 8;
static constexpr int RU_ROWS = 64;   // rows per block
static constexpr int RU_KG = 4;      // k groups per row (partial sums meet in LDS)
template <int NRHS>
__global__ void __launch_bounds__(RU_ROWS * RU_KG) k_rhs_update(const double* __restrict__ T, size_t ldt, size_t m, int w,
                                                                const double* __restrict__ Xblk, double* __restrict__ Xrest,
                                                                size_t ldx) {
    __shared__ double xs[NRHS][TRSM_W];
    __shared__ double part[RU_KG][NRHS][RU_ROWS];
    const int tr = threadIdx.x & (RU_ROWS - 1), kg = threadIdx.x / RU_ROWS;
    for (int i = threadIdx.x; i < NRHS * TRSM_W; i += RU_ROWS * RU_KG) {
        const int j = i / TRSM_W, k = i % TRSM_W;
        xs[j][k] = k < w ? Xblk[k + (size_t)j * ldx] : 0.0;
    }
    __syncthreads();
    const size_t r = (size_t)blockIdx.x * RU_ROWS + tr;
    double acc[NRHS];
#pragma unroll
    for (int j = 0; j < NRHS; ++j) acc[j] = 0.0;
    if (r < m) {
        const int kper = (w + RU_KG - 1) / RU_KG;
        const int k0 = kg * kper, k1 = (k0 + kper) < w ? (k0 + kper) : w;
        for (int kb = k0; kb < k1; kb += 8) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = (kb + u < k1) ? T[r + (size_t)(kb + u) * ldt] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kb + u < k1 ? kb + u : k1 - 1;
#pragma unroll
                for (int j = 0; j < NRHS; ++j) acc[j] += a[u] * xs[j][k];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NRHS; ++j) part[kg][j][tr] = acc[j];
    __syncthreads();
    if (kg == 0 && r < m) {
#pragma unroll
        for (int j = 0; j < NRHS; ++j) {
            double sum = part[0][j][tr];
#pragma unroll
            for (int g = 1; g < RU_KG; ++g) sum += part[g][j][tr];
            Xrest[r + (size_t)j * ldx] -= sum;
        }
    }
}

static int launch_rhs_update(Context* c, const double* T, size_t ldt, size_t m, size_t w, const double* Xblk, double* Xrest,
                             size_t ldx, size_t nrhs) {
    if (m == 0 || w == 0) return RMHIP_OK;
    const dim3 grid((unsigned)((m + RU_ROWS - 1) / RU_ROWS)), block(RU_ROWS * RU_KG);
    switch (nrhs) {
        case 1: hipLaunchKernelGGL(k_rhs_update<1>, grid, block, 0, c->stream, T, ldt, m, (int)w, Xblk, Xrest, ldx); break;
        case 2: hipLaunchKernelGGL(k_rhs_update<2>, grid, block, 0, c->stream, T, ldt, m, (int)w, Xblk, Xrest, ldx); break;
        case 3:
        case 4: {
            // a padded right-hand side count would read columns that do not exist: run 4 as 2 + 2, 3 as 2 + 1
            RMHIP_TRY(launch_rhs_update(c, T, ldt, m, w, Xblk, Xrest, ldx, 2));
            return launch_rhs_update(c, T, ldt, m, w, Xblk + 2 * ldx, Xrest + 2 * ldx, ldx, nrhs - 2);
        }
        default: {
            RMHIP_TRY(launch_rhs_update(c, T, ldt, m, w, Xblk, Xrest, ldx, 2));
            return launch_rhs_update(c, T, ldt, m, w, Xblk + 2 * ldx, Xrest + 2 * ldx, ldx, nrhs - 2);
        }
    }
    return launch_check(c);
}


// ---- the whole substitution of a few right-hand sides as ONE launch per direction ------------------------------------------
// 128-row block i is one workgroup.  It streams the tiles T[block i, block j] of the blocks solved before it (their values
// are prefetched into registers while it waits for x_j's flag), subtracts T_ij x_j from its right-hand side, solves its
// diagonal block from LDS (staged at kernel start, long before it is needed) and publishes x_i + a flag.  Workgroups only
// wait for LOWER blockIdx values, and workgroups are dispatched in blockIdx order, so the chain cannot deadlock whatever
// is co-resident; spins are bounded all the same and a timeout raises *err (the caller then repeats the solve with the
// launch-per-block form).  The chain step is flag + last tile + diagonal solve + publish, ~9 us, against ~22-31 us for the
// pair of launches per block (k_trsm_fused stages its triangle after the launch, on the critical path).
// MODE 0: forward, unit lower (blocks ascend); MODE 2: backward, upper with diagonal (logical block = nblk-1-blockIdx.x).
static constexpr int SC_THREADS = 512;
template <int MODE, int NRHS>
__global__ void __launch_bounds__(SC_THREADS) k_subst_chain(const double* __restrict__ LU, size_t lda, size_t n, double* X, size_t ldx,
                                                     unsigned* flags, int* err) {
    extern __shared__ double sc_lds[];
    constexpr int SW = TRSM_W + 1;
    double* Ts = sc_lds;                       // [128][129] triangle of the diagonal block
    double* xs = Ts + TRSM_W * SW;             // [NRHS][128] x_j of the tile in flight, later this block's right-hand side
    double* part = xs + NRHS * TRSM_W;         // [3][NRHS][128] partial sums of the other column quarters
    const int nblk = (int)((n + TRSM_W - 1) / TRSM_W);
    const int ib = MODE == 0 ? (int)blockIdx.x : nblk - 1 - (int)blockIdx.x;
    const size_t r0g = (size_t)ib * TRSM_W;
    const int w = (int)((n - r0g) < (size_t)TRSM_W ? (n - r0g) : (size_t)TRSM_W);
    const int t = threadIdx.x, r = t & (TRSM_W - 1), h = t >> 7;  // row of the block, column quarter of a tile (32 columns)
    const bool row_ok = r < w;
    // stage the needed triangle (eight loads in flight per thread, as k_trsm_fused)
    for (int base = 0; base < TRSM_W * TRSM_W; base += 8 * SC_THREADS) {
        double stage[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * SC_THREADS + t;
            const int rr = idx & (TRSM_W - 1), k = idx >> 7;
            const bool need = k < w && rr < w && (MODE == 0 ? rr > k : rr <= k);
            stage[u] = need ? LU[r0g + rr + (r0g + k) * lda] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * SC_THREADS + t;
            Ts[(idx >> 7) * SW + (idx & (TRSM_W - 1))] = stage[u];
        }
    }
    // ---- off-diagonal tiles: blocks 0..ib-1 (forward) / nblk-1..ib+1 (backward), in the order they finish
    double acc[NRHS];
#pragma unroll
    for (int q = 0; q < NRHS; ++q) acc[q] = 0.0;
    const int ntiles = MODE == 0 ? ib : nblk - 1 - ib;
    double tv[32];  // this thread's 32 values of the tile in flight: row r, columns 32h .. 32h+31
    auto tile_block = [&](int tix) { return MODE == 0 ? tix : nblk - 1 - tix; };
    auto load_tile = [&](int jb) {
        const size_t c0 = (size_t)jb * TRSM_W + 32 * h;
#pragma unroll
        for (int u = 0; u < 32; ++u) tv[u] = (row_ok && c0 + u < n) ? LU[r0g + r + (c0 + u) * lda] : 0.0;
    };
    if (ntiles > 0) load_tile(tile_block(0));
    for (int tix = 0; tix < ntiles; ++tix) {
        const int jb = tile_block(tix);
        // wait for x_jb (one lane polls, the block follows through the barrier)
        if (t == 0) {
            int spins = 0;
            while (__hip_atomic_load(flags + jb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++spins > PK_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // acquire side of the producer's release store (agent scope): pairs the flag with the x values stored before it.  The x
            // reads below are agent-scope atomic loads (served by L2, the coherence point) and would see them anyway; the fence makes
            // the protocol correct by the memory model, not only by the cache hierarchy ("ordering protocols")
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();  // also: every thread is done with the previous tile's xs
        if (t < TRSM_W) {
#pragma unroll
            for (int q = 0; q < NRHS; ++q) {
                const size_t g = (size_t)jb * TRSM_W + t;
                // the flag store follows the x stores in program order behind a fence; read x past the caches
                xs[q * TRSM_W + t] = g < n ? __hip_atomic_load(X + g + (size_t)q * ldx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
        }
        __syncthreads();
        double cur[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) cur[u] = tv[u];
        if (tix + 1 < ntiles) load_tile(tile_block(tix + 1));  // next tile's values travel while this one is applied
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int q = 0; q < NRHS; ++q) acc[q] += cur[u] * xs[q * TRSM_W + 32 * h + u];
        }
    }
    __syncthreads();
    // ---- this block's right-hand side: b - (both column halves), into xs
    if (h > 0) {
#pragma unroll
        for (int q = 0; q < NRHS; ++q) part[((h - 1) * NRHS + q) * TRSM_W + r] = acc[q];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int q = 0; q < NRHS; ++q) {
            const double b = row_ok ? X[r0g + r + (size_t)q * ldx] : 0.0;
            xs[q * TRSM_W + r] = b - (((acc[q] + part[q * TRSM_W + r]) + part[(NRHS + q) * TRSM_W + r]) + part[(2 * NRHS + q) * TRSM_W + r]);
        }
    }
    __syncthreads();
    // ---- diagonal solve: wave q takes right-hand side q; lane i holds rows i and 64 + i (k_trsm_fused's scheme)
    const int lane = t & 63, wv = t >> 6;
    if (wv < NRHS) {
        const int i = lane;
        const bool two = w > 64;
        double x0 = xs[wv * TRSM_W + i], x1 = xs[wv * TRSM_W + 64 + i];
        if (MODE == 0) {
            const int k0end = w < 64 ? w : 64;
#pragma unroll 1
            for (int kb = 0; kb < k0end; kb += 4) {
                double l0[4], l1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u < w ? kb + u : w - 1;
                    l0[u] = Ts[k * SW + i];
                    l1[u] = two ? Ts[k * SW + 64 + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u;
                    if (k < k0end) {
                        const double xk = bcast_lane(x0, k);
                        const double u0 = x0 - l0[u] * xk;
                        x0 = i > k ? u0 : x0;
                        x1 = two ? x1 - l1[u] * xk : x1;
                    }
                }
            }
#pragma unroll 1
            for (int kb = 64; kb < w; kb += 4) {
                double l1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u < w ? kb + u : w - 1;
                    l1[u] = Ts[k * SW + 64 + i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u;
                    if (k < w) {
                        const double xk = bcast_lane(x1, k - 64);
                        const double u1 = x1 - l1[u] * xk;
                        x1 = 64 + i > k ? u1 : x1;
                    }
                }
            }
        } else {
            const double d0 = i < w ? Ts[i * SW + i] : 1.0, d1 = 64 + i < w ? Ts[(64 + i) * SW + 64 + i] : 1.0;
            // x * (1/d) instead of x / d on the dependency chain when every reciprocal is finite and normal (as k_trsm_fused)
            const double rc0 = 1.0 / d0, rc1 = 1.0 / d1;
            const bool recip_ok = !__any(!(fabs(rc0) < 1.0e300 && fabs(rc0) > 1.0e-300 && fabs(rc1) < 1.0e300 && fabs(rc1) > 1.0e-300));
            auto scale = [=](double x, double d, double rc) { return recip_ok ? x * rc : x / d; };
#pragma unroll 1
            for (int kb = w - 1; kb >= 64; kb -= 4) {
                double u0[4], u1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u >= 64 ? kb - u : 64;
                    u0[u] = Ts[k * SW + i];
                    u1[u] = Ts[k * SW + 64 + i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u;
                    if (k >= 64) {
                        const double xk = bcast_lane(scale(x1, d1, rc1), k - 64);
                        const double v1 = x1 - u1[u] * xk;
                        x1 = (64 + i == k) ? xk : (64 + i < k ? v1 : x1);
                        x0 -= u0[u] * xk;
                    }
                }
            }
#pragma unroll 1
            for (int kb = (w < 64 ? w : 64) - 1; kb >= 0; kb -= 4) {
                double u0[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u >= 0 ? kb - u : 0;
                    u0[u] = Ts[k * SW + i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb - u;
                    if (k >= 0) {
                        const double xk = bcast_lane(scale(x0, d0, rc0), k);
                        const double v0 = x0 - u0[u] * xk;
                        x0 = (i == k) ? xk : (i < k ? v0 : x0);
                    }
                }
            }
        }
        if (i < w) __hip_atomic_store(X + r0g + i + (size_t)wv * ldx, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (64 + i < w) __hip_atomic_store(X + r0g + 64 + i + (size_t)wv * ldx, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (t == 0) __hip_atomic_store(flags + ib, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
static int launch_subst_chain(Context* c, const double* LU, size_t lda, size_t n, double* X, size_t ldx, size_t nrhs, unsigned* flags,
                              int* err) {
    const unsigned nblk = (unsigned)((n + TRSM_W - 1) / TRSM_W);
    const size_t lds = ((size_t)TRSM_W * (TRSM_W + 1) + 4 * nrhs * TRSM_W) * sizeof(double);
    switch (nrhs) {
        case 1:
            c->ensure_max_lds((const void*)k_subst_chain<MODE, 1>, 160 * 1024 - 256);
            hipLaunchKernelGGL((k_subst_chain<MODE, 1>), dim3(nblk), dim3(SC_THREADS), lds, c->stream, LU, lda, n, X, ldx, flags, err);
            break;
        case 2:
            c->ensure_max_lds((const void*)k_subst_chain<MODE, 2>, 160 * 1024 - 256);
            hipLaunchKernelGGL((k_subst_chain<MODE, 2>), dim3(nblk), dim3(SC_THREADS), lds, c->stream, LU, lda, n, X, ldx, flags, err);
            break;
        case 3:
            c->ensure_max_lds((const void*)k_subst_chain<MODE, 3>, 160 * 1024 - 256);
            hipLaunchKernelGGL((k_subst_chain<MODE, 3>), dim3(nblk), dim3(SC_THREADS), lds, c->stream, LU, lda, n, X, ldx, flags, err);
            break;
        default:
            c->ensure_max_lds((const void*)k_subst_chain<MODE, 4>, 160 * 1024 - 256);
            hipLaunchKernelGGL((k_subst_chain<MODE, 4>), dim3(nblk), dim3(SC_THREADS), lds, c->stream, LU, lda, n, X, ldx, flags, err);
            break;
    }
    return launch_check(c);
}

static int substitute_few_rhs(Context* c, const double* LU, size_t n, size_t lda, double* X, size_t ldx, size_t nrhs) {
    // One launch per direction (k_subst_chain) when the system is big enough for the launches to matter and the chain's
    // workgroups (one per 128 rows) fit the chip; RMHIP_LU_SUBST=pair keeps the launch-per-block form.
    const char* subst_env = std::getenv("RMHIP_LU_SUBST");  // read per call (tests switch it)
    const int chain = (subst_env && subst_env[0] == 'p') ? 0 : 1;
    const size_t nblk_chain = (n + TRSM_W - 1) / TRSM_W;
    if (chain && nblk_chain >= 4 && nblk_chain <= (size_t)c->num_cus && !c->subst_chain_failed) {
        std::shared_ptr<Allocation> ctl;  // flags of both directions + the error word, pooled
        RMHIP_TRY(c->alloc_device(nblk_chain + 2, &ctl));
        unsigned* flags = (unsigned*)ctl->ptr;
        int* err = (int*)(flags + 2 * nblk_chain);
        RMHIP_HIP_CHECK(hipMemsetAsync(flags, 0, (2 * nblk_chain + 1) * sizeof(unsigned), c->stream));
        for (size_t q0 = 0; q0 < nrhs; q0 += 4) {
            const size_t nq = nrhs - q0 < 4 ? nrhs - q0 : 4;
            if (q0) RMHIP_HIP_CHECK(hipMemsetAsync(flags, 0, 2 * nblk_chain * sizeof(unsigned), c->stream));
            RMHIP_TRY(launch_subst_chain<0>(c, LU, lda, n, X + q0 * ldx, ldx, nq, flags, err));
            RMHIP_TRY(launch_subst_chain<2>(c, LU, lda, n, X + q0 * ldx, ldx, nq, flags + nblk_chain, err));
        }
        int h_err = 0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&h_err, err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (!h_err && std::getenv("RMHIP_LU_TEST_SUBST_RETRY")) h_err = 1;  // test hook for the fallback below
        if (!h_err) return RMHIP_OK;
        c->lu_subst_timeouts++;
        c->subst_chain_failed = true;  // a spin timed out: X is clobbered, the caller gathers the right-hand side again
        return RMHIP_SUBST_RETRY;
    }
    // forward: unit lower
    for (size_t ib = 0; ib < n; ib += TRSM_W) {
        const size_t w = (n - ib) < (size_t)TRSM_W ? (n - ib) : (size_t)TRSM_W;
        RMHIP_TRY(launch_trsm_fused<0>(c, LU + ib + ib * lda, lda, w, X + ib, ldx, nrhs));
        RMHIP_TRY(launch_rhs_update(c, LU + (ib + w) + ib * lda, lda, n - ib - w, w, X + ib, X + ib + w, ldx, nrhs));
    }
    // backward: upper, last block first (blocks aligned to multiples of TRSM_W from the top)
    const size_t nblk = (n + TRSM_W - 1) / TRSM_W;
    for (size_t bi = nblk; bi-- > 0;) {
        const size_t ib = bi * TRSM_W;
        const size_t w = (n - ib) < (size_t)TRSM_W ? (n - ib) : (size_t)TRSM_W;
        RMHIP_TRY(launch_trsm_fused<2>(c, LU + ib + ib * lda, lda, w, X + ib, ldx, nrhs));
        RMHIP_TRY(launch_rhs_update(c, LU + ib * lda, lda, ib, w, X + ib, X, ldx, nrhs));
    }
    return RMHIP_OK;
}

// W (np x np, ld ldw) holds an n x n matrix in its top-left corner: make the rest [0; 0 I] - the padded system [A 0; 0 I] has the
// factors of A in its leading block, the padded rows are never chosen as pivots for A's columns (zeros there) and the padded
// unknowns come out zero.  Used by the solves for n that is not a multiple of 128 (rmhip_ops.cpp: lu_pad_rows).
__global__ void __launch_bounds__(256) k_lu_pad_identity(double* __restrict__ W, size_t ldw, size_t n, size_t np) {
    const size_t pad = np - n;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // row 0 .. np-1
    const size_t j = blockIdx.y;                               // 0 .. np-1: columns < n get their rows >= n, columns >= n whole
    if (i >= np || j >= np) return;
    if (j < n) {
        if (i < pad) W[(n + i) + j * ldw] = 0.0;
    } else {
        W[i + j * ldw] = (i == j) ? 1.0 : 0.0;
    }
}
int lu_pad_identity_device(Context* c, double* W, size_t ldw, size_t n, size_t np) {
    if (np <= n) return RMHIP_OK;
    if (np > 65535) {  // blockIdx.y
        return fail(RMHIP_ERR_UNSUPPORTED, "lu padding: dimension %zu too large", np);
    }
    hipLaunchKernelGGL(k_lu_pad_identity, dim3((unsigned)((np + 255) / 256), (unsigned)np), dim3(256), 0, c->stream, W, ldw, n, np);
    return launch_check(c);
}

// X = U^-1 L^-1 (P B) for square LU (n x n).
int lu_solve_device(Context* c, const double* LU, size_t n, size_t lda, const int* perm_dev, const double* B,
                    size_t nrhs, size_t ldb, double* X, size_t ldx) {
    if (n == 0 || nrhs == 0) return RMHIP_OK;
    if (nrhs > 65535) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: more than 65535 right-hand sides");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256), (unsigned)nrhs), dim3(256), 0, c->stream, B, ldb,
                       perm_dev, n, nrhs, X, ldx);
    RMHIP_TRY(launch_check(c));
    if (nrhs <= (size_t)RU_MAX_RHS && !(lu_skip_mask() & 16)) {
        int rc = substitute_few_rhs(c, LU, n, lda, X, ldx, nrhs);
        if (rc == RMHIP_SUBST_RETRY) {  // the chain kernel timed out (never seen; context flag now selects the pair form)
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256), (unsigned)nrhs), dim3(256), 0, c->stream, B, ldb,
                               perm_dev, n, nrhs, X, ldx);
            RMHIP_TRY(launch_check(c));
            rc = substitute_few_rhs(c, LU, n, lda, X, ldx, nrhs);
        }
        return rc;
    }
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
