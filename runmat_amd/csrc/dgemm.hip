// dgemm.hip -- fp64 GEMM on the CDNA4 matrix cores: C = alpha*A*B + beta*C, column-major.
// Implements `AccelProvider::matmul` (crates/runmat-accelerate-api/src/lib.rs:2375-2381; reference
// semantics crates/runmat-accelerate/src/simple_provider.rs:7698-7741 and the CPU triple loop
// crates/runmat-runtime/src/builtins/common/linalg.rs:6-32) and the trailing update of the blocked
// LU behind `mldivide`.  The reference's own GPU kernel is a 32x32 LDS tile with one thread per C
// element (backend/wgpu/shaders/matmul.rs:1-72); nothing of it is reused.
//
// Design (gfx950, wave64):
//   * v_mfma_f64_16x16x4_f64: one instruction = 16x16x4 = 2048 flop, 64 cycles on a SIMD.  Peak
//     = 256 CU * 4 SIMD * 32 flop/clk = 78.6 TFLOP/s @ 2.4 GHz.  Operands: ONE f64 per lane for A
//     and for B, four f64 accumulators per lane.
//   * Block = 256 threads (4 waves, 2x2), block tile 128x128, K step 16; each wave owns a 64x64
//     sub-tile = 4x4 MFMA tiles = 64 accumulator VGPR pairs... 128 VGPRs.  Per 4-deep k-step a wave
//     issues 8 ds_read_b64 for 16 MFMAs (1024 pipe cycles): LDS bandwidth is irrelevant, the job
//     is to keep the matrix pipe fed, which two resident blocks per CU (2 waves/SIMD) do.
//   * Roles are transposed so that the MFMA "column" index (lane&15, the lane-contiguous one) is
//     the memory-contiguous row index m of the column-major C:  D[r][c] = C[m=c][n=r], hence
//     MFMA-A operand = B^T tile, MFMA-B operand = A^T tile.  Stores are 128-byte segments.
//   * LDS: A tile kept [k][m] (m contiguous, row stride 144 doubles: 144 % 32 == 16 makes the two
//     k-rows a ds_read_b64 half-wave touches land in disjoint bank halves); B tile kept [n][k]
//     (k contiguous, row stride 18 doubles = 2*odd: 16 n-rows x 2 k cover all 64 banks once).
//     Both conflict-free for reads; writes are 16-byte ds_write_b128.
//   * Global -> registers -> LDS double buffering: tile t+1 is fetched (16-byte loads, 1 KiB per
//     wave instruction for A) before the MFMAs of tile t and written to the other LDS buffer after
//     them; one barrier per K tile.
//   * blockIdx -> tile map is XCD-aware: block b runs on XCD b % 8 (observed), so ids are first
//     remapped so each XCD gets a contiguous id range, then laid out in 8-tile-tall groups; the 64
//     blocks co-resident on an XCD then share A rows / B columns through that XCD's 4 MiB L2.
//   * Numerics: each MFMA is a k-ordered fma chain in f64; the CPU reference rounds the product and
//     the sum separately (sum += a*b).  Results agree to ~k*eps*sum|a||b| (tests state the bound).
#include "common.h"

namespace rmhip {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef v2d v2du __attribute__((aligned(8)));  // the same pair on an 8-byte aligned address (guarded kernels)

static constexpr int BM = 128, BN = 128, BK = 16;
static constexpr int SA = BM + 16;  // A tile row stride in doubles ([k][m])
static constexpr int SB = BK + 2;   // B tile row stride in doubles ([n][k])
static constexpr int A_TILE = BK * SA;  // doubles
static constexpr int B_TILE = BN * SB;
// tile rows per group of the XCD-aware order; interleaved A/B at 8192^3 (scripts/dgemm_ab.py): 4 -> 68.7, 8 -> 68.6,
// 16 -> 69.0, 32 -> 69.0 TFLOP/s
#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 16
#endif
static constexpr int GROUP_M = GEMM_GROUP_M;
// after which of the four k-steps of a tile the next tile's registers go to LDS (3 = after the last MFMA;
// measured at 8192^3: 3 -> 68.6, 2 -> 66.4, 1 -> 67.1 TFLOP/s; s_setprio around the MFMA section: no effect)
#ifndef GEMM_PIPE_W8
#define GEMM_PIPE_W8 1
#endif
#ifndef GEMM_PIPE
#define GEMM_PIPE 1
#endif
#ifndef GEMM_STASH_KK
#define GEMM_STASH_KK 3
#endif
static_assert(A_TILE == B_TILE, "the transposed-operand variants swap the two staging patterns between the tiles");

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    unsigned long long lda, ldb, ldc;
    unsigned m, n, k;
    unsigned tiles_m, tiles_n;
    double alpha, beta;
    int rowmap;  // accumulator row formula selector (see store code)
    int vec_a, vec_b;  // EDGE kernel: 16-byte loads are legal for A / B (aligned base, even leading dimension)
    GemmEpilogue ep;   // matmul_epilogue: flags == 0 for plain GEMM
    // split-K (few output tiles, long k): blockIdx.y owns k in [y*k_chunk, min(k, (y+1)*k_chunk)) and writes its
    // partial product (alpha = 1, beta = 0) to C + y*c_split_stride; k_reduce_splits adds them in order.
    unsigned k_chunk;
    unsigned long long c_split_stride;
    // k_dgemm_w8p (persistent form): tile counter (zero before the launch) and the XCD to stay away from (may be null / -1)
    unsigned* tile_counter;
    const int* avoid_xcc;
    // eight-wave tile on the LU's update streams: a device word naming the CU (key: bit 31 | XCC id << 8 | HW_ID cu/sh/se byte) on which
    // k_rp_top is running right now; a block on that CU pauses until the word changes (nullptr: no check)
    const unsigned* yield_word;
    unsigned* announce;  // the same table, for a block of the LU's main stream: it counts itself in on its CU while it runs (CuAnnounce), or nullptr
    int prio;  // nonzero: raise the wave priority (s_setprio 3) - the LU's main-stream updates, which share SIMDs with the update streams' blocks
};

// MatmulEpilogue on one output element, order of crates/runmat-accelerate/src/simple_provider.rs:7800-7836
// The same without the pow step, inlined: what the eight-wave kernel uses when the request has no exponent.  (The out-of-line
// function with pow in it costs the caller 224 bytes of scratch per lane and a large store phase that evicts its neighbours' k loop
// from the instruction cache: 62.7 TFLOP/s at 8192^3 against 70+ for the inlined short form.)
__device__ __forceinline__ double apply_epilogue_nopow(const GemmEpilogue& e, double acc, unsigned mm, unsigned nn) {
    double v = acc * e.alpha + e.beta_add;
    if (e.flags & EP_ROW) v = (e.flags & EP_ROW_DIV) ? v / e.row_scale[mm] : v * e.row_scale[mm];
    if (e.flags & EP_COL) v = (e.flags & EP_COL_DIV) ? v / e.col_scale[nn] : v * e.col_scale[nn];
    if (e.flags & EP_CLAMP_MIN) v = fmax(v, e.clamp_min);
    if (e.flags & EP_CLAMP_MAX) v = fmin(v, e.clamp_max);
    if ((e.flags & EP_DIAG) && mm == nn) e.diag[mm] = v;
    return v;
}
__device__ __noinline__ double apply_epilogue(const GemmEpilogue& e, double acc, unsigned mm, unsigned nn) {
    double v = acc * e.alpha + e.beta_add;
    if (e.flags & EP_ROW) v = (e.flags & EP_ROW_DIV) ? v / e.row_scale[mm] : v * e.row_scale[mm];
    if (e.flags & EP_COL) v = (e.flags & EP_COL_DIV) ? v / e.col_scale[nn] : v * e.col_scale[nn];
    if (e.flags & EP_CLAMP_MIN) v = fmax(v, e.clamp_min);  // Rust f64::max: NaN loses
    if (e.flags & EP_CLAMP_MAX) v = fmin(v, e.clamp_max);
    if (e.flags & EP_POW) v = pow(v, e.pow_exp);
    if ((e.flags & EP_DIAG) && mm == nn) e.diag[mm] = v;
    return v;
}

__device__ __forceinline__ void tile_of_block(const GemmArgs& g, unsigned& tm, unsigned& tn) {
    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned b = blockIdx.x;
    // bijective XCD remap (cdna guide 5.x): blocks of one XCD get consecutive ids
    const unsigned xcd = b & 7u, idx = b >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // grouped ordering: GROUP_M tile rows at a time, column by column
    const unsigned per_group = GROUP_M * g.tiles_n;
    const unsigned group = wg / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = (g.tiles_m - first_m) < GROUP_M ? (g.tiles_m - first_m) : GROUP_M;
    const unsigned in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
}

// EDGE = false: m % 128 == 0, n % 128 == 0, k % 16 == 0, even leading dimensions, 16-byte aligned bases.
// EPI = true: the MatmulEpilogue variant.  It is a SEPARATE instantiation on purpose: inlining the
// epilogue (pow!) into the plain kernel cost 2.7x (66.8 -> 24.8 TFLOP/s) -- waves in the bloated
// store phase evicted the main loop of their CU pair's instruction cache.
// TA / TB: the operand is given TRANSPOSED in memory (op(A) = At', At is k x m with k contiguous; op(B) =
// Bt', Bt is n x k with n contiguous) -- RunMat's transpose views (`handle_transpose_info`, lib.rs:218-245;
// `A'*B`, `syrk`).  A transposed A has exactly the memory pattern of a plain B (k contiguous per tile row)
// and vice versa, so the two staging patterns below simply swap roles and the LDS tiles keep their sizes:
//   pattern M: 128 tile rows contiguous in memory, LDS [k][x] with row stride SA   (A plain, B transposed)
//   pattern K: k contiguous in memory,            LDS [y][k] with row stride SB   (B plain, A transposed)
// PRE (only with EDGE = EPI = TA = TB = false): C <- C - A*B, the rank-k update of the blocked LU.  The C tile is loaded into
// the accumulators (negated) BEFORE the k loop, where its latency hides behind the first operand tiles, and the epilogue
// is stores only (-acc).  The plain beta path reads C at the end: four dependent load -> store round trips with the
// matrix pipe idle - 10-25 % of a k = 256..512 tile when one block per CU runs (the look-ahead's update stream).
template <bool EDGE, bool EPI, bool TA, bool TB, bool PRE = false>
__global__ void __launch_bounds__(256, 2) k_dgemm(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (g.prio) __builtin_amdgcn_s_setprio(3);
    const CuAnnounce on_cu(g.announce);
    double* As = lds;                    // [2][A_TILE]
    double* Bs = lds + 2 * A_TILE;       // [2][B_TILE]   (A_TILE == B_TILE)

    unsigned tm, tn;
    tile_of_block(g, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    // this block's k range (the whole k unless the launch is split)
    const unsigned kbeg = blockIdx.y * g.k_chunk;
    const unsigned klen = (g.k - kbeg) < g.k_chunk ? (g.k - kbeg) : g.k_chunk;
    const double* const Ab = TA ? g.A + kbeg : g.A + (size_t)kbeg * g.lda;
    const double* const Bb = TB ? g.B + (size_t)kbeg * g.ldb : g.B + kbeg;
    double* const Cb = g.C + (size_t)blockIdx.y * g.c_split_stride;

    // staging assignment
    const int p_xp = t & 63;   // pattern M: pair index along the contiguous tile dimension (x = 2*p_xp)
    const int p_kc = t >> 6;   //            0..3, k = p_kc + 4*p
    const int q_kp = t & 7;    // pattern K: k pair index (k = 2*q_kp)
    const int q_y = t >> 3;    //            0..31, y = q_y + 32*p

    v2d ra[4], rb[4];

    // pattern M fetch: element (x, kk) = ptr[x + kk * ld]
    auto fetchM = [&](const double* ptr, unsigned long long ld, unsigned x0, unsigned xlim, unsigned k0, int vec, v2d* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned kk = k0 + p_kc + 4 * p;
            const unsigned xx = x0 + 2 * p_xp;
            if (!EDGE) {
                r[p] = *(const v2d*)(ptr + (size_t)kk * ld + xx);
            } else {
                v2d v = {0.0, 0.0};
                if (kk < klen) {
                    const double* src = ptr + (size_t)kk * ld + xx;
                    if (vec && xx + 1 < xlim) {
                        v = *(const v2d*)src;
                    } else {
                        if (xx < xlim) v.x = src[0];
                        if (xx + 1 < xlim) v.y = src[1];
                    }
                }
                r[p] = v;
            }
        }
    };
    // pattern K fetch: element (y, kk) = ptr[y * ld + kk]
    auto fetchK = [&](const double* ptr, unsigned long long ld, unsigned y0, unsigned ylim, unsigned k0, int vec, v2d* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned yy = y0 + q_y + 32 * p;
            const unsigned kk = k0 + 2 * q_kp;
            if (!EDGE) {
                r[p] = *(const v2d*)(ptr + (size_t)yy * ld + kk);
            } else {
                v2d v = {0.0, 0.0};
                if (yy < ylim) {
                    const double* src = ptr + (size_t)yy * ld + kk;
                    if (vec && kk + 1 < klen) {
                        v = *(const v2d*)src;
                    } else {
                        if (kk < klen) v.x = src[0];
                        if (kk + 1 < klen) v.y = src[1];
                    }
                }
                r[p] = v;
            }
        }
    };
    auto stashM = [&](double* tile, const v2d* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2d*)(tile + (p_kc + 4 * p) * SA + 2 * p_xp) = r[p];
    };
    auto stashK = [&](double* tile, const v2d* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2d*)(tile + (q_y + 32 * p) * SB + 2 * q_kp) = r[p];
    };
    auto fetch = [&](unsigned k0) {
        if (TA) fetchK(Ab, g.lda, m0, g.m, k0, g.vec_a, ra);
        else fetchM(Ab, g.lda, m0, g.m, k0, g.vec_a, ra);
        if (TB) fetchM(Bb, g.ldb, n0, g.n, k0, g.vec_b, rb);
        else fetchK(Bb, g.ldb, n0, g.n, k0, g.vec_b, rb);
    };
    auto stash = [&](int buf) {
        if (TA) stashK(As + buf * A_TILE, ra);
        else stashM(As + buf * A_TILE, ra);
        if (TB) stashM(Bs + buf * B_TILE, rb);
        else stashK(Bs + buf * B_TILE, rb);
    };

    v4d acc[4][4];  // [tj (n)][ti (m)]
    if (PRE) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned mm = m0 + wm * 64 + i * 16 + l15;
                    const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                    const unsigned nn = n0 + wn * 64 + j * 16 + row;
                    acc[j][i][r] = -Cb[(size_t)nn * g.ldc + mm];
                }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = v4d{0.0, 0.0, 0.0, 0.0};
    }

    const unsigned ktiles = (klen + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();

    // fragment addressing: pattern M tiles are read [k][x], pattern K tiles [y][k]
    const int a_off = TA ? (wm * 64 + l15) * SB + lq : lq * SA + wm * 64 + l15;
    const int b_off = TB ? lq * SA + wn * 64 + l15 : (wn * 64 + l15) * SB + lq;
    constexpr int A_KSTEP = TA ? 4 : 4 * SA, A_ISTEP = TA ? 16 * SB : 16;
    constexpr int B_KSTEP = TB ? 4 * SA : 4, B_JSTEP = TB ? 16 : 16 * SB;

#if GEMM_PIPE
    if (!EDGE && !PRE) {  // (the C-preloading variant has no registers to spare for the second prefetch set: 132 bytes of scratch)
        // Software-pipelined k loop (one barrier per tile, nothing exposed around it): the fragments of step kk+1 are read while
        // step kk multiplies; the next tile is stashed during step 2 from registers that were loaded a whole tile earlier (and
        // refilled at once with the tile after it); the barrier sits before step 3 - every wave has issued all its reads of the
        // current buffer by then, so the buffer may be overwritten any time after it - and step 3 already reads the next
        // tile's first fragments.  The barrier is raw (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads would also wait for
        // the global loads just issued.
        auto frags = [&](const double* a, const double* b, int kk, double (&af)[4], double (&bf)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[j * B_JSTEP + kk * B_KSTEP];
        };
        auto mma = [&](const double (&af)[4], const double (&bf)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
        };
        // global prefetch distance: TWO tiles (register sets A = ra/rb and B = ra2/rb2 alternate; the loop is unrolled by two)
        v2d ra2[4], rb2[4];
        auto fetch_into = [&](unsigned k0, v2d* pa, v2d* pb) {
            if (TA) fetchK(Ab, g.lda, m0, g.m, k0, g.vec_a, pa);
            else fetchM(Ab, g.lda, m0, g.m, k0, g.vec_a, pa);
            if (TB) fetchM(Bb, g.ldb, n0, g.n, k0, g.vec_b, pb);
            else fetchK(Bb, g.ldb, n0, g.n, k0, g.vec_b, pb);
        };
        auto stash_from = [&](int buf, const v2d* pa, const v2d* pb) {
            if (TA) stashK(As + buf * A_TILE, pa);
            else stashM(As + buf * A_TILE, pa);
            if (TB) stashM(Bs + buf * B_TILE, pb);
            else stashK(Bs + buf * B_TILE, pb);
        };
        auto clampt = [&](unsigned t) { return (t < ktiles ? t : ktiles - 1) * BK; };
        fetch_into(clampt(1), ra, rb);
        fetch_into(clampt(2), ra2, rb2);
        double af0[4], bf0[4], af1[4], bf1[4];
        frags(As + a_off, Bs + b_off, 0, af0, bf0);
        auto step = [&](unsigned kt, v2d* pa, v2d* pb) {
            const int cur = kt & 1;
            const double* a = As + cur * A_TILE + a_off;
            const double* b = Bs + cur * B_TILE + b_off;
            // (sched_group_barrier: mask 0x100 = LDS reads, 0x200 = LDS writes, 0x020 = global loads, 0x008 = MFMA.  Left alone the
            // scheduler issues each step's reads BEHIND the previous step's MFMAs and then waits for them.)
            frags(a, b, 1, af1, bf1);
            mma(af0, bf0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
            frags(a, b, 2, af0, bf0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
            // (no branches here: the last tiles stash / fetch a tile nobody reads, so that the writes and loads can sit between MFMAs)
            stash_from(cur ^ 1, pa, pb);     // tile kt + 1, loaded two tiles ago
            fetch_into(clampt(kt + 3), pa, pb);
            frags(a, b, 3, af1, bf1);
            mma(af0, bf0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            frags(As + (cur ^ 1) * A_TILE + a_off, Bs + (cur ^ 1) * B_TILE + b_off, 0, af0, bf0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the prefetched fragments are "used" here, behind 16 MFMAs: the wait the compiler owes them lands where it is free
            // instead of in front of the next step's first MFMA, where it would also cover that step's own reads
            asm volatile("" : "+v"(af0[0]), "+v"(af0[1]), "+v"(af0[2]), "+v"(af0[3]), "+v"(bf0[0]), "+v"(bf0[1]), "+v"(bf0[2]), "+v"(bf0[3]));
        };
        for (unsigned kt = 0; kt < ktiles; kt += 2) {
            step(kt, ra, rb);
            if (kt + 1 < ktiles) step(kt + 1, ra2, rb2);
        }
    } else
#endif
    for (unsigned kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) fetch((kt + 1) * BK);
        const double* a = As + cur * A_TILE + a_off;
        const double* b = Bs + cur * B_TILE + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[j * B_JSTEP + kk * B_KSTEP];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
            if (kk == GEMM_STASH_KK && kt + 1 < ktiles) stash(cur ^ 1);
        }
        __syncthreads();
    }

    // epilogue: D[r][c] -> C[m = c][n = r];  c = lane & 15.
    // rowmap 0: r = 4*reg + (lane >> 4)   (f64 16x16x4 map per the CDNA4 guide)
    // rowmap 1: r = 4*(lane >> 4) + reg   (the f32-family map; kept selectable for bring-up)
    // beta != 0 (LU trailing update): the 16 C values of one n-tile row are loaded together before
    // any store, so the read-modify-write costs one memory round trip per group instead of 64
    // serialized ones (a load cannot be hoisted above an earlier store to the same array).
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double* dst[16];
        bool ok[16];
        double prev[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mm = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                const unsigned nn = n0 + wn * 64 + j * 16 + row;
                ok[i * 4 + r] = !EDGE || (mm < g.m && nn < g.n);
                dst[i * 4 + r] = Cb + (size_t)nn * g.ldc + mm;
            }
        }
        if (!EPI && !PRE && g.beta != 0.0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) prev[e] = ok[e] ? *dst[e] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = i * 4 + r;
                double v;
                if (EPI) {
                    const unsigned mm = m0 + wm * 64 + i * 16 + l15;
                    const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                    const unsigned nn = n0 + wn * 64 + j * 16 + row;
                    v = ok[e] ? apply_epilogue(g.ep, acc[j][i][r], mm, nn) : 0.0;
                } else if (PRE) {
                    v = -acc[j][i][r];
                } else {
                    v = g.alpha * acc[j][i][r];
                    if (g.beta != 0.0) v = g.beta * prev[e] + v;
                }
                if (ok[e]) *dst[e] = v;
            }
        }
    }
    on_cu.done();
}


// ---- small-tile variant: 64 x 64 block tile, the same four waves each owning 32 x 32 (2 x 2 MFMA tiles) -------------------
// For products with few 128 x 128 tiles and a short k - the in-panel and look-ahead updates of the blocked LU (rows x 64..256
// x 64..256, the main stream's critical path under look-ahead), small `matmul`s.  There the big kernel is latency bound:
// a 128 x 128 x 128 block is eight dependent k tiles of 16 MFMAs per wave each (1.7 us per tile) on 1/4 of the CUs; four
// times the blocks with a quarter of the MFMAs per tile finish the same product in about a third of the time.
// Plain operands only (no transposed views, no epilogue), m % 64 == n % 64 == 0, k % 16 == 0, 16-byte aligned bases
// and even leading dimensions; everything else stays with k_dgemm.  Same LDS layouts and operand roles as k_dgemm.
// (Tried: for k <= 128 request every operand tile before the first MFMA - one memory latency instead of one per tile.  8192 x 64 x 64
// alone 7.2 -> 6.9 us, launch overhead dominates; inside the LU the 188 VGPRs it needs cost more than that: n = 8192 solve 34.6 -> 35.7 ms.)
static constexpr int SM = 64, SN = 64;
static constexpr int SSA = SM + 16;          // A tile [k][m] row stride (80 % 32 == 16: the bank-half trick of k_dgemm)
static constexpr int S_A_TILE = BK * SSA;    // 1280 doubles
static constexpr int S_B_TILE = SN * SB;     // 1152 doubles
// GUARD: m, n need not be multiples of 64 nor k of 16 (even m, k and leading dimensions, aligned bases): tile rows / columns beyond
// the matrix re-read its last ones, the k tail of the last tile is zeroed, stores are checked - as in the eight-wave tile below.
// 1000^3 on the guarded 128 x 128 kernel: 173 us (64 blocks, every element checked); 1024^3 here: 65 us.
template <bool PRE, bool GUARD = false>  // PRE: C <- C - A*B with the C tile preloaded into the accumulators (see k_dgemm)
__global__ void __launch_bounds__(256) k_dgemm_small(const GemmArgs g) {
    if (g.prio) __builtin_amdgcn_s_setprio(3);
    const CuAnnounce on_cu(g.announce);
    __shared__ __attribute__((aligned(16))) double As[2][S_A_TILE];
    __shared__ __attribute__((aligned(16))) double Bs[2][S_B_TILE];
    const unsigned tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;  // column-major tile order: neighbours share B
    const unsigned m0 = tm * SM, n0 = tn * SN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int p_xp = t & 31, p_kc = t >> 5;  // A: pair along m (x = 2*p_xp), k = p_kc + 8*p
    const int q_kp = t & 7, q_y = t >> 3;    // B: pair along k (k = 2*q_kp), y = q_y + 32*p
    const unsigned am = m0 + 2 * p_xp;
    const unsigned a_r0 = (GUARD && am >= g.m) ? g.m - 1 : am;
    const bool a_single = GUARD && a_r0 + 1 >= g.m;                 // the matrix's last row alone (odd m) or a clamped pair
    const bool a_edge = GUARD && m0 + SM > g.m;  // uniform: such threads exist in this block (an even m too: its clamped threads sit on the last row and must not read a pair)
    const double* const Ap = g.A + a_r0;
    const double* const Bp = g.B + 2 * q_kp;
    size_t b_row[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const unsigned bn = n0 + q_y + 32 * p;
        b_row[p] = (size_t)((GUARD && bn >= g.n) ? g.n - 1 : bn) * g.ldb;
    }
    auto ld2 = [&](const double* q) -> v2d { return GUARD ? (v2d)(*(const v2du*)q) : *(const v2d*)q; };
    auto ldM = [&](const double* q) -> v2d {
        if (a_edge && a_single) return v2d{*q, 0.0};
        return ld2(q);
    };
    v2d ra[2], rb[2];
    auto fetch = [&](unsigned k0) {
        if (GUARD && k0 + BK > g.k) {  // last, partial k tile (uniform)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned km = k0 + p_kc + 8 * p, kp = k0 + 2 * q_kp;
                ra[p] = km < g.k ? ldM(Ap + (size_t)km * g.lda) : v2d{0.0, 0.0};
                const double* pair = Bp + b_row[p] + k0;
                rb[p] = kp + 1 < g.k ? ld2(pair) : (kp < g.k ? v2d{*pair, 0.0} : v2d{0.0, 0.0});
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ra[p] = ldM(Ap + (size_t)(k0 + p_kc + 8 * p) * g.lda);
            rb[p] = ld2(Bp + b_row[p] + k0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *(v2d*)(&As[buf][(p_kc + 8 * p) * SSA + 2 * p_xp]) = ra[p];
            *(v2d*)(&Bs[buf][(q_y + 32 * p) * SB + 2 * q_kp]) = rb[p];
        }
    };
    v4d acc[2][2];  // [tj (n)][ti (m)]
    double* dst[16];
    bool ok[16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned mm = m0 + wm * 32 + i * 16 + l15;
                const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                const unsigned nn = n0 + wn * 32 + j * 16 + row;
                dst[(j * 2 + i) * 4 + r] = g.C + (size_t)nn * g.ldc + mm;
                ok[(j * 2 + i) * 4 + r] = !GUARD || (mm < g.m && nn < g.n);
            }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][i][r] = (PRE && ok[(j * 2 + i) * 4 + r]) ? -*dst[(j * 2 + i) * 4 + r] : 0.0;
    const unsigned ktiles = GUARD ? (g.k + BK - 1) / BK : g.k / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int a_off = lq * SSA + wm * 32 + l15;
    const int b_off = (wn * 32 + l15) * SB + lq;
    for (unsigned kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) fetch((kt + 1) * BK);
        const double* a = &As[cur][a_off];
        const double* b = &Bs[cur][b_off];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = a[kk * 4 * SSA + i * 16];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = b[j * 16 * SB + kk * 4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < ktiles) stash(cur ^ 1);
        __syncthreads();
    }
    // epilogue as in k_dgemm: all loads of the read-modify-write first, then the stores
    double prev[16];
    if (!PRE && g.beta != 0.0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) prev[e] = ok[e] ? *dst[e] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (j * 2 + i) * 4 + r;
                double v;
                if (PRE) {
                    v = -acc[j][i][r];
                } else {
                    v = g.alpha * acc[j][i][r];
                    if (g.beta != 0.0) v = g.beta * prev[e] + v;
                }
                if (ok[e]) *dst[e] = v;
            }
    on_cu.done();
}


// ---- eight-wave variant of the 128 x 128 tile: two waves per SIMD inside ONE block ------------------------------------------
// Under the LU's look-ahead the update stream may keep only one dgemm block per CU (its LDS pad leaves the rest of the CU
// to the main stream).  With k_dgemm that is one wave per SIMD, and every barrier, LDS refill and late global load of
// the k loop is a bubble in the matrix pipe: 52-56 TFLOP/s alone against 62-65 with two blocks per CU.  Here the same
// tile is computed by 2 (m) x 4 (n) waves of 64 x 32 each (64 accumulator registers instead of 128), so a second wave
// is there to issue MFMAs while the first waits - at the same 73.7 KiB of LDS and 125-147 VGPRs (two blocks of the plain
// variant per CU).  Plain operands, unguarded shapes only (m, n % 128, k % 16, even
// leading dimensions, aligned bases); PRE as in k_dgemm.  Same k-ordered MFMA chain per element: bit-identical results.
// TA / TB: the operand is stored transposed, as in k_dgemm (a transposed A is staged with B's pattern and vice versa).
// EPI: the MatmulEpilogue on every output element (separate instantiations, as for k_dgemm): 1 = without a pow step (inlined),
// 2 = any request (out of line).
// GUARD: m, n need not be multiples of 128 nor k of 16 (still: even leading dimensions and aligned bases, an even k, an even m for
// a plain A, no transposed B).  Tile rows / columns beyond the matrix re-read its last ones - their accumulators are never stored -
// and the k tail of the LAST tile is zeroed in both operands; every other iteration runs the unguarded loads.  The guarded k_dgemm
// checks every element of every tile: 8200^3 58.8 TFLOP/s against 72.5 at 8192^3.
template <bool PRE, bool TA = false, bool TB = false, int EPI = 0, bool GUARD = false, bool YIELD = false>
__device__ __forceinline__ void w8_tile(const GemmArgs& g, const unsigned tm, const unsigned tn, double* As, double* Bs,
                                        const unsigned kbeg = 0, const unsigned klen_or_0 = 0, const size_t c_off = 0) {
    // (kbeg, klen, c_off): the k slice and the partial-product offset of a split-K block; defaults = the whole product
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const double* const gA = g.A + (TA ? (size_t)kbeg : (size_t)kbeg * g.lda);
    const double* const gB = g.B + (TB ? (size_t)kbeg * g.ldb : (size_t)kbeg);
    double* const gC = g.C + c_off;
    const unsigned gk = klen_or_0 ? klen_or_0 : g.k;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;  // wn 0..3: 32 columns each
    const int l15 = lane & 15, lq = lane >> 4;
    const int p_xp = t & 63, p_kc = t >> 6;  // pattern M (128 contiguous x 16 k): pair along the tile dimension, k = p_kc + 8*p
    const int q_kp = t & 7, q_y = t >> 3;    // pattern K (k contiguous): pair along k, y = q_y + 64*p
    static_assert(!(GUARD && TB), "the guarded tile is not instantiated for a transposed B");
    // A: plain = pattern M on (m, k) with ld = lda; transposed (stored k x m... as At[k + m*lda]) = pattern K over rows m
    // B: plain = pattern K over rows n (B[k + n*ldb]); transposed (Bt[n + k*ldb]) = pattern M on (n, k)
    // GUARD: any m, n, k, leading dimensions and 8-byte aligned bases.  The row (pattern K) or first row of the pair (pattern M) this
    // thread stages is clamped into the matrix; 16-byte loads are issued on 8-byte aligned addresses (v2du: the hardware splits the
    // ones that straddle); a pair whose second element lies outside the matrix - the last row of an odd m, the last k of an odd
    // k - is loaded as a scalar, in the hot loop only by the blocks on the matrix's lower edge (uniform flag).
    auto rowY = [&](unsigned r, unsigned lim) { return (GUARD && r >= lim) ? lim - 1 : r; };
    const unsigned a_r0 = rowY(m0 + 2 * p_xp, g.m);
    const bool a_single = GUARD && !TA && a_r0 + 1 >= g.m;                 // this thread's pair is the matrix's last row alone (or clamped)
    const bool a_edge = GUARD && !TA && m0 + BM > g.m;  // uniform: such threads exist in this block (an even m too: its clamped threads sit on the last row and must not read a pair)
    const double* const Ap = TA ? gA + 2 * q_kp : gA + a_r0;
    const double* const Bp = TB ? gB + n0 + 2 * p_xp : gB + 2 * q_kp;
    size_t a_row[2], b_row[2];  // pattern K: element offset of this thread's two rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a_row[p] = (size_t)rowY(m0 + q_y + 64 * p, g.m) * g.lda;
        b_row[p] = (size_t)rowY(n0 + q_y + 64 * p, g.n) * g.ldb;
    }
    auto ld2 = [&](const double* q) -> v2d { return GUARD ? (v2d)(*(const v2du*)q) : *(const v2d*)q; };
    auto ldM = [&](const double* q) -> v2d {  // pattern M pair of A
        if (a_edge && a_single) return v2d{*q, 0.0};
        return ld2(q);
    };
    v2d ra[2], rb[2];
    auto fetch = [&](unsigned k0) {
        if (GUARD && k0 + BK > gk) {  // the last, partial k tile (uniform): nothing beyond k is loaded, the tail is zero in both operands
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned km = k0 + p_kc + 8 * p, kp = k0 + 2 * q_kp;  // pattern M: one k per load; pattern K: a pair along k
                auto ldK = [&](const double* pair) -> v2d {  // pair = address of element kp of the row (Ap / Bp carry the + 2 q_kp)
                    if (kp + 1 < gk) return ld2(pair);
                    if (kp < gk) return v2d{*pair, 0.0};
                    return v2d{0.0, 0.0};
                };
                ra[p] = TA ? ldK(Ap + a_row[p] + k0) : (km < gk ? ldM(Ap + (size_t)km * g.lda) : v2d{0.0, 0.0});
                rb[p] = TB ? (km < gk ? *(const v2d*)(Bp + (size_t)km * g.ldb) : v2d{0.0, 0.0}) : ldK(Bp + b_row[p] + k0);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ra[p] = TA ? ld2(Ap + a_row[p] + k0) : ldM(Ap + (size_t)(k0 + p_kc + 8 * p) * g.lda);
            rb[p] = TB ? *(const v2d*)(Bp + (size_t)(k0 + p_kc + 8 * p) * g.ldb) : ld2(Bp + b_row[p] + k0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (TA) *(v2d*)(As + buf * A_TILE + (q_y + 64 * p) * SB + 2 * q_kp) = ra[p];
            else *(v2d*)(As + buf * A_TILE + (p_kc + 8 * p) * SA + 2 * p_xp) = ra[p];
            if (TB) *(v2d*)(Bs + buf * B_TILE + (p_kc + 8 * p) * SA + 2 * p_xp) = rb[p];
            else *(v2d*)(Bs + buf * B_TILE + (q_y + 64 * p) * SB + 2 * q_kp) = rb[p];
        }
    };
    v4d acc[2][4];  // [tj (n)][ti (m)]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (PRE) {
                    const unsigned mm = m0 + wm * 64 + i * 16 + l15;
                    const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                    const unsigned nn = n0 + wn * 32 + j * 16 + row;
                    acc[j][i][r] = (!GUARD || (mm < g.m && nn < g.n)) ? -gC[(size_t)nn * g.ldc + mm] : 0.0;
                } else {
                    acc[j][i][r] = 0.0;
                }
            }
    const unsigned ktiles = GUARD ? (gk + BK - 1) / BK : gk / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int a_off = TA ? (wm * 64 + l15) * SB + lq : lq * SA + wm * 64 + l15;
    const int b_off = TB ? lq * SA + wn * 32 + l15 : (wn * 32 + l15) * SB + lq;
    constexpr int A_KSTEP = TA ? 4 : 4 * SA, A_ISTEP = TA ? 16 * SB : 16;
    constexpr int B_KSTEP = TB ? 4 * SA : 4, B_JSTEP = TB ? 16 : 16 * SB;
#if GEMM_PIPE_W8
    {
        // the software-pipelined k loop of k_dgemm (there: two blocks per CU fill each other's bubbles and it is worth 1 %; here all
        // eight waves of the CU share ONE barrier, and what surrounds it - LDS write completion, the first reads of the next tile -
        // is matrix-pipe idle time).  One tile of global prefetch only: with two (140-162 VGPRs) one padded block per CU gains
        // another 2.8 % at 8192^3 (66.9 TFLOP/s) but nothing at the LU's k = 128..512 (solve 100.4-100.9 vs 100.4-100.6 ms), and
        // the plain variant would no longer fit two blocks per CU.  (Capped at 128 VGPRs - so that a main-stream dgemm wave still
        // shares the SIMD - it spills and the solve takes 101.6 ms.)
        auto frags = [&](const double* a, const double* b, int kk, double (&af)[4], double (&bf)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = b[j * B_JSTEP + kk * B_KSTEP];
        };
        auto mma = [&](const double (&af)[4], const double (&bf)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
        };
        auto clampt = [&](unsigned tt) { return (tt < ktiles ? tt : ktiles - 1) * BK; };
        fetch(clampt(1));
        double af0[4], bf0[2], af1[4], bf1[2];
        frags(As + a_off, Bs + b_off, 0, af0, bf0);
        // cooperative yield (YIELD instantiation, GemmArgs::yield_word): the word read during the previous k tile names the CU on which the
        // LU's k_rp_top is running; if that is this CU, sleep until it changes (bounded: ~2 ms).  A separate instantiation: the check in the
        // plain kernel's pipelined loop cost the 8192^3 product 5 % (15.2 -> 16.0 ms)
        const unsigned* my_slot = nullptr;
        unsigned yv = 0;
        if (YIELD && g.yield_word) my_slot = g.yield_word + cu_slot();  // this CU's counter of the table (common.h: CuAnnounce)
        for (unsigned kt = 0; kt < ktiles; ++kt) {
            if (YIELD && g.yield_word) {
                if (__builtin_amdgcn_readfirstlane(yv) != 0) {
                    for (int spin = 0; spin < 4096 && __hip_atomic_load(my_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; ++spin)
                        __builtin_amdgcn_s_sleep(16);
                }
                yv = __hip_atomic_load(my_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int cur = kt & 1;
            const double* a = As + cur * A_TILE + a_off;
            const double* b = Bs + cur * B_TILE + b_off;
            frags(a, b, 1, af1, bf1);
            mma(af0, bf0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            frags(a, b, 2, af0, bf0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            stash(cur ^ 1);            // tile kt + 1 (the last tile stashes a copy nobody reads: no branch between the MFMAs)
            fetch(clampt(kt + 2));
            frags(a, b, 3, af1, bf1);
            mma(af0, bf0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            frags(As + (cur ^ 1) * A_TILE + a_off, Bs + (cur ^ 1) * B_TILE + b_off, 0, af0, bf0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(af0[0]), "+v"(af0[1]), "+v"(af0[2]), "+v"(af0[3]), "+v"(bf0[0]), "+v"(bf0[1]));
        }
        __syncthreads();  // the caller may reuse the LDS tiles (persistent form)
    }
#else
    for (unsigned kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) fetch((kt + 1) * BK);
        const double* a = As + cur * A_TILE + a_off;
        const double* b = Bs + cur * B_TILE + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = b[j * B_JSTEP + kk * B_KSTEP];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < ktiles) stash(cur ^ 1);
        __syncthreads();
    }
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double* dst[16];
        double prev[16];
        bool ok[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned mm = m0 + wm * 64 + i * 16 + l15;
                const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                const unsigned nn = n0 + wn * 32 + j * 16 + row;
                dst[i * 4 + r] = gC + (size_t)nn * g.ldc + mm;
                ok[i * 4 + r] = !GUARD || (mm < g.m && nn < g.n);
            }
        if (!PRE && !EPI && g.beta != 0.0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) prev[e] = ok[e] ? *dst[e] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = i * 4 + r;
                double v;
                if (PRE) {
                    v = -acc[j][i][r];
                } else if (EPI) {
                    const unsigned mm = m0 + wm * 64 + i * 16 + l15;
                    const unsigned row = g.rowmap ? (4 * lq + r) : (4 * r + lq);
                    if (GUARD && !ok[e]) continue;
                    v = EPI == 1 ? apply_epilogue_nopow(g.ep, acc[j][i][r], mm, n0 + wn * 32 + j * 16 + row)
                                 : apply_epilogue(g.ep, acc[j][i][r], mm, n0 + wn * 32 + j * 16 + row);
                } else {
                    v = g.alpha * acc[j][i][r];
                    if (g.beta != 0.0) v = g.beta * prev[e] + v;
                }
                if (ok[e]) *dst[e] = v;
            }
    }
}

// gridDim.y > 1: split-K (GemmArgs::k_chunk) - blockIdx.y owns a slice of k and writes its partial product
template <bool PRE, bool TA = false, bool TB = false, int EPI = 0, bool YIELD = false>
__global__ void __launch_bounds__(512) k_dgemm_w8(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    unsigned tm, tn;
    tile_of_block(g, tm, tn);
    const unsigned kbeg = blockIdx.y * g.k_chunk;  // one slice (gridDim.y == 1): k_chunk == k
    const unsigned klen = (g.k - kbeg) < g.k_chunk ? (g.k - kbeg) : g.k_chunk;
    w8_tile<PRE, TA, TB, EPI, false, YIELD>(g, tm, tn, lds, lds + 2 * A_TILE, kbeg, klen, (size_t)blockIdx.y * g.c_split_stride);
}
// the guarded tile (any shape).  Four waves per SIMD (two blocks per CU) are asked for explicitly: the edge bookkeeping would otherwise push the kernel a
// couple of registers over the 128 that four waves per SIMD allow (one block per CU: 59.5 instead of 63.8 TFLOP/s at 8200^3).
template <bool PRE, bool TA = false, int EPI = 0>
__global__ void __launch_bounds__(512, 4) k_dgemm_w8g(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    unsigned tm, tn;
    tile_of_block(g, tm, tn);
    const unsigned kbeg = blockIdx.y * g.k_chunk;
    const unsigned klen = (g.k - kbeg) < g.k_chunk ? (g.k - kbeg) : g.k_chunk;
    w8_tile<PRE, TA, false, EPI, true>(g, tm, tn, lds, lds + 2 * A_TILE, kbeg, klen, (size_t)blockIdx.y * g.c_split_stride);
}

// Persistent form for the look-ahead LU's late phase: one workgroup per CU, tiles handed out by a counter, and workgroups
// that find themselves on the XCD the panel kernel occupies (g.avoid_xcc, written by k_lu_panel2) leave at once - the
// update then runs on the other seven XCDs and the panel's XCD keeps free CUs and a quiet L2 (lu.hip, getrf_blocked).
// (Before the pipelined k loop this kernel had 118 VGPRs and a placeholder workgroup fitted beside a 270-register panel wave; at
// 147 the ones dispatched to the panel's CUs start - and leave - when the panel block retires.  Measured: no difference.)
__global__ void __launch_bounds__(512) k_dgemm_w8p(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned s_tile;
    __shared__ int s_avoid;
    if (g.avoid_xcc && gridDim.x >= 16) {  // (a grid that small may sit on the avoided XCD entirely: somebody has to do the work)
        // ONE read per workgroup: the panel kernel may write *avoid_xcc while this workgroup starts, and waves that saw different
        // values would split it - half a workgroup computing half of every tile it draws
        if (threadIdx.x == 0) s_avoid = __hip_atomic_load(g.avoid_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((int)(xcc & 0xf) == s_avoid) return;
    }
    const unsigned nwg = g.tiles_m * g.tiles_n;
    for (;;) {
        if (threadIdx.x == 0) s_tile = atomicAdd(g.tile_counter, 1u);
        __syncthreads();
        const unsigned wg = s_tile;
        if (wg >= nwg) break;
        // grouped ordering as tile_of_block, without the per-XCD id remap (tiles are not tied to an XCD here)
        const unsigned per_group = GROUP_M * g.tiles_n;
        const unsigned group = wg / per_group;
        const unsigned first_m = group * GROUP_M;
        const unsigned gsz = (g.tiles_m - first_m) < GROUP_M ? (g.tiles_m - first_m) : GROUP_M;
        const unsigned in_group = wg - group * per_group;
        w8_tile<false>(g, first_m + in_group % gsz, in_group / gsz, lds, lds + 2 * A_TILE);
        __syncthreads();  // LDS and s_tile are reused
    }
}

static int g_rowmap = -1;

static int launch_dgemm_impl(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                             const double* B, size_t ldb, double beta, double* C, size_t ldc, const GemmEpilogue* ep,
                             bool ta, bool tb);

int launch_dgemm(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                 const double* B, size_t ldb, double beta, double* C, size_t ldc) {
    return launch_dgemm_impl(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, false, false);
}

int launch_dgemm_epilogue(Context* c, size_t m, size_t n, size_t k, const double* A, size_t lda, const double* B,
                          size_t ldb, double* C, size_t ldc, const GemmEpilogue& ep) {
    return launch_dgemm_impl(c, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc, &ep, false, false);
}

// C (m x n) = alpha * op(A) * op(B) + beta * C with op(X) = X' when tX: A is then stored k x m (lda >= k),
// B stored n x k (ldb >= n).  Both transposed at once is not instantiated (callers materialise one operand).
int launch_dgemm_trans(Context* c, bool ta, bool tb, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                       const double* B, size_t ldb, double beta, double* C, size_t ldc) {
    if (ta && tb) return fail(RMHIP_ERR_UNSUPPORTED, "dgemm: A' * B' is not instantiated");
    return launch_dgemm_impl(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, ta, tb);
}

template <bool EDGE, bool EPI, bool TA, bool TB, bool PRE = false>
static void launch_variant(Context* c, unsigned blocks, unsigned splits, size_t lds_bytes, size_t max_lds, const GemmArgs& g) {
    c->ensure_max_lds((const void*)k_dgemm<EDGE, EPI, TA, TB, PRE>, max_lds);
    hipLaunchKernelGGL((k_dgemm<EDGE, EPI, TA, TB, PRE>), dim3(blocks, splits), dim3(256), lds_bytes, c->stream, g);
}

// C = alpha * (P_0 + P_1 + ... + P_{S-1}) + beta * C, partials m x n dense (ld m), summed in split order
__global__ void __launch_bounds__(256) k_reduce_splits(const double* __restrict__ P, size_t mn, size_t m, unsigned splits,
                                                       double alpha, double beta, double* __restrict__ C, size_t ldc) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += (size_t)gridDim.x * 256) {
        double s = P[i];
        for (unsigned z = 1; z < splits; ++z) s += P[i + (size_t)z * mn];
        double* dst = C + (i % m) + (i / m) * ldc;
        double v = alpha * s;
        if (beta != 0.0) v = beta * (*dst) + v;
        *dst = v;
    }
}

static int launch_dgemm_impl(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                             const double* B, size_t ldb, double beta, double* C, size_t ldc, const GemmEpilogue* ep,
                             bool ta, bool tb) {
    if (m == 0 || n == 0) return RMHIP_OK;
    if (m > 0xffffffffULL || n > 0xffffffffULL || k > 0xffffffffULL)
        return fail(RMHIP_ERR_UNSUPPORTED, "dgemm: dimension exceeds 2^32");
    if (g_rowmap < 0) {
        const char* v = std::getenv("RMHIP_MFMA_ROWMAP");
        g_rowmap = (v && *v == '1') ? 1 : 0;
    }
    GemmArgs g;
    g.A = A;
    g.B = B;
    g.C = C;
    g.lda = lda;
    g.ldb = ldb;
    g.ldc = ldc;
    g.m = (unsigned)m;
    g.n = (unsigned)n;
    g.k = (unsigned)k;
    g.tiles_m = (unsigned)((m + BM - 1) / BM);
    g.tiles_n = (unsigned)((n + BN - 1) / BN);
    g.alpha = alpha;
    g.beta = beta;
    g.rowmap = g_rowmap;
    g.vec_a = ((((uintptr_t)A & 15) == 0) && (lda % 2 == 0)) ? 1 : 0;
    g.vec_b = ((((uintptr_t)B & 15) == 0) && (ldb % 2 == 0)) ? 1 : 0;
    if (ep) g.ep = *ep;
    else g.ep = GemmEpilogue{0, 1.0, 0.0, nullptr, nullptr, 0.0, 0.0, 0.0, nullptr};
    const size_t lds_base = (size_t)(2 * A_TILE + 2 * B_TILE) * sizeof(double);
    static long dev_pad = -1;  // developer knob: extra dynamic LDS for every launch (blocks per CU experiments)
    if (dev_pad < 0) {
        const char* v = std::getenv("RMHIP_GEMM_LDS_PAD");
        dev_pad = v ? std::atol(v) : 0;
    }
    const size_t lds_bytes = lds_base + (c->gemm_lds_pad ? c->gemm_lds_pad : (size_t)dev_pad);
    const bool fast = (m % BM == 0) && (n % BN == 0) && (k % BK == 0) && k > 0 && (lda % 2 == 0) && (ldb % 2 == 0) &&
                      (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
    const unsigned blocks = g.tiles_m * g.tiles_n;
    const size_t kMaxLds = 128 * 1024;  // room for the look-ahead pad
    if (lds_bytes > kMaxLds) return fail(RMHIP_ERR_INVALID, "dgemm: LDS pad too large");
    // Split-K: few output tiles but a long k (A'*A of a tall matrix, dot-product-like shapes) would leave most CUs
    // idle; give every CU about two blocks.  Chunks are multiples of 1024 so the unguarded kernel stays eligible.
    unsigned splits = 1;
    g.k_chunk = (unsigned)k;
    g.c_split_stride = 0;
    g.tile_counter = nullptr;
    g.avoid_xcc = nullptr;
    g.yield_word = (c->gemm_lds_pad != 0) ? c->gemm_yield_word : nullptr;  // update streams of the two-level LU only
    g.prio = (c->in_lookahead && c->gemm_lds_pad == 0 && c->gemm_chain_prio) ? 1 : 0;  // the look-ahead LU's main stream
    g.announce = (c->in_lookahead && c->gemm_lds_pad == 0) ? c->gemm_announce : nullptr;     // ... whose blocks ask the update blocks on their CU to pause
    std::shared_ptr<Allocation> partials;
    // (round 3: from k = 1024 - slices of multiples of 128 - outside the LU.  A block walks its k range at about 1 us per 16 columns
    // - one memory latency per tile with nothing else resident - so few blocks with a long k are latency bound whatever the tile:
    // 4096 x 100 with k = 4096 (32 tiles) 547 -> 97 us, 32 x 512 with k = 8192 on one slice 1100 us.)
    const size_t split_min_k = c->gemm_split_min_k ? c->gemm_split_min_k : (c->in_lookahead ? 8192 : 1024);
    if (!ep && blocks * 4 <= (unsigned)c->num_cus && k >= split_min_k) {
        const size_t want = (2 * (size_t)c->num_cus + blocks - 1) / blocks;
        size_t chunk = (k + want - 1) / want;
        const size_t gran = k >= 8192 ? 1024 : (k >= 2048 ? 256 : 128);
        chunk = ((chunk + gran - 1) / gran) * gran;
        splits = (unsigned)((k + chunk - 1) / chunk);
        if (splits > 1) {
            // a caller that runs several streams (the two-level LU) lends a workspace per stream: a pooled block released at the end of
            // this call could be handed to another stream's call while this one's kernels are still queued
            double* ws = nullptr;
            if (c->gemm_split_ws && c->gemm_split_ws_elems >= (size_t)splits * m * n) ws = c->gemm_split_ws;
            else RMHIP_TRY(c->alloc_device((size_t)splits * m * n, &partials));
            g.k_chunk = (unsigned)chunk;
            g.C = ws ? ws : partials->ptr;
            g.ldc = m;
            g.c_split_stride = (unsigned long long)m * n;
            g.alpha = 1.0;
            g.beta = 0.0;
        } else {
            splits = 1;
        }
    }
    const bool fast_k = fast && (splits == 1 || k % g.k_chunk == 0);
    // C <- C - A*B (the LU's updates): preload C into the accumulators (RMHIP_GEMM_PRELOAD=0 keeps the read at the end)
    static int preload_on = -1;
    if (preload_on < 0) {
        const char* v = std::getenv("RMHIP_GEMM_PRELOAD");
        preload_on = (v && *v == '0') ? 0 : 1;
    }
    const bool preload = preload_on && !ep && !ta && !tb && splits == 1 && alpha == -1.0 && beta == 1.0 && k > 0;
    // small-tile kernel: the big tiles would cover at most half of the CUs and k is short (RMHIP_GEMM_SMALL=0 disables).  Not on
    // the look-ahead update stream (gemm_lds_pad != 0): its blocks must stay too big to share a CU with a panel block.
    static int small_on = -1;
    if (small_on < 0) {
        const char* v = std::getenv("RMHIP_GEMM_SMALL");
        small_on = (v && *v == '0') ? 0 : 1;
    }
    static long small_force = -1, small_pad = 0;  // developer knobs: RMHIP_GEMM_SMALL_FORCE=1 (any size), RMHIP_GEMM_SMALL_PAD=<bytes of extra LDS>
    if (small_force < 0) {
        small_force = std::getenv("RMHIP_GEMM_SMALL_FORCE") ? std::atol(std::getenv("RMHIP_GEMM_SMALL_FORCE")) : 0;
        small_pad = std::getenv("RMHIP_GEMM_SMALL_PAD") ? std::atol(std::getenv("RMHIP_GEMM_SMALL_PAD")) : 0;
    }
    // shapes that are not whole tiles - any m, n, k, leading dimensions, 8-byte aligned bases - run the same kernels with clamped
    // operand loads, a zeroed k tail and checked stores (GUARD).  RMHIP_GEMM_GUARD=0 sends them to the element-checking k_dgemm as
    // before (which keeps A * B' and the epilogue of a transposed product).
    static const int guard_on = std::getenv("RMHIP_GEMM_GUARD") ? std::atoi(std::getenv("RMHIP_GEMM_GUARD")) : 1;
    const bool vec_ok = (lda % 2 == 0) && (ldb % 2 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
    static const int guard_lu = std::getenv("RMHIP_GEMM_GUARD_LU") ? std::atoi(std::getenv("RMHIP_GEMM_GUARD_LU")) : 1;  // A/B: guarded tiles inside the look-ahead LU
    const bool guard_ok = guard_on && !tb && k >= 1 && (guard_lu || (!c->in_lookahead && c->gemm_lds_pad == 0));
    // (skinny products - at most 64 rows or columns of output - waste half of every 128 x 128 tile or more: 100000 x 50 x 50 52 -> 30 us,
    // 200000 x 16 x 16 34 -> 20 us on the 64 x 64 tiles, whatever the block count)
    const bool skinny = (m <= (size_t)SM || n <= (size_t)SN) && k < 1024 && !c->in_lookahead;  // (longer k: split-K above)
    // The look-ahead LU's update streams run their few-tile products - the dgemm steps of the W-wide triangular solves, 14-112 tiles of
    // 128 x 128 for 0.1-0.2 ms each - on the 64 x 64 tiles as well: four times the blocks, 16384 69.1 -> 67.6 ms, 8192 20.6 -> 20.2
    // (RMHIP_LU_SMALL_UPD = largest k, 0 = off; _BLOCKS = most 128 x 128 tiles)
    static const long small_upd = std::getenv("RMHIP_LU_SMALL_UPD") ? std::atol(std::getenv("RMHIP_LU_SMALL_UPD")) : 1024;
    static const long small_upd_blocks = std::getenv("RMHIP_LU_SMALL_UPD_BLOCKS") ? std::atol(std::getenv("RMHIP_LU_SMALL_UPD_BLOCKS")) : 128;
    const bool upd_small = small_upd > 0 && c->gemm_lds_pad != 0 && c->in_lookahead && (long)k <= small_upd && (long)blocks <= small_upd_blocks;
    const bool small_shape = !ep && !ta && !tb && splits == 1 && (c->gemm_lds_pad == 0 || small_force || upd_small) && k > 0 &&
                             ((k <= 1024 && (size_t)blocks * 2 <= (size_t)c->num_cus) || skinny || small_force || upd_small);
    const bool small_whole = (m % SM == 0) && (n % SN == 0) && (k % BK == 0) && vec_ok;
    if (small_on && small_shape && (small_whole || guard_ok)) {
        g.tiles_m = (unsigned)((m + SM - 1) / SM);
        g.tiles_n = (unsigned)((n + SN - 1) / SN);
        const dim3 sgrid(g.tiles_m * g.tiles_n);
        if (small_whole) {
            if (preload) hipLaunchKernelGGL(k_dgemm_small<true>, sgrid, dim3(256), (size_t)small_pad, c->stream, g);
            else hipLaunchKernelGGL(k_dgemm_small<false>, sgrid, dim3(256), (size_t)small_pad, c->stream, g);
        } else {
            if (preload) hipLaunchKernelGGL((k_dgemm_small<true, true>), sgrid, dim3(256), (size_t)small_pad, c->stream, g);
            else hipLaunchKernelGGL((k_dgemm_small<false, true>), sgrid, dim3(256), (size_t)small_pad, c->stream, g);
        }
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    // eight-wave tile: on the look-ahead's update stream (one padded block per CU), and for every plain unguarded product
    // (scripts/gemm_w8_ab.py, TFLOP/s four-wave -> eight-wave, both with the pipelined k loop: 4096^3 68.6 -> 71.9, 8192^3 69.5 ->
    // 72.3, 4096^2 x 16384 68.8 -> 72.5, 8192^2 x 1024 68.4 -> 70.7, 12288 x 4096 x 2048 68.5 -> 71.4; equal at 2048^3 and at
    // 16384^2 x 128 / 256: four pipelined waves per SIMD leave the matrix pipe fewer gaps than two) - but not on the main stream
    // inside the look-ahead LU, whose blocks must fit beside the update stream's.
    // RMHIP_GEMM_W8: 0 never, 2 always (A/B).
    static int w8_mode = -1;
    if (w8_mode < 0) {
        const char* v = std::getenv("RMHIP_GEMM_W8");
        w8_mode = v ? std::atoi(v) : 1;
    }
    if (fast_k && splits == 1 && !ep && !ta && !tb && c->gemm_tile_counters && c->gemm_lds_pad != 0 && w8_mode != 0 &&
        c->gemm_counter_next < c->gemm_counter_cap) {
        // update stream of the look-ahead LU while the panels sit on one XCD: persistent eight-wave kernel that stays off it
        g.tile_counter = c->gemm_tile_counters + c->gemm_counter_next++;
        g.avoid_xcc = c->gemm_avoid_xcc;
        static const long grid_cap = std::getenv("RMHIP_GEMM_W8P_GRID") ? std::atol(std::getenv("RMHIP_GEMM_W8P_GRID")) : 0;  // A/B: fewer persistent workgroups than CUs
        unsigned grid = (unsigned)c->num_cus < blocks ? (unsigned)c->num_cus : blocks;
        if (grid_cap > 0 && grid > (unsigned)grid_cap) grid = (unsigned)grid_cap;
        c->ensure_max_lds((const void*)k_dgemm_w8p, kMaxLds);
        hipLaunchKernelGGL(k_dgemm_w8p, dim3(grid), dim3(512), lds_bytes, c->stream, g);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    const bool w8_fast = fast && g.k_chunk % BK == 0;  // every k slice a whole number of tiles
    auto reduce_splits = [&]() -> int {
        if (splits > 1) {
            const size_t mn = m * n;
            const unsigned rgrid = (unsigned)((mn + 255) / 256 < (size_t)c->num_cus * 4 ? (mn + 255) / 256 : (size_t)c->num_cus * 4);
            hipLaunchKernelGGL(k_reduce_splits, dim3(rgrid), dim3(256), 0, c->stream, (const double*)g.C, mn, m, splits, alpha, beta, C, ldc);
            c->tel.kernel_launches++;
            RMHIP_HIP_CHECK(hipGetLastError());
        }
        return RMHIP_OK;
    };
    if (w8_fast && ep && !ta && !tb && w8_mode != 0 && !c->in_lookahead && c->gemm_lds_pad == 0) {
        // matmul_epilogue on the eight-wave tile (splits == 1 whenever there is an epilogue)
        if (g.ep.flags & EP_POW) {
            c->ensure_max_lds((const void*)k_dgemm_w8<false, false, false, 2>, kMaxLds);
            hipLaunchKernelGGL((k_dgemm_w8<false, false, false, 2>), dim3(blocks), dim3(512), lds_bytes, c->stream, g);
        } else {
            c->ensure_max_lds((const void*)k_dgemm_w8<false, false, false, 1>, kMaxLds);
            hipLaunchKernelGGL((k_dgemm_w8<false, false, false, 1>), dim3(blocks), dim3(512), lds_bytes, c->stream, g);
        }
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    if (w8_fast && !ep && ta && !tb && w8_mode != 0 && !c->in_lookahead && c->gemm_lds_pad == 0) {
        // A' * B (transpose views, syrk, the Gram matrices of covariance and least squares): 70.7 -> 72.5 TFLOP/s at 8192^3 as
        // well.  (A * B' measured 69.8 against 70.2 in this form and stays on k_dgemm.)
        // Split-K (tall A'*A: few output tiles, long k) runs the same kernel with blockIdx.y over the k slices.
        c->ensure_max_lds((const void*)k_dgemm_w8<false, true, false>, kMaxLds);
        hipLaunchKernelGGL((k_dgemm_w8<false, true, false>), dim3(blocks, splits), dim3(512), lds_bytes, c->stream, g);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return reduce_splits();
    }
    if (w8_fast && splits > 1 && !ep && !ta && !tb && w8_mode != 0 && !c->in_lookahead && c->gemm_lds_pad == 0) {
        c->ensure_max_lds((const void*)k_dgemm_w8<false>, kMaxLds);
        hipLaunchKernelGGL(k_dgemm_w8<false>, dim3(blocks, splits), dim3(512), lds_bytes, c->stream, g);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return reduce_splits();
    }
    if (fast_k && splits == 1 && !ep && !ta && !tb &&
        ((w8_mode == 1 && (c->gemm_lds_pad != 0 || !c->in_lookahead)) || w8_mode == 2)) {
        if (preload && g.yield_word) {  // the two-level LU's update streams: the variant that checks the yield word
            c->ensure_max_lds((const void*)k_dgemm_w8<true, false, false, 0, true>, kMaxLds);
            hipLaunchKernelGGL((k_dgemm_w8<true, false, false, 0, true>), dim3(blocks), dim3(512), lds_bytes, c->stream, g);
        } else if (preload) {
            c->ensure_max_lds((const void*)k_dgemm_w8<true>, kMaxLds);
            hipLaunchKernelGGL(k_dgemm_w8<true>, dim3(blocks), dim3(512), lds_bytes, c->stream, g);
        } else {
            c->ensure_max_lds((const void*)k_dgemm_w8<false>, kMaxLds);
            hipLaunchKernelGGL(k_dgemm_w8<false>, dim3(blocks), dim3(512), lds_bytes, c->stream, g);
        }
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    // (inside the look-ahead LU only the update stream - the one with the LDS pad - runs eight-wave blocks, as for whole tiles)
    if (!fast && guard_ok && w8_mode != 0 && (splits == 1 || g.k_chunk % BK == 0) && !(ep && ta) && (c->gemm_lds_pad != 0 || !c->in_lookahead)) {
        // guarded eight-wave tile (plain or transposed A; plain, preloaded-C or epilogue store; split-K slices are whole tiles except the last)
        const dim3 ggrid(blocks, splits);
#define RMHIP_W8G(...)                                                                         \
    do {                                                                                       \
        c->ensure_max_lds((const void*)k_dgemm_w8g<__VA_ARGS__>, kMaxLds);                     \
        hipLaunchKernelGGL((k_dgemm_w8g<__VA_ARGS__>), ggrid, dim3(512), lds_bytes, c->stream, g); \
    } while (0)
        if (ep) {
            if (g.ep.flags & EP_POW) RMHIP_W8G(false, false, 2);
            else RMHIP_W8G(false, false, 1);
        } else if (ta) {
            RMHIP_W8G(false, true, 0);
        } else if (preload) {
            RMHIP_W8G(true, false, 0);
        } else {
            RMHIP_W8G(false, false, 0);
        }
#undef RMHIP_W8G
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return reduce_splits();
    }
    if (ep) {
        if (ta || tb) return fail(RMHIP_ERR_UNSUPPORTED, "dgemm: epilogue with transposed operands is not instantiated");
        launch_variant<true, true, false, false>(c, blocks, splits, lds_bytes, kMaxLds, g);
    } else if (ta) {
        if (fast_k) launch_variant<false, false, true, false>(c, blocks, splits, lds_bytes, kMaxLds, g);
        else launch_variant<true, false, true, false>(c, blocks, splits, lds_bytes, kMaxLds, g);
    } else if (tb) {
        if (fast_k) launch_variant<false, false, false, true>(c, blocks, splits, lds_bytes, kMaxLds, g);
        else launch_variant<true, false, false, true>(c, blocks, splits, lds_bytes, kMaxLds, g);
    } else {
        if (fast_k && preload) launch_variant<false, false, false, false, true>(c, blocks, splits, lds_bytes, kMaxLds, g);
        else if (fast_k) launch_variant<false, false, false, false>(c, blocks, splits, lds_bytes, kMaxLds, g);
        else launch_variant<true, false, false, false>(c, blocks, splits, lds_bytes, kMaxLds, g);
    }
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    if (splits > 1) {
        const size_t mn = m * n;
        const unsigned rgrid = (unsigned)((mn + 255) / 256 < (size_t)c->num_cus * 4 ? (mn + 255) / 256 : (size_t)c->num_cus * 4);
        hipLaunchKernelGGL(k_reduce_splits, dim3(rgrid), dim3(256), 0, c->stream, (const double*)g.C, mn, m, splits, alpha, beta, C,
                           ldc);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

}  // namespace rmhip
