// sgemm.hip -- f32 GEMM on the CDNA4 matrix cores for precision-32 providers: C = op(A) * op(B), column-major,
// f32 operands and result in HBM.  `AccelProvider::matmul` (crates/runmat-accelerate-api/src/lib.rs:2375-2381) when the
// provider reports `ProviderPrecision::F32` (lib.rs:815-818); the reference's F32 backend accumulates in f32 too
// (backend/wgpu/shaders/matmul.rs) and its own checks allow 1e-4 relative / 1e-5 absolute
// (src/bin/wgpu_profile.rs:20-21,160-163).
//
// Same block design as dgemm.hip (128x128 block tile, 2x2 waves of 64x64, K step 16, register -> LDS double
// buffering, XCD-aware tile order, roles transposed so that the lane-contiguous MFMA index is the memory-contiguous
// row index of C) with the f32 instruction: v_mfma_f32_16x16x4_f32 = 2048 flop in 32 cycles, i.e. 64 flop/clk/SIMD,
// 157 TFLOP/s at 2.4 GHz -- twice the f64 rate for the same instruction count, so every tile has half the time to
// hide its loads in; the 64 accumulator VGPRs (f64: 128) leave room for more resident blocks per CU instead.
//   LDS (floats): pattern-M tile [k][x] with row stride 144 (144 % 64 == 16: the four k-rows of a ds_read_b32 wave
//   access fall in disjoint 16-bank groups), pattern-K tile [y][k] with row stride 20 (20 * l15 + lq covers the 64
//   banks exactly once).  Staging moves 8-byte pairs.
// Accumulation is f32 in the matrix core: the result differs from the f64 path (RMHIP_F32_MATMUL=f64: widen, dgemm,
// round once -- the CPU's `single` semantics exactly) by at most ~k * eps32 * sum|a||b|; tests state the bound.
#include "common.h"

namespace rmhip {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef v2f v2fu __attribute__((aligned(4)));  // the same pair on a 4-byte aligned address (guarded kernel)

namespace sg {
static constexpr int BM = 128, BN = 128, BK = 16;
static constexpr int SA = BM + 16;  // pattern M row stride (floats)
static constexpr int SB = BK + 4;   // pattern K row stride (floats)
static constexpr int TILE = (BK * SA > BN * SB) ? BK * SA : BN * SB;  // floats; either pattern fits either buffer
#ifndef SGEMM_GROUP_M
#define SGEMM_GROUP_M 8
#endif
static constexpr int GROUP_M = SGEMM_GROUP_M;
}  // namespace sg

struct SgemmArgs {
    const float* A;
    const float* B;
    float* C;
    unsigned long long lda, ldb, ldc;
    unsigned m, n, k;
    unsigned tiles_m, tiles_n;
    int vec_a, vec_b;  // EDGE kernel: 8-byte loads are legal for A / B (aligned base, even leading dimension)
    // split-K (few output tiles, long k -- A'*A of a tall matrix): blockIdx.y owns k in [y*k_chunk, min(k, (y+1)*k_chunk))
    // and writes its partial product to C + y*c_split_stride; k_sreduce_splits adds the partials in split order
    unsigned k_chunk;
    unsigned long long c_split_stride;
};

__device__ __forceinline__ void sg_tile_of_block(const SgemmArgs& g, unsigned& tm, unsigned& tn) {
    using namespace sg;
    const unsigned nwg = g.tiles_m * g.tiles_n;
    const unsigned b = blockIdx.x;
    const unsigned xcd = b & 7u, idx = b >> 3;  // block b runs on XCD b % 8: give each XCD a contiguous id range
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const unsigned per_group = GROUP_M * g.tiles_n;
    const unsigned group = wg / per_group;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = (g.tiles_m - first_m) < GROUP_M ? (g.tiles_m - first_m) : GROUP_M;
    const unsigned in_group = wg - group * per_group;
    tm = first_m + in_group % gsz;
    tn = in_group / gsz;
}

#ifndef SGEMM_BLOCKS_PER_CU
#define SGEMM_BLOCKS_PER_CU 2
#endif

// EDGE = false: m % 128 == 0, n % 128 == 0, k % 16 == 0, even leading dimensions, 8-byte aligned bases.
// TA / TB: the operand is stored transposed (RunMat's transpose views), exactly as in k_dgemm.
template <bool EDGE, bool TA, bool TB>
__global__ void __launch_bounds__(256, SGEMM_BLOCKS_PER_CU) k_sgemm(const SgemmArgs g) {
    using namespace sg;
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];
    float* As = lds;             // [2][TILE]
    float* Bs = lds + 2 * TILE;  // [2][TILE]

    unsigned tm, tn;
    sg_tile_of_block(g, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned kbeg = blockIdx.y * g.k_chunk;
    const unsigned klen = (g.k - kbeg) < g.k_chunk ? (g.k - kbeg) : g.k_chunk;
    const float* const Ab = TA ? g.A + kbeg : g.A + (size_t)kbeg * g.lda;
    const float* const Bb = TB ? g.B + (size_t)kbeg * g.ldb : g.B + kbeg;
    float* const Cb = g.C + (size_t)blockIdx.y * g.c_split_stride;

    const int p_xp = t & 63;  // pattern M: pair index along the contiguous tile dimension
    const int p_kc = t >> 6;  //            k = p_kc + 4*p
    const int q_kp = t & 7;   // pattern K: k pair index
    const int q_y = t >> 3;   //            y = q_y + 32*p

    v2f ra[4], rb[4];

    auto fetchM = [&](const float* ptr, unsigned long long ld, unsigned x0, unsigned xlim, unsigned k0, int vec, v2f* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned kk = k0 + p_kc + 4 * p;
            const unsigned xx = x0 + 2 * p_xp;
            if (!EDGE) {
                r[p] = *(const v2f*)(ptr + (size_t)kk * ld + xx);
            } else {
                v2f v = {0.f, 0.f};
                if (kk < klen) {
                    const float* src = ptr + (size_t)kk * ld + xx;
                    if (vec && xx + 1 < xlim) {
                        v = *(const v2f*)src;
                    } else {
                        if (xx < xlim) v.x = src[0];
                        if (xx + 1 < xlim) v.y = src[1];
                    }
                }
                r[p] = v;
            }
        }
    };
    auto fetchK = [&](const float* ptr, unsigned long long ld, unsigned y0, unsigned ylim, unsigned k0, int vec, v2f* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned yy = y0 + q_y + 32 * p;
            const unsigned kk = k0 + 2 * q_kp;
            if (!EDGE) {
                r[p] = *(const v2f*)(ptr + (size_t)yy * ld + kk);
            } else {
                v2f v = {0.f, 0.f};
                if (yy < ylim) {
                    const float* src = ptr + (size_t)yy * ld + kk;
                    if (vec && kk + 1 < klen) {
                        v = *(const v2f*)src;
                    } else {
                        if (kk < klen) v.x = src[0];
                        if (kk + 1 < klen) v.y = src[1];
                    }
                }
                r[p] = v;
            }
        }
    };
    auto stashM = [&](float* tile, const v2f* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2f*)(tile + (p_kc + 4 * p) * SA + 2 * p_xp) = r[p];
    };
    auto stashK = [&](float* tile, const v2f* r) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2f*)(tile + (q_y + 32 * p) * SB + 2 * q_kp) = r[p];
    };
    auto fetch = [&](unsigned k0) {
        if (TA) fetchK(Ab, g.lda, m0, g.m, k0, g.vec_a, ra);
        else fetchM(Ab, g.lda, m0, g.m, k0, g.vec_a, ra);
        if (TB) fetchM(Bb, g.ldb, n0, g.n, k0, g.vec_b, rb);
        else fetchK(Bb, g.ldb, n0, g.n, k0, g.vec_b, rb);
    };
    auto stash = [&](int buf) {
        if (TA) stashK(As + buf * TILE, ra);
        else stashM(As + buf * TILE, ra);
        if (TB) stashM(Bs + buf * TILE, rb);
        else stashK(Bs + buf * TILE, rb);
    };

    v4f acc[4][4];  // [tj (n)][ti (m)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = v4f{0.f, 0.f, 0.f, 0.f};

    const unsigned ktiles = (klen + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();

    const int a_off = TA ? (wm * 64 + l15) * SB + lq : lq * SA + wm * 64 + l15;
    const int b_off = TB ? lq * SA + wn * 64 + l15 : (wn * 64 + l15) * SB + lq;
    constexpr int A_KSTEP = TA ? 4 : 4 * SA, A_ISTEP = TA ? 16 * SB : 16;
    constexpr int B_KSTEP = TB ? 4 * SA : 4, B_JSTEP = TB ? 16 : 16 * SB;

    for (unsigned kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) fetch((kt + 1) * BK);
        const float* a = As + cur * TILE + a_off;
        const float* b = Bs + cur * TILE + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[j * B_JSTEP + kk * B_KSTEP];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[j][i], 0, 0, 0);
            if (kk == BK / 4 - 1 && kt + 1 < ktiles) stash(cur ^ 1);
        }
        __syncthreads();
    }

    // D[r][c] -> C[m = c][n = r]; c = lane & 15, r = 4 * (lane >> 4) + reg (the f32 16x16x4 accumulator layout)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mm = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned nn = n0 + wn * 64 + j * 16 + 4 * lq + r;
                if (!EDGE || (mm < g.m && nn < g.n)) Cb[(size_t)nn * g.ldc + mm] = acc[j][i][r];
            }
        }
}

// ---- eight-wave form of the 128 x 128 tile with the software-pipelined k loop (dgemm.hip: k_dgemm_w8) ---------------------------
// 2 (m) x 4 (n) waves of 64 x 32; fragments of step kk+1 are read under step kk, the next tile is stashed during step 2 from
// registers loaded two tiles earlier, ONE raw LDS-only barrier per tile before step 3, and step 3 already reads the next tile's
// first fragments.  An f32 tile is 1024 MFMA cycles per wave, so the bubbles around the barrier weigh four times what they do in
// the f64 kernel: with several pipelined waves per SIMD the matrix pipe stays fed.  Plain operands, unguarded shapes, one split.
// Same k-ordered chain per element as k_sgemm: bit-identical results.
// (86 VGPRs: five waves per SIMD by registers, four - two blocks per CU - by LDS.  Capped at 80 for a third block it spills and
// runs 122 instead of 136 TFLOP/s.)
// GUARD (k_sgemm_w8g): any m, n, k, leading dimensions, 4-byte aligned bases - as the guarded f64 tile (dgemm.hip: w8_tile): rows / columns
// beyond the matrix re-read its last ones, 8-byte loads on 4-byte aligned addresses, scalar loads for the pairs that straddle the
// matrix edge (in the hot loop only in blocks on the lower edge of an odd m), a zero k tail in the last tile, checked stores.
template <bool TA, bool GUARD>  // TA: A is stored transposed (k contiguous per tile row), staged with B's pattern as in k_sgemm
__device__ __forceinline__ void sgemm_w8_body(const SgemmArgs& g, float* lds) {
    using namespace sg;
    float* As = lds;
    float* Bs = lds + 2 * TILE;
    unsigned tm, tn;
    sg_tile_of_block(g, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;  // wn 0..3: 32 columns each
    const int l15 = lane & 15, lq = lane >> 4;
    const int p_xp = t & 63, p_kc = t >> 6;  // A (pattern M): pair along m, k = p_kc + 8*p
    const int q_kp = t & 7, q_y = t >> 3;    // B (pattern K): pair along k, y = q_y + 64*p
    auto rowY = [&](unsigned r, unsigned lim) { return (GUARD && r >= lim) ? lim - 1 : r; };
    const unsigned a_r0 = rowY(m0 + 2 * p_xp, g.m);
    const bool a_single = GUARD && !TA && a_r0 + 1 >= g.m;            // the matrix's last row alone (odd m) or a clamped pair
    const bool a_edge = GUARD && !TA && m0 + BM > g.m;  // uniform: such threads exist in this block (an even m too: its clamped threads sit on the last row and must not read a pair)
    const float* const Ap = TA ? g.A + 2 * q_kp : g.A + a_r0;
    const float* const Bp = g.B + 2 * q_kp;
    size_t a_row[2], b_row[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a_row[p] = (size_t)rowY(m0 + q_y + 64 * p, g.m) * g.lda;
        b_row[p] = (size_t)rowY(n0 + q_y + 64 * p, g.n) * g.ldb;
    }
    auto ld2 = [&](const float* q) -> v2f { return GUARD ? (v2f)(*(const v2fu*)q) : *(const v2f*)q; };
    auto ldM = [&](const float* q) -> v2f {
        if (a_edge && a_single) return v2f{*q, 0.f};
        return ld2(q);
    };
    v2f ra[2], rb[2], ra2[2], rb2[2];
    auto fetch_into = [&](unsigned k0, v2f* pa, v2f* pb) {
        if (GUARD && k0 + BK > g.k) {  // the last, partial k tile (uniform)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const unsigned km = k0 + p_kc + 8 * p, kp = k0 + 2 * q_kp;
                auto ldK = [&](const float* pair) -> v2f {
                    if (kp + 1 < g.k) return ld2(pair);
                    if (kp < g.k) return v2f{*pair, 0.f};
                    return v2f{0.f, 0.f};
                };
                pa[p] = TA ? ldK(Ap + a_row[p] + k0) : (km < g.k ? ldM(Ap + (size_t)km * g.lda) : v2f{0.f, 0.f});
                pb[p] = ldK(Bp + b_row[p] + k0);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            pa[p] = TA ? ld2(Ap + a_row[p] + k0) : ldM(Ap + (size_t)(k0 + p_kc + 8 * p) * g.lda);
            pb[p] = ld2(Bp + b_row[p] + k0);
        }
    };
    auto stash_from = [&](int buf, const v2f* pa, const v2f* pb) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (TA) *(v2f*)(As + buf * TILE + (q_y + 64 * p) * SB + 2 * q_kp) = pa[p];
            else *(v2f*)(As + buf * TILE + (p_kc + 8 * p) * SA + 2 * p_xp) = pa[p];
            *(v2f*)(Bs + buf * TILE + (q_y + 64 * p) * SB + 2 * q_kp) = pb[p];
        }
    };
    v4f acc[2][4];  // [tj (n)][ti (m)]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = v4f{0.f, 0.f, 0.f, 0.f};
    const unsigned ktiles = GUARD ? (g.k + BK - 1) / BK : g.k / BK;
    auto clampt = [&](unsigned tt) { return (tt < ktiles ? tt : ktiles - 1) * BK; };
    fetch_into(0, ra, rb);
    stash_from(0, ra, rb);
    fetch_into(clampt(1), ra, rb);
    fetch_into(clampt(2), ra2, rb2);
    __syncthreads();
    const int a_off = TA ? (wm * 64 + l15) * SB + lq : lq * SA + wm * 64 + l15;
    const int b_off = (wn * 32 + l15) * SB + lq;
    constexpr int A_KSTEP = TA ? 4 : 4 * SA, A_ISTEP = TA ? 16 * SB : 16;
    auto frags = [&](const float* a, const float* b, int kk, float (&af)[4], float (&bf)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = a[kk * A_KSTEP + i * A_ISTEP];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = b[j * 16 * SB + kk * 4];
    };
    auto mma = [&](const float (&af)[4], const float (&bf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[j][i], 0, 0, 0);
    };
    float af0[4], bf0[2], af1[4], bf1[2];
    frags(As + a_off, Bs + b_off, 0, af0, bf0);
    auto step = [&](unsigned kt, v2f* pa, v2f* pb) {
        const int cur = kt & 1;
        const float* a = As + cur * TILE + a_off;
        const float* b = Bs + cur * TILE + b_off;
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
        stash_from(cur ^ 1, pa, pb);  // tile kt + 1 (the last tiles stash / fetch a copy nobody reads: no branches here)
        fetch_into(clampt(kt + 3), pa, pb);
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
        frags(As + (cur ^ 1) * TILE + a_off, Bs + (cur ^ 1) * TILE + b_off, 0, af0, bf0);
        mma(af1, bf1);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        // the prefetched fragments are "used" here: the wait the compiler owes them lands behind these MFMAs
        asm volatile("" : "+v"(af0[0]), "+v"(af0[1]), "+v"(af0[2]), "+v"(af0[3]), "+v"(bf0[0]), "+v"(bf0[1]));
    };
    for (unsigned kt = 0; kt < ktiles; kt += 2) {
        step(kt, ra, rb);
        if (kt + 1 < ktiles) step(kt + 1, ra2, rb2);
    }
    // D[r][c] -> C[m = c][n = r]; c = lane & 15, r = 4 * (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned mm = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned nn = n0 + wn * 32 + j * 16 + 4 * lq + r;
                if (!GUARD || (mm < g.m && nn < g.n)) g.C[(size_t)nn * g.ldc + mm] = acc[j][i][r];
            }
        }
}
template <bool TA>
__global__ void __launch_bounds__(512) k_sgemm_w8(const SgemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[4 * sg::TILE];
    sgemm_w8_body<TA, false>(g, lds);
}
template <bool TA>
__global__ void __launch_bounds__(512, 4) k_sgemm_w8g(const SgemmArgs g) {  // four waves per SIMD = two blocks per CU, as the plain form
    __shared__ __attribute__((aligned(16))) float lds[4 * sg::TILE];
    sgemm_w8_body<TA, true>(g, lds);
}

template <bool EDGE, bool TA, bool TB>
static void sg_launch(Context* c, unsigned blocks, unsigned splits, const SgemmArgs& g) {
    hipLaunchKernelGGL((k_sgemm<EDGE, TA, TB>), dim3(blocks, splits), dim3(256), 0, c->stream, g);
}

// C = P_0 + P_1 + ... + P_{S-1} (partials m x n dense, ld m), summed in split order in f64 and rounded once
__global__ void __launch_bounds__(256) k_sreduce_splits(const float* __restrict__ P, size_t mn, size_t m, unsigned splits,
                                                        float* __restrict__ C, size_t ldc) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += (size_t)gridDim.x * 256) {
        double s = (double)P[i];
        for (unsigned z = 1; z < splits; ++z) s += (double)P[i + (size_t)z * mn];
        C[(i % m) + (i / m) * ldc] = (float)s;
    }
}

// C (m x n, f32) = op(A) * op(B); op(X) = X' when tX (A then stored k x m, B stored n x k).  k == 0 gives zeros.
int launch_sgemm_trans(Context* c, bool ta, bool tb, size_t m, size_t n, size_t k, const float* A, size_t lda, const float* B,
                       size_t ldb, float* C, size_t ldc) {
    using namespace sg;
    if (m == 0 || n == 0) return RMHIP_OK;
    if (ta && tb) return fail(RMHIP_ERR_UNSUPPORTED, "sgemm: A' * B' is not instantiated");
    if (m > 0xffffffffULL || n > 0xffffffffULL || k > 0xffffffffULL)
        return fail(RMHIP_ERR_UNSUPPORTED, "sgemm: dimension exceeds 2^32");
    SgemmArgs g;
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
    g.vec_a = ((((uintptr_t)A & 7) == 0) && (lda % 2 == 0)) ? 1 : 0;
    g.vec_b = ((((uintptr_t)B & 7) == 0) && (ldb % 2 == 0)) ? 1 : 0;
    const bool fast = (m % BM == 0) && (n % BN == 0) && (k % BK == 0) && k > 0 && g.vec_a && g.vec_b;
    const unsigned blocks = g.tiles_m * g.tiles_n;
    // Split-K as in dgemm.hip: few output tiles but a long k would leave most CUs idle; about two blocks per CU.
    unsigned splits = 1;
    g.k_chunk = (unsigned)k;
    g.c_split_stride = 0;
    std::shared_ptr<Allocation> partials;
    if (blocks * 4 <= (unsigned)c->num_cus && k >= 1024) {  // (from k = 1024, as dgemm.hip: few blocks with a long k are latency bound)
        const size_t want = (2 * (size_t)c->num_cus + blocks - 1) / blocks;
        size_t chunk = (k + want - 1) / want;
        const size_t gran = k >= 8192 ? 1024 : (k >= 2048 ? 256 : 128);  // multiples of the k tile keep the unguarded kernel eligible
        chunk = ((chunk + gran - 1) / gran) * gran;
        splits = (unsigned)((k + chunk - 1) / chunk);
        if (splits > 1) {
            RMHIP_TRY(c->alloc_device(((size_t)splits * m * n + 1) / 2, &partials));
            g.k_chunk = (unsigned)chunk;
            g.C = reinterpret_cast<float*>(partials->ptr);
            g.ldc = m;
            g.c_split_stride = (unsigned long long)m * n;
        } else {
            splits = 1;
        }
    }
    const bool fast_k = fast && (splits == 1 || k % g.k_chunk == 0);
    // shapes that are not whole tiles: the guarded eight-wave tile (RMHIP_GEMM_GUARD=0: the element-checking k_sgemm as before)
    static const int guard_on = std::getenv("RMHIP_GEMM_GUARD") ? std::atoi(std::getenv("RMHIP_GEMM_GUARD")) : 1;
    const bool w8_env_off = std::getenv("RMHIP_SGEMM_W8") && *std::getenv("RMHIP_SGEMM_W8") == '0';
    if (!fast && guard_on && !w8_env_off && !tb && splits == 1 && k >= 1) {
        if (ta) hipLaunchKernelGGL(k_sgemm_w8g<true>, dim3(blocks), dim3(512), 0, c->stream, g);
        else hipLaunchKernelGGL(k_sgemm_w8g<false>, dim3(blocks), dim3(512), 0, c->stream, g);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;
    }
    if (ta) {
        if (fast_k && splits == 1 && !w8_env_off)
            hipLaunchKernelGGL(k_sgemm_w8<true>, dim3(blocks), dim3(512), 0, c->stream, g);  // A' * B (syrk, covariance, transpose views)
        else if (fast_k) sg_launch<false, true, false>(c, blocks, splits, g);
        else sg_launch<true, true, false>(c, blocks, splits, g);
    } else if (tb) {
        if (fast_k) sg_launch<false, false, true>(c, blocks, splits, g);
        else sg_launch<true, false, true>(c, blocks, splits, g);
    } else {
        static int w8_mode = -1;  // RMHIP_SGEMM_W8=0: the four-wave kernel for everything (A/B)
        if (w8_mode < 0) {
            const char* v = std::getenv("RMHIP_SGEMM_W8");
            w8_mode = (v && *v == '0') ? 0 : 1;
        }
        if (fast_k && splits == 1 && w8_mode) hipLaunchKernelGGL(k_sgemm_w8<false>, dim3(blocks), dim3(512), 0, c->stream, g);
        else if (fast_k) sg_launch<false, false, false>(c, blocks, splits, g);
        else sg_launch<true, false, false>(c, blocks, splits, g);
    }
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    if (splits > 1) {
        const size_t mn = m * n;
        const size_t want_blocks = (mn + 255) / 256, cap = (size_t)c->num_cus * 4;
        hipLaunchKernelGGL(k_sreduce_splits, dim3((unsigned)(want_blocks < cap ? want_blocks : cap)), dim3(256), 0, c->stream,
                           reinterpret_cast<const float*>(partials->ptr), mn, m, splits, C, ldc);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

}  // namespace rmhip
