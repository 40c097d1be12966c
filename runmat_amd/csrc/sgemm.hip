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
    if (blocks * 4 <= (unsigned)c->num_cus && k >= 8192) {
        const size_t want = (2 * (size_t)c->num_cus + blocks - 1) / blocks;
        size_t chunk = (k + want - 1) / want;
        chunk = ((chunk + 1023) / 1024) * 1024;  // multiples of 1024 keep the unguarded kernel eligible
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
    if (ta) {
        if (fast_k) sg_launch<false, true, false>(c, blocks, splits, g);
        else sg_launch<true, true, false>(c, blocks, splits, g);
    } else if (tb) {
        if (fast_k) sg_launch<false, false, true>(c, blocks, splits, g);
        else sg_launch<true, false, true>(c, blocks, splits, g);
    } else {
        if (fast_k) sg_launch<false, false, false>(c, blocks, splits, g);
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
