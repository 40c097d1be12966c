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

static constexpr int BM = 128, BN = 128, BK = 16;
static constexpr int SA = BM + 16;  // A tile row stride in doubles ([k][m])
static constexpr int SB = BK + 2;   // B tile row stride in doubles ([n][k])
static constexpr int A_TILE = BK * SA;  // doubles
static constexpr int B_TILE = BN * SB;
static constexpr int GROUP_M = 8;

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
};

// MatmulEpilogue on one output element, order of crates/runmat-accelerate/src/simple_provider.rs:7800-7836
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

// EDGE = false: m % 128 == 0, n % 128 == 0, k % 16 == 0, lda/ldb even, 16-byte aligned bases.
// EPI = true: the MatmulEpilogue variant.  It is a SEPARATE instantiation on purpose: inlining the
// epilogue (pow!) into the plain kernel cost 2.7x (66.8 -> 24.8 TFLOP/s) -- waves in the bloated
// store phase evicted the main loop of their CU pair's instruction cache.
template <bool EDGE, bool EPI>
__global__ void __launch_bounds__(256, 2) k_dgemm(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* As = lds;                    // [2][BK][SA]
    double* Bs = lds + 2 * A_TILE;       // [2][BN][SB]

    unsigned tm, tn;
    tile_of_block(g, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // staging assignment
    const int a_mp = t & 63;   // m pair index (m = 2*a_mp)
    const int a_kc = t >> 6;   // 0..3, k = a_kc + 4*p
    const int b_kp = t & 7;    // k pair index (k = 2*b_kp)
    const int b_n = t >> 3;    // 0..31, n = b_n + 32*p

    const double* Ag = g.A + (size_t)m0 + 2 * a_mp;
    const double* Bg = g.B + (size_t)(n0 + b_n) * g.ldb + 2 * b_kp;

    v2d ra[4], rb[4];

    auto fetch = [&](unsigned k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned kk = k0 + a_kc + 4 * p;
            if (!EDGE) {
                ra[p] = *(const v2d*)(Ag + (size_t)kk * g.lda);
            } else {
                const unsigned mm = m0 + 2 * a_mp;
                v2d v = {0.0, 0.0};
                if (kk < g.k) {
                    const double* src = g.A + (size_t)kk * g.lda + mm;
                    if (g.vec_a && mm + 1 < g.m) {
                        v = *(const v2d*)src;
                    } else {
                        if (mm < g.m) v.x = src[0];
                        if (mm + 1 < g.m) v.y = src[1];
                    }
                }
                ra[p] = v;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!EDGE) {
                rb[p] = *(const v2d*)(Bg + (size_t)(32 * p) * g.ldb + k0);
            } else {
                const unsigned nn = n0 + b_n + 32 * p;
                const unsigned kk = k0 + 2 * b_kp;
                v2d v = {0.0, 0.0};
                if (nn < g.n) {
                    const double* src = g.B + (size_t)nn * g.ldb + kk;
                    if (g.vec_b && kk + 1 < g.k) {
                        v = *(const v2d*)src;
                    } else {
                        if (kk < g.k) v.x = src[0];
                        if (kk + 1 < g.k) v.y = src[1];
                    }
                }
                rb[p] = v;
            }
        }
    };
    auto stash = [&](int buf) {
        double* a = As + buf * A_TILE;
        double* b = Bs + buf * B_TILE;
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2d*)(a + (a_kc + 4 * p) * SA + 2 * a_mp) = ra[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) *(v2d*)(b + (b_n + 32 * p) * SB + 2 * b_kp) = rb[p];
    };

    v4d acc[4][4];  // [tj (n)][ti (m)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = v4d{0.0, 0.0, 0.0, 0.0};

    const unsigned ktiles = (g.k + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();

    const int a_off = lq * SA + wm * 64 + l15;        // + kk*4*SA + ti*16
    const int b_off = (wn * 64 + l15) * SB + lq;      // + tj*16*SB + kk*4

    for (unsigned kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) fetch((kt + 1) * BK);
        const double* a = As + cur * A_TILE + a_off;
        const double* b = Bs + cur * B_TILE + b_off;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = a[kk * 4 * SA + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = b[j * 16 * SB + kk * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < ktiles) stash(cur ^ 1);
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
                dst[i * 4 + r] = g.C + (size_t)nn * g.ldc + mm;
            }
        }
        if (!EPI && g.beta != 0.0) {
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
                } else {
                    v = g.alpha * acc[j][i][r];
                    if (g.beta != 0.0) v = g.beta * prev[e] + v;
                }
                if (ok[e]) *dst[e] = v;
            }
        }
    }
}

static int g_rowmap = -1;

static int launch_dgemm_impl(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                             const double* B, size_t ldb, double beta, double* C, size_t ldc, const GemmEpilogue* ep);

int launch_dgemm(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                 const double* B, size_t ldb, double beta, double* C, size_t ldc) {
    return launch_dgemm_impl(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, nullptr);
}

int launch_dgemm_epilogue(Context* c, size_t m, size_t n, size_t k, const double* A, size_t lda, const double* B,
                          size_t ldb, double* C, size_t ldc, const GemmEpilogue& ep) {
    return launch_dgemm_impl(c, m, n, k, 1.0, A, lda, B, ldb, 0.0, C, ldc, &ep);
}

static int launch_dgemm_impl(Context* c, size_t m, size_t n, size_t k, double alpha, const double* A, size_t lda,
                             const double* B, size_t ldb, double beta, double* C, size_t ldc, const GemmEpilogue* ep) {
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
    const size_t lds_bytes = lds_base + c->gemm_lds_pad;
    const bool fast = (m % BM == 0) && (n % BN == 0) && (k % BK == 0) && k > 0 && (lda % 2 == 0) && (ldb % 2 == 0) &&
                      (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
    const unsigned blocks = g.tiles_m * g.tiles_n;
    static bool attr_set = false;
    const size_t kMaxLds = 128 * 1024;  // room for the look-ahead pad
    if (lds_bytes > kMaxLds) return fail(RMHIP_ERR_INVALID, "dgemm: LDS pad too large");
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_dgemm<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        (void)hipFuncSetAttribute((const void*)k_dgemm<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        (void)hipFuncSetAttribute((const void*)k_dgemm<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        attr_set = true;
    }
    if (ep)
        hipLaunchKernelGGL((k_dgemm<true, true>), dim3(blocks), dim3(256), lds_bytes, c->stream, g);
    else if (fast)
        hipLaunchKernelGGL((k_dgemm<false, false>), dim3(blocks), dim3(256), lds_bytes, c->stream, g);
    else
        hipLaunchKernelGGL((k_dgemm<true, false>), dim3(blocks), dim3(256), lds_bytes, c->stream, g);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

}  // namespace rmhip
