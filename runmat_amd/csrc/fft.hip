// fft.hip -- discrete Fourier transforms along one dimension of a resident tensor, and the complex-interleaved storage they produce.
//   fft_dim / ifft_dim        crates/runmat-accelerate-api/src/lib.rs:2622-2638   (semantics: the wgpu provider's host form,
//                             runmat-accelerate/src/backend/wgpu/provider/ops/fft/fallback.rs:4-150 - zero-pad / truncate to `len`,
//                             unnormalised forward transform, inverse scaled by 1 / len, complex-interleaved result)
//   fft_extract_real          lib.rs:2639-2644    (ifft(..., 'symmetric'): builtins/math/fft/ifft.rs:362-372)
//   complex_from_real(_imag)  lib.rs:1940-1959
// The reference transforms with rustfft 6.4.1 (Cargo.lock:6116-6118; not under /root/reference): parity is by tolerance against the
// DFT definition (tests/test_gpu_fft.py states it), not by bits.
//
// Kernel: `k_fft_tile` - a workgroup holds a tile of up to 4096 complex points (B lines x m points, m a power of two) in LDS as
// split re / im arrays with one pad slot per 32, runs in-place decimation-in-frequency passes of radix 8 (then 4 or 2) with ONE
// barrier per pass, and leaves through a digit-reversed LDS read, so both the loads and the stores are coalesced: along the points
// when the line is contiguous, across neighbouring lines otherwise (lines along a trailing dimension).  Lengths above the tile take
// two such passes (n = m1 * m2: strided length-m1 transforms, the step twiddle, length-m2 transforms); lengths that are not powers of
// two go through Bluestein's chirp convolution on the same kernel.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

// parity here is by tolerance against the exact transform (the reference's rustfft has its own operation order): fused
// multiply-adds are allowed in this file - fewer instructions and one rounding less per complex product
#pragma clang fp contract(fast)

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
constexpr int FT = 256;         // threads per workgroup of the small elementwise kernels
constexpr int TILE = 4096;      // complex points per workgroup (512 threads, two workgroups per CU) ...
constexpr int TILE_SMALL = 2048;  // ... half of that for lines of up to 2048 points (256 threads, four per CU) ...
constexpr int TILE_BIG = 8192;  // ... or a whole 8192-point line (1024 threads, one workgroup per CU: 132 KiB of LDS)
constexpr int MAXB = 256;       // lines per workgroup
constexpr u64 NOLINE = ~0ull;
constexpr int STEP_LO = 12;     // step twiddle w_n^t = lo[t & 4095] * hi[t >> 12]

struct Side {  // where point p of line (i, q, o) lives: element offset i*si + q*sq + o*so + p*sp
    u64 si, sq, so, sp;
};

struct FftPass {
    u64 nlines, inner, qcnt;  // line l = i + inner * (q + qcnt * o)
    const double* in;
    double* out;
    Side a, b;                // input / output addressing (in complex elements, or real elements for a real input)
    u64 in_pk, in_qk, in_len; // point p of a line with sub-index q exists iff p*in_pk + q*in_qk < in_len (else it reads as zero)
    const double2* tw;        // w_m^k = exp(-2 pi i k / m), k < m
    const double2* step_lo;   // step twiddle tables of the enclosing length (two-pass transforms), or null
    const double2* step_hi;
    const double2* mul_in;    // per-point factor on load, indexed by the point's position p*in_pk + q*in_qk (Bluestein's chirp), or null
    double scale;
    int log2m, log2b;
    int in_complex, mode, step_by_i, conj_in, conj_out, round32;  // mode: see k_fft_tile
    int nrad, rad[5];
};

__device__ __forceinline__ void cmul(double& xr, double& xi, double wr, double wi) {
    const double r = xr * wr - xi * wi, i = xr * wi + xi * wr;
    xr = r;
    xi = i;
}

// natural-order in, natural-order out, forward sign
template <int R>
__device__ __forceinline__ void dft_small(double (&xr)[R], double (&xi)[R]);

template <>
__device__ __forceinline__ void dft_small<2>(double (&xr)[2], double (&xi)[2]) {
    const double ar = xr[0] + xr[1], ai = xi[0] + xi[1], br = xr[0] - xr[1], bi = xi[0] - xi[1];
    xr[0] = ar, xi[0] = ai, xr[1] = br, xi[1] = bi;
}

__device__ __forceinline__ void dft4(double& r0, double& i0, double& r1, double& i1, double& r2, double& i2, double& r3, double& i3) {
    const double t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const double t2r = r1 + r3, t2i = i1 + i3;
    const double t3r = i1 - i3, t3i = -(r1 - r3);  // (a1 - a3) * (-i)
    r0 = t0r + t2r, i0 = t0i + t2i;
    r2 = t0r - t2r, i2 = t0i - t2i;
    r1 = t1r + t3r, i1 = t1i + t3i;
    r3 = t1r - t3r, i3 = t1i - t3i;
}

template <>
__device__ __forceinline__ void dft_small<4>(double (&xr)[4], double (&xi)[4]) {
    dft4(xr[0], xi[0], xr[1], xi[1], xr[2], xi[2], xr[3], xi[3]);
}

template <>
__device__ __forceinline__ void dft_small<8>(double (&xr)[8], double (&xi)[8]) {
    // even and odd halves, then X[k] = E[k] + w8^k O[k], X[k + 4] = E[k] - w8^k O[k]
    dft4(xr[0], xi[0], xr[2], xi[2], xr[4], xi[4], xr[6], xi[6]);
    dft4(xr[1], xi[1], xr[3], xi[3], xr[5], xi[5], xr[7], xi[7]);
    constexpr double h = 0.70710678118654752440;
    double er[4] = {xr[0], xr[2], xr[4], xr[6]}, ei[4] = {xi[0], xi[2], xi[4], xi[6]};
    double orr[4], oi[4];
    orr[0] = xr[1], oi[0] = xi[1];
    orr[1] = h * (xr[3] + xi[3]), oi[1] = h * (xi[3] - xr[3]);   // * (1 - i) / sqrt 2
    orr[2] = xi[5], oi[2] = -xr[5];                              // * (-i)
    orr[3] = h * (xi[7] - xr[7]), oi[3] = -h * (xr[7] + xi[7]);  // * (-1 - i) / sqrt 2
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xr[k] = er[k] + orr[k], xi[k] = ei[k] + oi[k];
        xr[k + 4] = er[k] - orr[k], xi[k + 4] = ei[k] - oi[k];
    }
}

// ---- the tile kernel ---------------------------------------------------------------------------------------------------------------
// Stockham autosort passes over a tile of B lines x m points, eight points per thread (one radix-8 butterfly, two radix-4 or four
// radix-2).  Pass s (Ns = product of the radices before it) takes butterfly j's inputs from points j + q m/R - whatever Ns is, so
// neighbouring threads always read neighbouring points -, multiplies them by w_{Ns R}^(kq), k = j mod Ns, and files result r at point
// (j - k) R + k + r Ns.  The FIRST pass therefore loads straight from global memory and the LAST one (k = j) stores straight to it,
// both coalesced; LDS only carries the exchanges in between (read all - barrier - write all - barrier: one copy of the tile).
// Thread-to-butterfly order and LDS layout follow what is contiguous in memory:
//   mode 0  points of a line are contiguous (a transform along dimension 0): threads along j, LDS index b m + point;
//   mode 1  neighbouring lines are contiguous (a trailing dimension, or the strided first step of a long transform): threads along b,
//           LDS index point B + b;
//   mode 2  contiguous points in, neighbouring lines out (the second step of a long transform - the transposition of the four-step
//           algorithm): the tile is first staged through LDS (4 points x 16 lines per wave), then as mode 1.
// One pad slot per eight keeps the exchange patterns (stride R, stride 8 R, runs of Ns) at two lanes per bank.
__device__ __forceinline__ int padi(int i) { return i + (i >> 3); }

template <int NT>
struct TileCtx {
    double* re;
    double* im;
    const u64* ibase;
    const u64* obase;
    const unsigned* lpos;
    const unsigned* lmul;
    int log2m, log2b, lines_fast;
    __device__ __forceinline__ int lidx(int b, int pt) const { return padi(lines_fast ? (pt << log2b) + b : (b << log2m) + pt); }
};

template <int R>
struct RadixLog {
    static constexpr int v = R == 8 ? 3 : (R == 4 ? 2 : 1);
};

// one pass: K = 8 / R butterflies per thread.  FIRST: inputs come from global memory (unless STAGED: from LDS, like a later pass);
// LAST: results go to global memory.
template <int NT, int R, bool first_from_global, bool last>
__device__ __forceinline__ void stockham_pass(const FftPass& P, const TileCtx<NT>& T, int log2ns) {
    constexpr int LR = RadixLog<R>::v, K = 8 / R;
    const int log2m = T.log2m, log2b = T.log2b;
    const int nb = 1 << (log2m + log2b - LR);  // butterflies in the tile
    const int log2s = log2m - LR;              // input stride m / R
    double xr[K][R], xi[K][R];
    int bb[K], jj[K];
    bool live[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        const int id = threadIdx.x + NT * kk;
        live[kk] = id < nb;
        if (T.lines_fast) bb[kk] = id & ((1 << log2b) - 1), jj[kk] = id >> log2b;
        else jj[kk] = id & ((1 << log2s) - 1), bb[kk] = id >> log2s;
    }
    if (first_from_global) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const u64 base = live[kk] ? T.ibase[bb[kk]] : NOLINE;
            const u64 lp = live[kk] ? (u64)T.lpos[bb[kk]] : 0;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int pt = jj[kk] + (q << log2s);
                const u64 pos = (u64)pt * P.in_pk + lp;
                double vr = 0.0, vi = 0.0;
                if (base != NOLINE && pos < P.in_len) {
                    const u64 off = base + (u64)pt * P.a.sp;
                    if (P.in_complex) {
                        const double2 v = reinterpret_cast<const double2*>(P.in)[off];
                        vr = v.x, vi = v.y;
                    } else {
                        vr = P.in[off];
                    }
                    if (P.conj_in) vi = -vi;
                    if (P.mul_in) {
                        const double2 f = P.mul_in[pos];
                        cmul(vr, vi, f.x, f.y);
                    }
                }
                xr[kk][q] = vr, xi[kk][q] = vi;
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int a = T.lidx(bb[kk], jj[kk] + (q << log2s));
                xr[kk][q] = live[kk] ? T.re[a] : 0.0, xi[kk][q] = live[kk] ? T.im[a] : 0.0;
            }
    }
    if (!first_from_global && log2ns > 0) {  // (the first pass has Ns = 1: no twiddles)
        const int sh = log2m - log2ns - LR;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int k = jj[kk] & ((1 << log2ns) - 1);
            if (R == 8) {
                // w, w^2, w^4 from the table; w^3 = w w^2, w^5 = w w^4, w^6 = w^2 w^4, w^7 = w^3 w^4 (at most two roundings more than a
                // table entry; seven loads per butterfly were as much L1 traffic as the tile's LDS exchange)
                const double2 w1 = P.tw[k << sh], w2 = P.tw[(2 * k) << sh], w4 = P.tw[(4 * k) << sh];
                double w3r = w1.x, w3i = w1.y, w5r = w1.x, w5i = w1.y, w6r = w2.x, w6i = w2.y;
                cmul(w3r, w3i, w2.x, w2.y);
                cmul(w5r, w5i, w4.x, w4.y);
                cmul(w6r, w6i, w4.x, w4.y);
                double w7r = w3r, w7i = w3i;
                cmul(w7r, w7i, w4.x, w4.y);
                cmul(xr[kk][1], xi[kk][1], w1.x, w1.y);
                cmul(xr[kk][2], xi[kk][2], w2.x, w2.y);
                cmul(xr[kk][3], xi[kk][3], w3r, w3i);
                cmul(xr[kk][4], xi[kk][4], w4.x, w4.y);
                cmul(xr[kk][5], xi[kk][5], w5r, w5i);
                cmul(xr[kk][6], xi[kk][6], w6r, w6i);
                cmul(xr[kk][7], xi[kk][7], w7r, w7i);
            } else {
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    const double2 w = P.tw[(k * q) << sh];
                    cmul(xr[kk][q], xi[kk][q], w.x, w.y);
                }
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) dft_small<R>(xr[kk], xi[kk]);
    if (!last) {
        __syncthreads();  // every thread has read its inputs: the tile may be overwritten
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            if (!live[kk]) continue;
            const int k = jj[kk] & ((1 << log2ns) - 1);
            const int o0 = ((jj[kk] - k) << LR) + k;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int a = T.lidx(bb[kk], o0 + (r << log2ns));
                T.re[a] = xr[kk][r], T.im[a] = xi[kk][r];
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        if (!live[kk]) continue;
        const u64 base = T.obase[bb[kk]];
        if (base == NOLINE) continue;
        const unsigned lm = T.lmul[bb[kk]];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pt = jj[kk] + (r << log2ns);  // (in the last pass k = j and Ns = m / R)
            double vr = xr[kk][r], vi = xi[kk][r];
            if (P.step_lo) {
                const u64 tt = (u64)pt * lm;
                const double2 wl = P.step_lo[tt & ((1u << STEP_LO) - 1)], wh = P.step_hi[tt >> STEP_LO];
                double wr = wl.x, wi = wl.y;
                cmul(wr, wi, wh.x, wh.y);
                cmul(vr, vi, wr, wi);
            }
            if (P.conj_out) vi = -vi;
            vr *= P.scale, vi *= P.scale;
            if (P.round32) vr = (double)(float)vr, vi = (double)(float)vi;
            double2 v;
            v.x = vr, v.y = vi;
            reinterpret_cast<double2*>(P.out)[base + (u64)pt * P.b.sp] = v;
        }
    }
}

template <int NT, int R>
__device__ __forceinline__ void run_pass(int variant, const FftPass& P, const TileCtx<NT>& T, int log2ns) {
    switch (variant) {
        case 0: stockham_pass<NT, R, false, false>(P, T, log2ns); break;
        case 1: stockham_pass<NT, R, false, true>(P, T, log2ns); break;
        case 2: stockham_pass<NT, R, true, false>(P, T, log2ns); break;
        default: stockham_pass<NT, R, true, true>(P, T, log2ns); break;
    }
}

template <int NT>
__global__ void __launch_bounds__(NT, 4) k_fft_tile(const FftPass P) {
    extern __shared__ __attribute__((aligned(16))) double fft_lds[];
    const int t = threadIdx.x;
    const int log2m = P.log2m, log2b = P.log2b, B = 1 << log2b, tile = 1 << (log2m + log2b);
    const int plane = tile + (tile >> 3) + 1;
    TileCtx<NT> T;
    T.re = fft_lds;
    T.im = T.re + plane;
    u64* const ibase = reinterpret_cast<u64*>(T.im + plane);
    u64* const obase = ibase + B;
    unsigned* const lpos = reinterpret_cast<unsigned*>(obase + B);
    unsigned* const lmul = lpos + B;
    T.ibase = ibase, T.obase = obase, T.lpos = lpos, T.lmul = lmul;
    T.log2m = log2m, T.log2b = log2b, T.lines_fast = P.mode != 0;
    for (int b = t; b < B; b += NT) {
        const u64 l = (u64)blockIdx.x * B + b;
        if (l < P.nlines) {
            const u64 i = l % P.inner, r = l / P.inner, q = r % P.qcnt, o = r / P.qcnt;
            ibase[b] = i * P.a.si + q * P.a.sq + o * P.a.so;
            obase[b] = i * P.b.si + q * P.b.sq + o * P.b.so;
            lpos[b] = (unsigned)(q * P.in_qk);
            lmul[b] = (unsigned)(P.step_by_i ? i : q);
        } else {
            ibase[b] = NOLINE, obase[b] = NOLINE, lpos[b] = 0, lmul[b] = 0;
        }
    }
    __syncthreads();
    bool from_global = true;
    if (P.mode == 2 || P.nrad == 0) {
        // staged load: a wave takes PW points x LW lines, so that a line's PW points are one contiguous piece of memory and the LW
        // lines are neighbours in the LDS layout (point B + b)
        int lpw = log2m < 2 ? log2m : 2;                       // up to four points of a line ...
        const int llw = log2b < 6 - lpw ? log2b : 6 - lpw;     // ... across up to sixteen lines (more when the lines are shorter)
        if (llw + lpw < 6) lpw = log2m < 6 - llw ? log2m : 6 - llw;
        const int cl = llw + lpw, nbb = B >> llw;              // chunk of 2^cl elements; chunks across the lines
        for (int e = t; e < tile; e += NT) {
            const int chunk = e >> cl, lane = e & ((1 << cl) - 1);
            const int b = ((chunk % nbb) << llw) + (lane >> lpw), pt = ((chunk / nbb) << lpw) + (lane & ((1 << lpw) - 1));
            double vr = 0.0, vi = 0.0;
            const u64 base = ibase[b];
            const u64 pos = (u64)pt * P.in_pk + lpos[b];
            if (base != NOLINE && pos < P.in_len) {
                const u64 off = base + (u64)pt * P.a.sp;
                if (P.in_complex) {
                    const double2 v = reinterpret_cast<const double2*>(P.in)[off];
                    vr = v.x, vi = v.y;
                } else {
                    vr = P.in[off];
                }
                if (P.conj_in) vi = -vi;
                if (P.mul_in) {
                    const double2 f = P.mul_in[pos];
                    cmul(vr, vi, f.x, f.y);
                }
            }
            const int a = T.lidx(b, pt);
            T.re[a] = vr, T.im[a] = vi;
        }
        __syncthreads();
        from_global = false;
    }
    if (P.nrad == 0) {  // one-point lines: the point itself
        for (int e = t; e < tile; e += NT) {
            const int b = e;
            if (obase[b] == NOLINE) continue;
            const int a = T.lidx(b, 0);
            double vr = T.re[a], vi = T.im[a];
            if (P.conj_out) vi = -vi;
            vr *= P.scale, vi *= P.scale;
            if (P.round32) vr = (double)(float)vr, vi = (double)(float)vi;
            double2 v;
            v.x = vr, v.y = vi;
            reinterpret_cast<double2*>(P.out)[obase[b]] = v;
        }
        return;
    }
    // the radices are (4 or 2,) 8, 8, ...: the first pass is dispatched on its radix, every later one is a radix-8 pass - the middle
    // ones in a loop of their own (one variant inside the loop: what the compiler hoists out of it stays small)
    const bool only = P.nrad == 1;
    const int variant = (from_global ? 2 : 0) | (only ? 1 : 0);
    int log2ns;
    if (P.rad[0] == 8) run_pass<NT, 8>(variant, P, T, 0), log2ns = 3;
    else if (P.rad[0] == 4) run_pass<NT, 4>(variant, P, T, 0), log2ns = 2;
    else run_pass<NT, 2>(variant, P, T, 0), log2ns = 1;
    if (only) return;
    for (int s = 1; s + 1 < P.nrad; ++s, log2ns += 3) stockham_pass<NT, 8, false, false>(P, T, log2ns);
    stockham_pass<NT, 8, false, true>(P, T, log2ns);
}

// tables ---------------------------------------------------------------------------------------------------------------------------------
// kind 0: w_n^k, k < count                 (count = n)
// kind 1: w_n^(k * 4096), k < count        (the coarse half of the step twiddle)
// kind 2: exp(-i pi k^2 / n), k < count    (Bluestein's chirp; k^2 mod 2n is taken in integers)
__global__ void __launch_bounds__(FT) k_fft_table(double2* __restrict__ out, u64 count, u64 n, int kind) {
    const u64 k = (u64)blockIdx.x * FT + threadIdx.x;
    if (k >= count) return;
    double s, co;
    if (kind == 2) {
        const u64 k2 = (unsigned long long)(((unsigned __int128)k * k) % (2 * n));
        sincospi((double)k2 / (double)n, &s, &co);
    } else {
        const u64 kk = kind == 1 ? ((k << STEP_LO) % n) : k;
        sincospi(2.0 * (double)kk / (double)n, &s, &co);
    }
    double2 v;
    v.x = co, v.y = -s;
    out[k] = v;
}

typedef std::shared_ptr<Allocation> Table;  // callers hold the reference while their launches are being queued
inline const double2* tptr(const Table& t) { return reinterpret_cast<const double2*>(t->ptr); }

int fft_table(Context* c, int kind, u64 n, u64 count, Table* out) {
    const uint64_t key = ((uint64_t)kind << 60) ^ (n << 28) ^ count;
    auto it = c->fft_tables.find(key);
    if (it == c->fft_tables.end()) {
        std::shared_ptr<Allocation> a;
        RMHIP_TRY(c->alloc_device(2 * count, &a));
        hipLaunchKernelGGL(k_fft_table, dim3((unsigned)((count + FT - 1) / FT)), dim3(FT), 0, c->stream, reinterpret_cast<double2*>(a->ptr), count, n, kind);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        if (c->fft_tables.size() > 64) c->fft_tables.clear();  // bounded: a table is a few microseconds to rebuild
        it = c->fft_tables.emplace(key, a).first;
    }
    *out = it->second;
    return RMHIP_OK;
}

inline int ilog2(u64 v) {
    int l = 0;
    while ((1ull << l) < v) ++l;
    return l;
}
inline bool is_pow2(u64 v) { return v && !(v & (v - 1)); }

void set_radices(FftPass& P) {
    int rem = P.log2m;
    P.nrad = 0;
    if (rem % 3 == 2) P.rad[P.nrad++] = 4, rem -= 2;  // the odd radix first: that pass has no twiddles
    if (rem % 3 == 1) P.rad[P.nrad++] = 2, rem -= 1;
    while (rem >= 3) P.rad[P.nrad++] = 8, rem -= 3;
}

int launch_pass(Context* c, FftPass& P) {
    set_radices(P);
    const int m = 1 << P.log2m;
    int lb = 0;
    // lines of up to 2048 points take half-size tiles (256 threads, 37 KiB: four workgroups per CU instead of two - the same threads per CU
    // in more independent phases: 3-7 % on every shape measured), unless that would leave fewer than eight neighbouring lines to a tile
    const size_t cap = (size_t)m <= (size_t)TILE_SMALL && (P.mode == 0 || (size_t)m * 8 <= (size_t)TILE_SMALL) ? (size_t)TILE_SMALL : (size_t)TILE;
    while (((size_t)m << (lb + 1)) <= cap && (1 << (lb + 1)) <= MAXB && (1ull << lb) < P.nlines) ++lb;
    P.log2b = lb;
    Table tw;
    RMHIP_TRY(fft_table(c, 0, (u64)m, (u64)m, &tw));
    P.tw = tptr(tw);
    const u64 blocks = (P.nlines + (1ull << lb) - 1) >> lb;
    if (blocks > 0x7fffffffull) return fail(RMHIP_ERR_UNSUPPORTED, "fft: %llu lines", P.nlines);
    const size_t tile = (size_t)m << lb, lds = (2 * (tile + (tile >> 3) + 1) + 2 * ((size_t)1 << lb)) * sizeof(double) + 2 * ((size_t)1 << lb) * sizeof(unsigned);
    if (tile > (size_t)TILE) hipLaunchKernelGGL(k_fft_tile<1024>, dim3((unsigned)blocks), dim3(1024), lds, c->stream, P);
    else if (tile <= (size_t)TILE_SMALL) hipLaunchKernelGGL(k_fft_tile<256>, dim3((unsigned)blocks), dim3(256), lds, c->stream, P);
    else hipLaunchKernelGGL(k_fft_tile<512>, dim3((unsigned)blocks), dim3(512), lds, c->stream, P);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// A set of lines in tensor layout: point k of line (i, o) at element i + inner * (k + len * o).
struct Lines {
    const double* in;
    int in_complex;
    u64 inner, outer, len_in;  // len_in: points the input holds per line
    double* out;               // complex, n points per line, same layout
};

// forward transform of conj^a(x), conjugated b times (a = b = 1: the unscaled inverse) - power-of-two length n along the lines; `scale` multiplies the result.
// `mul_in` (or null): a factor per input point k (Bluestein's chirp); points from `valid` on read as zero whatever the input holds.
int fft_pow2(Context* c, const Lines& L, u64 n, bool conj_in, bool conj_out, double scale, bool round32, const double2* mul_in = nullptr, u64 valid = ~0ull) {
    const int lg = ilog2(n);
    const u64 in_len = std::min<u64>(std::min<u64>(L.len_in, n), valid);
    const u64 lines = L.inner * L.outer;
    // a line along a trailing dimension is read across neighbouring lines: short transforms, many lines per tile
    const bool single = L.inner == 1 ? n <= (u64)TILE_BIG : lg <= 8;
    if (single) {
        FftPass P{};
        P.nlines = lines, P.inner = L.inner, P.qcnt = 1;
        P.in = L.in, P.out = L.out, P.in_complex = L.in_complex;
        P.a = Side{1, 0, L.len_in * L.inner, L.inner};
        P.b = Side{1, 0, n * L.inner, L.inner};
        P.in_pk = 1, P.in_qk = 0, P.in_len = in_len;
        P.mul_in = mul_in;
        P.scale = scale, P.log2m = lg;
        P.mode = L.inner > 1 ? 1 : 0;
        P.conj_in = conj_in, P.conj_out = conj_out;
        P.round32 = round32;
        return launch_pass(c, P);
    }
    if (lg > (L.inner == 1 ? 27 : 24)) return fail(RMHIP_ERR_UNSUPPORTED, "fft: length %llu", n);
    if (L.inner == 1 && lg >= 23) {  // measured crossover (scripts/fft_long_ab.py): 2^22 equal, 2^23 0.33 -> 0.22 ms, 2^24 0.78 -> 0.49 ms
        // Two factors of a very long line pass 2048 and a tile holds one or two lines: the strided side of each pass moves 16- or 32-byte
        // pieces (a 2^24-point vector: 0.80 ms).  Three passes instead, n = m1 m2 m3, k = k1 m2 m3 + k2 m3 + k3,
        // j = j1 + m1 j2 + m1 m2 j3, every one reading and writing neighbouring lines:
        //   (A) over k1 (stride m2 m3), lines (k', o), k' = k2 m3 + k3, times w_n^(j1 k')          -> T[o][j1][k']
        //   (B) over k2 (stride m3),   lines (k3, j1, o),            times w_{m2 m3}^(j2 k3)       -> T[o][j1][j2][k3]   (in place)
        //   (C) over k3 (contiguous),  lines (j1, j2, o)                                           -> y[o][j1 + m1 j2 + m1 m2 j3]
        const int l1 = lg / 3, l2 = (lg - l1) / 2, l3 = lg - l1 - l2;
        const u64 m1 = 1ull << l1, m2 = 1ull << l2, m3 = 1ull << l3, N1 = m2 * m3;
        std::shared_ptr<Allocation> t;
        RMHIP_TRY(c->alloc_device(2 * n * lines, &t));
        Table slo, shi, rlo, rhi;
        RMHIP_TRY(fft_table(c, 0, n, std::min<u64>(n, 1ull << STEP_LO), &slo));
        RMHIP_TRY(fft_table(c, 1, n, std::max<u64>(1, n >> STEP_LO), &shi));
        RMHIP_TRY(fft_table(c, 0, N1, std::min<u64>(N1, 1ull << STEP_LO), &rlo));
        RMHIP_TRY(fft_table(c, 1, N1, std::max<u64>(1, N1 >> STEP_LO), &rhi));
        {
            FftPass P{};
            P.nlines = lines * N1, P.inner = 1, P.qcnt = N1;
            P.in = L.in, P.out = t->ptr, P.in_complex = L.in_complex;
            P.a = Side{0, 1, L.len_in, N1};
            P.b = Side{0, 1, n, N1};
            P.in_pk = N1, P.in_qk = 1, P.in_len = in_len;
            P.mul_in = mul_in;
            P.step_lo = tptr(slo), P.step_hi = tptr(shi);
            P.scale = 1.0, P.log2m = l1, P.mode = 1;
            P.conj_in = conj_in;
            RMHIP_TRY(launch_pass(c, P));
        }
        {
            FftPass P{};
            P.nlines = lines * m1 * m3, P.inner = m3, P.qcnt = m1;
            P.in = t->ptr, P.out = t->ptr, P.in_complex = 1;
            P.a = Side{1, N1, n, m3};
            P.b = Side{1, N1, n, m3};
            P.in_pk = 1, P.in_qk = 0, P.in_len = m2;
            P.step_lo = tptr(rlo), P.step_hi = tptr(rhi), P.step_by_i = 1;
            P.scale = 1.0, P.log2m = l2, P.mode = 1;
            RMHIP_TRY(launch_pass(c, P));
        }
        {
            FftPass P{};
            P.nlines = lines * m1 * m2, P.inner = m1, P.qcnt = m2;
            P.in = t->ptr, P.out = L.out, P.in_complex = 1;
            P.a = Side{N1, m3, n, 1};
            P.b = Side{1, m1, n, m1 * m2};
            P.in_pk = 1, P.in_qk = 0, P.in_len = m3;
            P.scale = scale, P.log2m = l3, P.mode = 2;
            P.conj_out = conj_out;
            P.round32 = round32;
            RMHIP_TRY(launch_pass(c, P));
        }
        return RMHIP_OK;
    }
    // n = m1 * m2: (1) length-m1 transforms over k1 of x[k1 * m2 + k2], times w_n^(j1 k2), into T[j1 * m2 + k2];
    //              (2) length-m2 transforms over k2 of T[j1 * m2 + k2] into y[j1 + m1 * j2]
    const int l1 = lg / 2, l2 = lg - l1;
    const u64 m1 = 1ull << l1, m2 = 1ull << l2;
    std::shared_ptr<Allocation> tmp;
    RMHIP_TRY(c->alloc_device(2 * n * lines, &tmp));
    Table slo, shi;
    RMHIP_TRY(fft_table(c, 0, n, std::min<u64>(n, 1ull << STEP_LO), &slo));
    RMHIP_TRY(fft_table(c, 1, n, std::max<u64>(1, n >> STEP_LO), &shi));
    {
        FftPass P{};
        P.nlines = lines * m2, P.inner = L.inner, P.qcnt = m2;
        P.in = L.in, P.out = tmp->ptr, P.in_complex = L.in_complex;
        P.a = Side{1, L.inner, L.len_in * L.inner, m2 * L.inner};
        P.b = Side{1, L.inner, n * L.inner, m2 * L.inner};
        P.in_pk = m2, P.in_qk = 1, P.in_len = in_len;
        P.mul_in = mul_in;
        P.step_lo = tptr(slo), P.step_hi = tptr(shi);
        P.scale = 1.0, P.log2m = l1;
        P.mode = 1;  // neighbouring lines (i, then k2) are neighbours in memory
        P.conj_in = conj_in;
        RMHIP_TRY(launch_pass(c, P));
    }
    {
        FftPass P{};
        P.nlines = lines * m1, P.inner = L.inner, P.qcnt = m1;
        P.in = tmp->ptr, P.out = L.out, P.in_complex = 1;
        P.a = Side{1, L.inner * m2, n * L.inner, L.inner};
        P.b = Side{1, L.inner, n * L.inner, m1 * L.inner};
        P.in_pk = 1, P.in_qk = 0, P.in_len = m2;
        P.scale = scale, P.log2m = l2;
        P.mode = L.inner > 1 ? 1 : 2;
        P.conj_out = conj_out;
        P.round32 = round32;
        RMHIP_TRY(launch_pass(c, P));
    }
    return RMHIP_OK;
}

// Bluestein: X[j] = conj-chirp[j] * sum_k (x[k] chirp[k]) * conj(chirp)[j - k], chirp[k] = exp(-i pi k^2 / n); the convolution by
// power-of-two transforms of length M >= 2n - 1.  G = FFT_M(g), g[k] = conj(chirp[|k|]) wrapped, is computed once per (n, M).
__global__ void __launch_bounds__(FT) k_bluestein_kernel(const double2* __restrict__ chirp, u64 n, u64 M, double2* __restrict__ g) {
    const u64 k = (u64)blockIdx.x * FT + threadIdx.x;
    if (k >= M) return;
    double2 v;
    v.x = 0.0, v.y = 0.0;
    if (k < n) v.x = chirp[k].x, v.y = -chirp[k].y;
    else if (M - k < n) v.x = chirp[M - k].x, v.y = -chirp[M - k].y;
    g[k] = v;
}

// y(i, k, o) *= G[k] (in place; tensor layout with M points per line)
__global__ void __launch_bounds__(FT) k_line_mul(double2* __restrict__ a, const double2* __restrict__ g, u64 inner, u64 M, u64 total) {
    const u64 e = (u64)blockIdx.x * FT + threadIdx.x;
    if (e >= total) return;
    double2 v = a[e];
    const double2 w = g[(e / inner) % M];
    cmul(v.x, v.y, w.x, w.y);
    a[e] = v;
}

// out(i, j, o) = scale * op(chirp[j] * y(i, j, o)) for j < n: the first n points of the M-long lines
__global__ void __launch_bounds__(FT) k_bluestein_finish(const double2* __restrict__ y, const double2* __restrict__ chirp, u64 inner, u64 n, u64 M, u64 total,
                                                         double scale, int conj_out, int round32, double2* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * FT + threadIdx.x;  // output element i + inner * (j + n * o)
    if (e >= total) return;
    const u64 i = e % inner, r = e / inner, j = r % n, o = r / n;
    double2 v = y[i + inner * (j + M * o)];
    const double2 w = chirp[j];
    cmul(v.x, v.y, w.x, w.y);
    if (conj_out) v.y = -v.y;
    v.x *= scale, v.y *= scale;
    if (round32) v.x = (double)(float)v.x, v.y = (double)(float)v.y;
    out[e] = v;
}

int fft_bluestein(Context* c, const Lines& L, u64 n, bool inverse, double scale, bool round32) {
    u64 M = 1;
    while (M < 2 * n - 1) M <<= 1;
    if (M > (1ull << 24)) return fail(RMHIP_ERR_UNSUPPORTED, "fft: length %llu", n);
    Table chirp;
    RMHIP_TRY(fft_table(c, 2, n, n, &chirp));
    // G = FFT_M of the wrapped conjugate chirp, kept beside the tables (kind 3)
    const uint64_t gkey = (3ull << 60) ^ (n << 28) ^ M;
    Table G;
    auto it = c->fft_tables.find(gkey);
    if (it != c->fft_tables.end()) {
        G = it->second;
    } else {
        Table g;
        RMHIP_TRY(c->alloc_device(2 * M, &g));
        RMHIP_TRY(c->alloc_device(2 * M, &G));
        hipLaunchKernelGGL(k_bluestein_kernel, dim3((unsigned)((M + FT - 1) / FT)), dim3(FT), 0, c->stream, tptr(chirp), n, M, reinterpret_cast<double2*>(g->ptr));
        c->tel.kernel_launches++;
        Lines gl{g->ptr, 1, 1, 1, M, G->ptr};
        RMHIP_TRY(fft_pow2(c, gl, M, false, false, 1.0, false));
        c->fft_tables[gkey] = G;
    }
    // the two work tensors [inner, M, outer'] stay below ~1 GiB each: whole outer slabs at a time
    const u64 slab = L.inner * M;
    if (slab > (1ull << 27)) return fail(RMHIP_ERR_UNSUPPORTED, "fft: %llu lines of chirp length %llu along a trailing dimension", L.inner, M);
    const u64 outer_per = std::min<u64>(L.outer, std::max<u64>(1, (1ull << 26) / slab));
    Table A, Y;
    RMHIP_TRY(c->alloc_device(2 * slab * outer_per, &A));
    RMHIP_TRY(c->alloc_device(2 * slab * outer_per, &Y));
    const u64 valid = std::min<u64>(L.len_in, n);
    for (u64 o0 = 0; o0 < L.outer; o0 += outer_per) {
        const u64 oc = std::min<u64>(outer_per, L.outer - o0), total = oc * slab;
        const double* in = L.in + (L.in_complex ? 2 : 1) * (o0 * L.len_in * L.inner);
        double* out = L.out + 2 * (o0 * n * L.inner);
        Lines f{in, L.in_complex, L.inner, oc, L.len_in, A->ptr};  // a = op(x) .* chirp, zero-extended to M, transformed
        RMHIP_TRY(fft_pow2(c, f, M, inverse, false, 1.0, false, tptr(chirp), valid));
        hipLaunchKernelGGL(k_line_mul, dim3((unsigned)((total + FT - 1) / FT)), dim3(FT), 0, c->stream, reinterpret_cast<double2*>(A->ptr), tptr(G), L.inner, M, total);
        Lines b{A->ptr, 1, L.inner, oc, M, Y->ptr};
        RMHIP_TRY(fft_pow2(c, b, M, true, true, 1.0 / (double)M, false));
        const u64 ototal = oc * L.inner * n;
        hipLaunchKernelGGL(k_bluestein_finish, dim3((unsigned)((ototal + FT - 1) / FT)), dim3(FT), 0, c->stream, reinterpret_cast<const double2*>(Y->ptr), tptr(chirp),
                           L.inner, n, M, ototal, scale, inverse ? 1 : 0, round32 ? 1 : 0, reinterpret_cast<double2*>(out));
        c->tel.kernel_launches += 2;
        RMHIP_HIP_CHECK(hipGetLastError());
    }
    return RMHIP_OK;
}

int fft_entry(Context* c, rmhip_buf a, long long len_or_neg, int dim, bool inverse, rmhip_buf* out) {
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "fft: dim must be >= 0");
    Buffer ab;
    RMHIP_TRY(c->get_any(a, &ab));
    std::vector<size_t> shape = ab.shape;
    if (shape.empty()) shape.push_back(ab.numel);  // fallback.rs:20-29
    const size_t origin_rank = shape.size();
    while (shape.size() <= (size_t)dim) shape.push_back(1);
    const u64 cur = shape[dim], n = len_or_neg < 0 ? cur : (u64)len_or_neg;
    u64 inner = 1, outer = 1;
    for (int k = 0; k < dim; ++k) inner *= shape[k];
    for (size_t k = dim + 1; k < shape.size(); ++k) outer *= shape[k];
    std::vector<size_t> oshape = shape;
    oshape[dim] = n;
    // fft_trim_trailing_ones(out_shape, max(origin_rank, dim + 1)) (mod.rs:14-19): nothing beyond that rank was added, so only the
    // scalar normalisation is left
    (void)origin_rank;
    bool scalar = true;
    for (size_t e : oshape) scalar = scalar && e == 1;
    if (scalar) oshape = {1, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer_complex(oshape.data(), oshape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    const bool r32 = c->precision == 32;
    Lines L{ab.data(), ab.cplx ? 1 : 0, inner, outer, cur, ob.data()};
    const double scale = inverse ? 1.0 / (double)n : 1.0;
    int rc;
    if (n == 1 || cur == 0) {
        // a one-point transform is the point itself (or zero where the input has none): the tile kernel with m = 1 has no pass to run
        FftPass P{};
        P.nlines = inner * outer, P.inner = inner, P.qcnt = 1;
        P.in = L.in, P.out = L.out, P.in_complex = L.in_complex;
        P.a = Side{1, 0, cur * inner, inner};
        P.b = Side{1, 0, n * inner, inner};
        P.in_pk = 1, P.in_qk = 0, P.in_len = std::min<u64>(cur, n);
        P.scale = scale, P.log2m = 0, P.round32 = r32;
        P.mode = 1;
        if (n == 1) rc = launch_pass(c, P);
        else {
            RMHIP_HIP_CHECK(hipMemsetAsync(ob.data(), 0, 2 * ob.numel * sizeof(double), c->stream));
            rc = RMHIP_OK;
        }
    } else if (is_pow2(n)) {
        rc = fft_pow2(c, L, n, inverse, inverse, scale, r32);
    } else {
        rc = fft_bluestein(c, L, n, inverse, scale, r32);
    }
    if (rc != RMHIP_OK) {
        Buffer victim;
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->table.find(*out);
        if (it != c->table.end()) victim = std::move(it->second), c->table.erase(it);
        *out = 0;
    }
    return rc;
}

// the analytic-signal mask of hilbert (builtins/math/signal/hilbert.rs:349-412): the spectrum times 1 at frequency 0 (and at n / 2 for even n),
// 2 on the positive half, 0 on the negative half - in place, tensor layout [inner, n, outer]
__global__ void __launch_bounds__(FT) k_hilbert_mask(double2* __restrict__ a, u64 inner, u64 n, u64 total) {
    const u64 e = (u64)blockIdx.x * FT + threadIdx.x;
    if (e >= total) return;
    const u64 f = (e / inner) % n;
    double sc;
    if (f == 0) sc = 1.0;
    else if (n % 2 == 0) sc = f < n / 2 ? 2.0 : (f == n / 2 ? 1.0 : 0.0);
    else sc = f <= n / 2 ? 2.0 : 0.0;
    double2 v = a[e];
    v.x *= sc, v.y *= sc;
    a[e] = v;
}

__global__ void __launch_bounds__(FT) k_complex_make(const double* __restrict__ re, u64 re_n, const double* __restrict__ im, u64 im_n, u64 total, int round32,
                                                     double2* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * FT + threadIdx.x;
    if (e >= total) return;
    double2 v;
    v.x = re[re_n == 1 ? 0 : e];
    v.y = im ? im[im_n == 1 ? 0 : e] : 0.0;
    if (round32) v.x = (double)(float)v.x, v.y = (double)(float)v.y;
    out[e] = v;
}

__global__ void __launch_bounds__(FT) k_complex_real(const double2* __restrict__ a, u64 total, double* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * FT + threadIdx.x;
    if (e < total) out[e] = a[e].x;
}

}  // namespace
}  // namespace rmhip

extern "C" {

int rmhip_fft_dim(rmhip_ctx* ctx, rmhip_buf a, long long len_or_neg, int dim, int inverse, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    return fft_entry(c, a, len_or_neg, dim, inverse != 0, out);
}

int rmhip_hilbert(rmhip_ctx* ctx, rmhip_buf a, long long len_or_neg, int dim, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (len_or_neg == 0) return fail(RMHIP_ERR_INVALID, "signal_hilbert: invalid request");  // lib.rs:492-494
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));  // (real input: a complex one is refused here)
    if (dim < 0 || (size_t)dim >= ab.shape.size()) return fail(RMHIP_ERR_INVALID, "signal_hilbert: invalid request");  // lib.rs:495-497
    // hilbert.rs: the analytic signal ifft(fft(x, n, dim) .* mask, n, dim)
    rmhip_buf spec = 0;
    RMHIP_TRY(fft_entry(c, a, len_or_neg, dim, false, &spec));
    Buffer sb;
    int rc = c->get_any(spec, &sb);
    if (rc == RMHIP_OK && sb.numel) {
        u64 inner = 1;
        for (int k = 0; k < dim; ++k) inner *= sb.shape[k];
        const u64 n = (size_t)dim < sb.shape.size() ? sb.shape[dim] : 1;
        hipLaunchKernelGGL(k_hilbert_mask, dim3((unsigned)((sb.numel + FT - 1) / FT)), dim3(FT), 0, c->stream, reinterpret_cast<double2*>(sb.data()), inner, n, (u64)sb.numel);
        c->tel.kernel_launches++;
        if (hipGetLastError() != hipSuccess) rc = fail(RMHIP_ERR_HIP, "signal_hilbert: launch failed");
    }
    if (rc == RMHIP_OK) rc = fft_entry(c, spec, -1, dim, true, out);
    rmhip_free(ctx, spec);
    return rc;
}

int rmhip_complex(rmhip_ctx* ctx, rmhip_buf real, rmhip_buf imag_or_0, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer rb, ib;
    RMHIP_TRY(c->get(real, &rb));
    if (imag_or_0) RMHIP_TRY(c->get(imag_or_0, &ib));
    // complex(real, imag): equal shapes, or either operand a scalar that expands (lib.rs:1949-1959)
    const Buffer* shape_of = &rb;
    if (imag_or_0) {
        if (rb.numel == 1 && ib.numel != 1) shape_of = &ib;
        else if (ib.numel != 1 && ib.shape != rb.shape) return fail(RMHIP_ERR_SHAPE, "complex: real and imaginary parts must have the same size");
    }
    Buffer ob;
    RMHIP_TRY(c->new_buffer_complex(shape_of->shape.data(), shape_of->shape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    hipLaunchKernelGGL(k_complex_make, dim3((unsigned)((ob.numel + FT - 1) / FT)), dim3(FT), 0, c->stream, rb.data(), (u64)rb.numel, imag_or_0 ? ib.data() : nullptr,
                       (u64)(imag_or_0 ? ib.numel : 0), (u64)ob.numel, c->precision == 32 ? 1 : 0, reinterpret_cast<double2*>(ob.data()));
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

int rmhip_zeros_complex(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (rank && !shape)) return fail(RMHIP_ERR_INVALID, "zeros: null argument");
    Buffer ob;
    RMHIP_TRY(c->new_buffer_complex(shape, rank, out, &ob));
    if (ob.numel) RMHIP_HIP_CHECK(hipMemsetAsync(ob.data(), 0, 2 * ob.numel * sizeof(double), c->stream));
    return RMHIP_OK;
}

int rmhip_complex_real(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab;
    RMHIP_TRY(c->get_any(a, &ab));
    Buffer ob;
    RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if (!ab.cplx) {  // already real: the caller frees the transform's handle and keeps this one (ifft.rs:368-371), so it is a copy
        RMHIP_HIP_CHECK(hipMemcpyAsync(ob.data(), ab.data(), ob.numel * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return RMHIP_OK;
    }
    hipLaunchKernelGGL(k_complex_real, dim3((unsigned)((ob.numel + FT - 1) / FT)), dim3(FT), 0, c->stream, reinterpret_cast<const double2*>(ab.data()), (u64)ob.numel,
                       ob.data());
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

}  // extern "C"
