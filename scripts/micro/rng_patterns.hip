// Developer micro-benchmark (GPU box): write patterns of the randn generator.  The Box-Muller step is the library's (skel_rng.h); what
// varies is which pairs a thread generates and where it stores them.  1e8 samples.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Irunmat_amd/csrc scripts/micro/rng_patterns.hip -o scripts/micro/rng_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "rng_tables.h"
#include "skel_rng.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: grid-stride (the library's k_rng_normal); STORE = false keeps the arithmetic and drops the stores
template <bool STORE>
__global__ void __launch_bounds__(256) k_stride(unsigned long long state, double* __restrict__ out, size_t n, unsigned long long jm, unsigned long long jp) {
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t full = n / 2, g = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (g >= full) return;
    unsigned long long x1 = lcg_skip2(state, 512ULL * blockIdx.x, 2ULL * threadIdx.x + 1ULL);
    double acc = 0.0;
    for (size_t i = g; i < full; i += stride) {
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(lcg_step(x1), tb, &sn, &cs);
        if (STORE) *(v2d*)(out + 2 * i) = v2d{radius * cs, radius * sn};
        else acc += radius * cs + radius * sn;
        x1 = jm * x1 + jp;
    }
    if (!STORE && acc == 1.2345) out[g] = acc;
}

// B: every block owns one contiguous chunk of PPT * 256 pairs and walks it front to back (PPT iterations of 4 KiB); a thread's first
// state = table[thread] applied to the block's start state (one 64-bit multiply instead of nine doublings)
template <int PPT, bool TABLE>
__global__ void __launch_bounds__(256) k_chunk(unsigned long long state, double* __restrict__ out, size_t n, unsigned long long jm, unsigned long long jp,
                                              const unsigned long long* __restrict__ skip) {
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t full = n / 2, base = (size_t)blockIdx.x * (256 * PPT);
    unsigned long long x1;
    if (TABLE) {
        unsigned long long mb, pb;
        lcg_jump(2ULL * base, &mb, &pb);
        x1 = skip[2 * threadIdx.x] * (mb * state + pb) + skip[2 * threadIdx.x + 1];
    } else {
        x1 = lcg_skip2(state, 2ULL * base, 2ULL * threadIdx.x + 1ULL);
    }
#pragma unroll 1
    for (int it = 0; it < PPT; ++it) {
        const size_t i = base + (size_t)it * 256 + threadIdx.x;
        if (i >= full) break;
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(lcg_step(x1), tb, &sn, &cs);
        *(v2d*)(out + 2 * i) = v2d{radius * cs, radius * sn};
        x1 = jm * x1 + jp;
    }
}

// D: as B with the tables read from GLOBAL memory (L1 / L2 hits: 10 KiB) - no staging, no barrier per block, so small chunks become
// affordable (the pattern that serves k_fill best is 8 KiB per block)
template <int PPT>
__global__ void __launch_bounds__(256) k_chunk_gtab(unsigned long long state, double* __restrict__ out, size_t n, unsigned long long jm, unsigned long long jp,
                                                   const unsigned long long* __restrict__ skip) {
    BmTables tb;
    tb.sc = reinterpret_cast<const v2d*>(&kSinCosPi[0][0]);
    tb.lg = reinterpret_cast<const v2d*>(&kLogTab[0][0]);
    const size_t full = n / 2, base = (size_t)blockIdx.x * (256 * PPT);
    unsigned long long mb, pb;
    lcg_jump(2ULL * base, &mb, &pb);
    unsigned long long x1 = skip[2 * threadIdx.x] * (mb * state + pb) + skip[2 * threadIdx.x + 1];
#pragma unroll 1
    for (int it = 0; it < PPT; ++it) {
        const size_t i = base + (size_t)it * 256 + threadIdx.x;
        if (i >= full) break;
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(lcg_step(x1), tb, &sn, &cs);
        *(v2d*)(out + 2 * i) = v2d{radius * cs, radius * sn};
        x1 = jm * x1 + jp;
    }
}

// C: as B but a thread generates PPT consecutive pairs?  (no: stores would not coalesce) - instead each WAVE owns a contiguous piece:
// wave w of the block walks pairs [base + w * 64 * PPT, ...) in 1 KiB steps
template <int PPT>
__global__ void __launch_bounds__(256) k_wavechunk(unsigned long long state, double* __restrict__ out, size_t n, unsigned long long jm, unsigned long long jp,
                                                  const unsigned long long* __restrict__ skip) {
    __shared__ __attribute__((aligned(16))) double s_tab[kBmLdsDoubles];
    const BmTables tb = bm_stage_tables(s_tab, threadIdx.x, 256);
    const size_t full = n / 2;
    const size_t wbase = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * PPT);
    unsigned long long mb, pb;
    lcg_jump(2ULL * wbase, &mb, &pb);
    const unsigned lane = threadIdx.x & 63;
    unsigned long long x1 = skip[2 * lane] * (mb * state + pb) + skip[2 * lane + 1];
#pragma unroll 1
    for (int it = 0; it < PPT; ++it) {
        const size_t i = wbase + (size_t)it * 64 + lane;
        if (i >= full) break;
        const double radius = bm_radius(x1, tb);
        double sn, cs;
        bm_sincos(lcg_step(x1), tb, &sn, &cs);
        *(v2d*)(out + 2 * i) = v2d{radius * cs, radius * sn};
        x1 = jm * x1 + jp;
    }
}

int main() {
    const size_t n = 100000000;
    double* out; CK(hipMalloc(&out, n * 8));
    unsigned long long* skip; CK(hipMalloc(&skip, 256 * 16));
    unsigned long long h[512];
    for (int t = 0; t < 256; ++t) lcg_jump(2ULL * t + 1ULL, &h[2 * t], &h[2 * t + 1]);
    CK(hipMemcpy(skip, h, sizeof h, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned long long st = 0x9e3779b97f4a7c15ULL;
    auto timeit = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        float best = 1e30f, sum = 0;
        for (int r = 0; r < 10; ++r) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        double chk[4]; (void)hipMemcpy(chk, out + 12345678, 32, hipMemcpyDeviceToHost);
        printf("%-46s best %7.1f us  mean %7.1f us  %5.2f TB/s   z[12345678..] = %.17g %.17g\n", name, best * 1e3, sum * 1e2, 8e8 / (best * 1e-3) / 1e12, chk[0], chk[1]);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int grid : {2048, 4096, 8192}) {
            unsigned long long jm, jp; lcg_jump(2ULL * grid * 256, &jm, &jp);
            char nm[64]; snprintf(nm, sizeof nm, "A grid-stride, grid %d", grid);
            timeit(nm, [&] { hipLaunchKernelGGL(k_stride<true>, dim3(grid), dim3(256), 0, 0, st, out, n, jm, jp); });
        }
        { unsigned long long jm, jp; lcg_jump(2ULL * 2048 * 256, &jm, &jp);
          timeit("A grid-stride, grid 2048, NO stores", [&] { hipLaunchKernelGGL(k_stride<false>, dim3(2048), dim3(256), 0, 0, st, out, n, jm, jp); }); }
        unsigned long long jm, jp; lcg_jump(512ULL, &jm, &jp);
        #define CHUNK(PPT, TAB) { const unsigned g = (unsigned)((n / 2 + 256 * PPT - 1) / (256 * PPT)); \
            timeit(TAB ? "B block chunk PPT " #PPT ", skip table" : "B block chunk PPT " #PPT ", doubling skip", [&] { hipLaunchKernelGGL((k_chunk<PPT, TAB>), dim3(g), dim3(256), 0, 0, st, out, n, jm, jp, skip); }); }
        CHUNK(4, true) CHUNK(8, true) CHUNK(8, false) CHUNK(16, true) CHUNK(32, true) CHUNK(64, true)
        #define GCHUNK(PPT) { const unsigned g = (unsigned)((n / 2 + 256 * PPT - 1) / (256 * PPT)); \
            timeit("D block chunk PPT " #PPT ", tables from global memory", [&] { hipLaunchKernelGGL((k_chunk_gtab<PPT>), dim3(g), dim3(256), 0, 0, st, out, n, jm, jp, skip); }); }
        GCHUNK(1) GCHUNK(2) GCHUNK(4) GCHUNK(8) GCHUNK(16)
        unsigned long long wm, wp; lcg_jump(128ULL, &wm, &wp);
        #define WCHUNK(PPT) { const unsigned g = (unsigned)((n / 2 + 256 * PPT - 1) / (256 * PPT)); \
            timeit("C wave chunk PPT " #PPT, [&] { hipLaunchKernelGGL((k_wavechunk<PPT>), dim3(g), dim3(256), 0, 0, st, out, n, wm, wp, skip); }); }
        WCHUNK(8) WCHUNK(16) WCHUNK(32)
    }
    return 0;
}
