// Developer micro-benchmark (GPU box): 8192 x 8192 f64 transpose (dst[j + i*N] = src[i + j*N]) - which tile shape / access width gets
// closest to the streaming rate?  TB/s on the 1 GiB moved.
//   t64    64 x 64 tile, 8-byte accesses, 512-byte segments both sides (k_transpose / k_permute_tiled today)
//   t64v   64 x 64 tile, 16-byte loads and stores
//   t128x64  128 (source-contiguous) x 64 tile: 1 KiB load segments, 512-byte store segments, 8-byte accesses
//   t128v  128 x 128 tile, 16-byte accesses, 1 KiB segments both sides (132 KiB of LDS: one block per CU)
//   t32    32 x 32 tile (256-byte segments), many small blocks
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/transpose_patterns.hip -o scripts/micro/transpose_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2 __attribute__((ext_vector_type(2)));

template <int TS>
__global__ void __launch_bounds__(256) k_t(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    __shared__ double tile[TS][TS + 1];
    const size_t r0 = (size_t)blockIdx.x * TS, c0 = (size_t)blockIdx.y * TS;
    const int lane = threadIdx.x % TS, grp = threadIdx.x / TS;
    constexpr int G = 256 / TS;
#pragma unroll 4
    for (int p = 0; p < TS / G; ++p) tile[grp + G * p][lane] = src[(r0 + lane) + (c0 + grp + G * p) * n];
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < TS / G; ++p) dst[(c0 + lane) + (r0 + grp + G * p) * n] = tile[lane][grp + G * p];
}

// t64 with non-temporal loads and stores
__global__ void __launch_bounds__(256) k_t64nt(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    __shared__ double tile[64][65];
    const size_t r0 = (size_t)blockIdx.x * 64, c0 = (size_t)blockIdx.y * 64;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    double v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = __builtin_nontemporal_load(src + (r0 + lane) + (c0 + grp + 4 * p) * n);
#pragma unroll
    for (int p = 0; p < 16; ++p) tile[grp + 4 * p][lane] = v[p];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) __builtin_nontemporal_store(tile[lane][grp + 4 * p], dst + (c0 + lane) + (r0 + grp + 4 * p) * n);
}
// plain copy kernels for the ceiling: grid-stride 1024 threads NT v2 (k_stream1's shape) and chunk-per-block
__global__ void __launch_bounds__(1024) k_copy_gs(const double* __restrict__ src, double* __restrict__ dst, size_t n2) {
    const size_t nv = n2 >> 1, st = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < nv; i += st) __builtin_nontemporal_store(__builtin_nontemporal_load((const v2*)src + i), (v2*)dst + i);
}
__global__ void __launch_bounds__(256) k_copy_chunk(const double* __restrict__ src, double* __restrict__ dst, size_t n2) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = src[b + e * 256];
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[b + e * 256] = v[e];
}

// 64 x 64 tile, 16-byte accesses: 32 lanes cover a 64-element row
__global__ void __launch_bounds__(256) k_t64v(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    __shared__ double tile[64][65];
    const size_t r0 = (size_t)blockIdx.x * 64, c0 = (size_t)blockIdx.y * 64;
    const int l2 = threadIdx.x & 31, grp = threadIdx.x >> 5;  // 8 groups
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int cc = grp + 8 * p;
        const v2 v = *(const v2*)(src + (r0 + 2 * l2) + (c0 + cc) * n);
        tile[cc][2 * l2] = v.x;
        tile[cc][2 * l2 + 1] = v.y;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int rr = grp + 8 * p;
        const v2 v = {tile[2 * l2][rr], tile[2 * l2 + 1][rr]};
        *(v2*)(dst + (c0 + 2 * l2) + (r0 + rr) * n) = v;
    }
}

// 128 (rows, source contiguous) x 64 (cols) tile, 8-byte accesses, 512 threads
__global__ void __launch_bounds__(512) k_t128x64(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    __shared__ double tile[64][129];  // [col][row]
    const size_t r0 = (size_t)blockIdx.x * 128, c0 = (size_t)blockIdx.y * 64;
    const int lane = threadIdx.x & 127, grp = threadIdx.x >> 7;  // 4 groups
#pragma unroll 4
    for (int p = 0; p < 16; ++p) tile[grp + 4 * p][lane] = src[(r0 + lane) + (c0 + grp + 4 * p) * n];
    __syncthreads();
    const int l64 = threadIdx.x & 63, g8 = threadIdx.x >> 6;  // 8 groups of 64 lanes: a destination column = 64 contiguous elements
#pragma unroll 4
    for (int p = 0; p < 16; ++p) dst[(c0 + l64) + (r0 + g8 + 8 * p) * n] = tile[l64][g8 + 8 * p];
}

// 128 x 128 tile, 16-byte accesses, 512 threads, dynamic LDS
__global__ void __launch_bounds__(512) k_t128v(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    extern __shared__ double lds[];
    auto tile = [&](int a, int b) -> double& { return lds[a * 129 + b]; };  // [col][row]
    const size_t r0 = (size_t)blockIdx.x * 128, c0 = (size_t)blockIdx.y * 128;
    const int l2 = threadIdx.x & 63, grp = threadIdx.x >> 6;  // 8 groups of 64 lanes x 16 bytes = 128 elements
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int cc = grp + 8 * p;
        const v2 v = *(const v2*)(src + (r0 + 2 * l2) + (c0 + cc) * n);
        tile(cc, 2 * l2) = v.x;
        tile(cc, 2 * l2 + 1) = v.y;
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int rr = grp + 8 * p;
        const v2 v = {tile(2 * l2, rr), tile(2 * l2 + 1, rr)};
        *(v2*)(dst + (c0 + 2 * l2) + (r0 + rr) * n) = v;
    }
}

int main() {
    const size_t n = 8192;
    double *a, *b;
    hipMalloc(&a, n * n * 8); hipMalloc(&b, n * n * 8);
    hipMemset(a, 1, n * n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %6.3f TB/s\n", name, ms * 50.f, 2.0 * n * n * 8 / (ms / 20 * 1e-3) / 1e12);
    };
    run("t64   64x64, 8 B", [&] { hipLaunchKernelGGL(k_t<64>, dim3(n / 64, n / 64), dim3(256), 0, 0, a, b, n); });
    run("t64nt 64x64, 8 B, non-temporal", [&] { hipLaunchKernelGGL(k_t64nt, dim3(n / 64, n / 64), dim3(256), 0, 0, a, b, n); });
    run("copy  grid-stride 1024 thr NT v2, grid 4096", [&] { hipLaunchKernelGGL(k_copy_gs, dim3(4096), dim3(1024), 0, 0, a, b, n * n); });
    run("copy  chunk 1024 per block, plain", [&] { hipLaunchKernelGGL(k_copy_chunk, dim3(n * n / 1024), dim3(256), 0, 0, a, b, n * n); });
    run("t32   32x32, 8 B", [&] { hipLaunchKernelGGL(k_t<32>, dim3(n / 32, n / 32), dim3(256), 0, 0, a, b, n); });
    run("t64v  64x64, 16 B", [&] { hipLaunchKernelGGL(k_t64v, dim3(n / 64, n / 64), dim3(256), 0, 0, a, b, n); });
    run("t128x64  128x64, 8 B, 512 threads", [&] { hipLaunchKernelGGL(k_t128x64, dim3(n / 128, n / 64), dim3(512), 0, 0, a, b, n); });
    hipFuncSetAttribute((const void*)k_t128v, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8);
    run("t128v 128x128, 16 B, 512 threads", [&] { hipLaunchKernelGGL(k_t128v, dim3(n / 128, n / 128), dim3(512), 128 * 129 * 8, 0, a, b, n); });
    run("(copy b <- a, hipMemcpyAsync)", [&] { hipMemcpyAsync(b, a, n * n * 8, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
