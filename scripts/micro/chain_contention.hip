// Developer tool: what slows the LU chain's small kernels beside the update streams' dgemm - the matrix pipe (issue / fp64 datapath) or
// memory latency under load?  Two chain-like kernels are timed back to back (200 launches) alone, beside an MFMA-only background, beside
// a streaming-copy background, and beside both:
//   K1 "latency": 16 workgroups x 256 threads, 16 dependent global loads each (pointer chase) - the shape of k_laswp_lists
//   K2 "fp64":    256 workgroups x 64 threads, 2048 dependent fp64 FMAs with an LDS operand each - the shape of k_rp_below
//   K3 "int":     256 workgroups x 64 threads, 2048 dependent 32-bit integer multiply-adds
// hipcc --offload-arch=gfx950 -O3 chain_contention.hip -o chain_contention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_bg_mfma(const int* stop, double* sink) {
    extern __shared__ double lds[];
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-9, b = 1.0 + threadIdx.x * 1e-12;
    for (int it = 0; it < 200000; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        if ((it & 15) == 0 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345) sink[0] = s;
}
__global__ void __launch_bounds__(256) k_bg_copy(const int* stop, const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    for (int it = 0; it < 100000; ++it) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}
__global__ void __launch_bounds__(256) k_latency(const unsigned* __restrict__ next, unsigned* out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    unsigned p = (blockIdx.x * 256 + threadIdx.x) * 977u;
    for (int i = 0; i < 16; ++i) p = next[p & 0xffffffu];
    out[blockIdx.x * 256 + threadIdx.x] = p;
}
__global__ void __launch_bounds__(64) k_fp64(double* out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    __shared__ double u[64];
    u[threadIdx.x] = 1.0 + threadIdx.x * 1e-9;
    __syncthreads();
    double x = threadIdx.x;
    for (int i = 0; i < 2048; ++i) x = fma(x, u[i & 63], 1e-3);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) k_int(unsigned* out, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    __shared__ unsigned u[64];
    u[threadIdx.x] = 3 + threadIdx.x;
    __syncthreads();
    unsigned x = threadIdx.x;
    for (int i = 0; i < 2048; ++i) x = x * u[i & 63] + 7u;
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
    hipStream_t sa, sb, sc;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    int* stop; CK(hipMalloc(&stop, 64));
    double* sink; CK(hipMalloc(&sink, 1 << 20));
    const size_t ncopy = (size_t)1 << 26;  // 1 GiB of double2 each way
    double2 *src, *dst; CK(hipMalloc(&src, ncopy * 16)); CK(hipMalloc(&dst, ncopy * 16)); CK(hipMemset(src, 0, ncopy * 16));
    unsigned* next; CK(hipMalloc(&next, (size_t)(1 << 24) * 4));
    { std::vector<unsigned> h(1 << 24); unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; } CK(hipMemcpy(next, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k_bg_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int bg : {0, 1, 4}) {   // 0 none, 1 mfma one block per CU, 2 copy, 3 both, 4 mfma TWO blocks per CU
        CK(hipMemset(stop, 0, 64));
        if (bg == 1 || bg == 3) hipLaunchKernelGGL(k_bg_mfma, dim3(256), dim3(512), 84 * 1024, sa, stop, sink);
        if (bg == 4) hipLaunchKernelGGL(k_bg_mfma, dim3(512), dim3(512), 64 * 1024, sa, stop, sink);
        if (bg == 2 || bg == 3) hipLaunchKernelGGL(k_bg_copy, dim3(2048), dim3(256), 0, sb, stop, src, dst, ncopy);
        hipLaunchKernelGGL(k_int, dim3(16), dim3(64), 0, sc, out, 0); CK(hipStreamSynchronize(sc));
        for (int prio = 0; prio < 2; ++prio)
        for (int kind = 0; kind < 3; ++kind) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 200; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_latency, dim3(16), dim3(256), 0, sc, next, out, prio);
                if (kind == 1) hipLaunchKernelGGL(k_fp64, dim3(256), dim3(64), 0, sc, (double*)out, prio);
                if (kind == 2) hipLaunchKernelGGL(k_int, dim3(256), dim3(64), 0, sc, out, prio);
            }
            CK(hipStreamSynchronize(sc));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("background %d (%s): prio %d %s kernel %.1f us per launch\n", bg, bg == 0 ? "none" : bg == 1 ? "mfma f64, one 8-wave block per CU" : bg == 2 ? "streaming copy" : bg == 3 ? "mfma + copy" : "mfma f64, two blocks per CU", prio ? 3 : 0,
                   kind == 0 ? "latency" : kind == 1 ? "fp64   " : "int    ", us / 200);
        }
        const int one = 1; CK(hipMemcpy(stop, &one, 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
    }
    return 0;
}
