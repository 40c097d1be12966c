// Developer micro-benchmark (GPU box): write-only kernels over 512 MiB - which store pattern reaches the HBM write ceiling?
//   a  grid-stride, 1024-thread blocks, non-temporal 16-byte stores (k_fill)
//   b  the same with plain 16-byte stores
//   c  one block = 1024 consecutive doubles at one go (256 threads x 4 plain 8-byte stores), grid = all chunks (k_index_copy's shape)
//   d  as c with non-temporal stores          e  as c with 16-byte plain stores (256 threads x 2 vectors)
//   f  grid-stride, 256-thread blocks, cap 16 blocks per CU, plain 8-byte stores (k_linspace's shape)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/write_patterns.hip -o scripts/micro/write_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) ka(double* o, size_t n, double v) {
    const size_t nv = n >> 1, st = (size_t)gridDim.x * 1024;
    const v2 vv = {v, v};
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < nv; i += st) __builtin_nontemporal_store(vv, (v2*)o + i);
}
__global__ void __launch_bounds__(1024) kb(double* o, size_t n, double v) {
    const size_t nv = n >> 1, st = (size_t)gridDim.x * 1024;
    const v2 vv = {v, v};
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < nv; i += st) ((v2*)o)[i] = vv;
}
__global__ void __launch_bounds__(256) kc(double* o, size_t n, double v) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (b + e * 256 < n) o[b + e * 256] = v;
}
__global__ void __launch_bounds__(256) kd(double* o, size_t n, double v) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (b + e * 256 < n) __builtin_nontemporal_store(v, o + b + e * 256);
}
__global__ void __launch_bounds__(256) ke(double* o, size_t n, double v) {
    const size_t b = (size_t)blockIdx.x * 512 + threadIdx.x;
    const v2 vv = {v, v};
#pragma unroll
    for (int e = 0; e < 2; ++e) if (2 * (b + e * 256) < n) ((v2*)o)[b + e * 256] = vv;
}
__global__ void __launch_bounds__(256) kf(double* o, size_t n, double v) {
    const size_t st = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) o[i] = v;
}
int main() {
    const size_t n = (size_t)8192 * 8192;
    double* o; hipMalloc(&o, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-50s %7.1f us  %6.3f TB/s\n", name, ms * 50.f, n * 8 / (ms / 20 * 1e-3) / 1e12);
    };
    for (int g : {2048, 4096, 8192}) {
        char nm[80];
        snprintf(nm, 80, "a grid-stride 1024 thr NT v2, grid %d", g); run(nm, [&] { hipLaunchKernelGGL(ka, dim3(g), dim3(1024), 0, 0, o, n, 1.0); });
        snprintf(nm, 80, "b grid-stride 1024 thr plain v2, grid %d", g); run(nm, [&] { hipLaunchKernelGGL(kb, dim3(g), dim3(1024), 0, 0, o, n, 1.0); });
    }
    run("c chunk 1024 doubles per block, plain 8 B", [&] { hipLaunchKernelGGL(kc, dim3(n / 1024), dim3(256), 0, 0, o, n, 1.0); });
    run("d chunk 1024 doubles per block, NT 8 B", [&] { hipLaunchKernelGGL(kd, dim3(n / 1024), dim3(256), 0, 0, o, n, 1.0); });
    run("e chunk 1024 doubles per block, plain 16 B", [&] { hipLaunchKernelGGL(ke, dim3(n / 1024), dim3(256), 0, 0, o, n, 1.0); });
    for (int g : {4096, 16384}) {
        char nm[80];
        snprintf(nm, 80, "f grid-stride 256 thr plain 8 B, grid %d", g); run(nm, [&] { hipLaunchKernelGGL(kf, dim3(g), dim3(256), 0, 0, o, n, 1.0); });
    }
    return 0;
}
