// Developer micro-benchmark: per-launch cost of small dependent kernels on one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(double* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1; }
__global__ void k_touch(double* p, size_t ld) {   // each block reads + writes one element per thread (1 trip)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[i] + 1.0;
}
__global__ void k_chain2(double* p, int* idx, size_t n) {  // dependent: idx -> p[idx] -> write
    __shared__ double s;
    if (threadIdx.x == 0) s = p[idx[0] % n];
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    p[i] = p[i] + s * 0.0;
}
__global__ void k_cols(double* A, size_t lda, int ncols) {  // one row per thread, walk ncols columns (R+W)
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < ncols; ++j) A[r + (size_t)j * lda] += 1.0;
}
template <int NC>
__global__ void k_cols_u(double* A, size_t lda) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) v[j] = A[r + (size_t)j * lda];
#pragma unroll
    for (int j = 0; j < NC; ++j) A[r + (size_t)j * lda] = v[j] + 1.0;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t rows = 8192, lda = 8192 + 32, cols = 64;
    double* A; CK(hipMalloc(&A, lda * cols * 8)); CK(hipMemset(A, 0, lda * cols * 8));
    int* idx; CK(hipMalloc(&idx, 4)); CK(hipMemset(idx, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.2f us/launch\n", name, ms * 1000.f / N);
        return 0;
    };
    for (int blocks : {1, 16, 128}) {
        char nm[64];
        snprintf(nm, 64, "empty %d x 256", blocks); run(nm, [&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, A); });
        snprintf(nm, 64, "touch %d x 256", blocks); run(nm, [&] { hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, A, lda); });
        snprintf(nm, 64, "chain2 %d x 256", blocks); run(nm, [&] { hipLaunchKernelGGL(k_chain2, dim3(blocks), dim3(256), 0, s, A, idx, rows); });
    }
    for (int nc : {1, 8, 16, 32, 64}) {
        char nm[64];
        snprintf(nm, 64, "cols loop nc=%d, 128x64", nc); run(nm, [&] { hipLaunchKernelGGL(k_cols, dim3(128), dim3(64), 0, s, A, lda, nc); });
        snprintf(nm, 64, "cols loop nc=%d, 32x256", nc); run(nm, [&] { hipLaunchKernelGGL(k_cols, dim3(32), dim3(256), 0, s, A, lda, nc); });
    }
    run("cols unrolled 16, 128x64", [&] { hipLaunchKernelGGL(k_cols_u<16>, dim3(128), dim3(64), 0, s, A, lda); });
    run("cols unrolled 32, 128x64", [&] { hipLaunchKernelGGL(k_cols_u<32>, dim3(128), dim3(64), 0, s, A, lda); });
    run("cols unrolled 64, 128x64", [&] { hipLaunchKernelGGL(k_cols_u<64>, dim3(128), dim3(64), 0, s, A, lda); });
    run("cols unrolled 16, 32x256", [&] { hipLaunchKernelGGL(k_cols_u<16>, dim3(32), dim3(256), 0, s, A, lda); });
    // unpadded lda
    run("cols unrolled 32, 128x64, lda=8192", [&] { hipLaunchKernelGGL(k_cols_u<32>, dim3(128), dim3(64), 0, s, A, (size_t)8192); });
    return 0;
}
