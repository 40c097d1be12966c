// Developer tool (GPU box): issue rate of v_mfma_f64_16x16x4_f64 on gfx950 by waves per SIMD and by the number of independent accumulators
// a wave cycles through - what one block per CU can and cannot reach on the fp64 matrix pipe.
// Build + run: hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_f64_issue scripts/micro/mfma_f64_issue.hip && /tmp/mfma_f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(1024) k(double* out, int iters, long long* cyc) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd) {
    double* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 4000;
    const int threads = 256 * waves_per_simd;  // one block per CU: waves_per_simd waves on each of the 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<NACC><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    const double cycles = ms * 1e-3 * 2.4e9;
    std::printf("waves/SIMD %d, %d accumulators: %.1f cycles per MFMA per SIMD at 2.4 GHz (%.1f TFLOP/s on 256 CUs)\n", waves_per_simd, NACC,
                cycles / mfma_per_simd, 256.0 * 4 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<1>(w);
        run<2>(w);
        run<4>(w);
        run<8>(w);
    }
    return 0;
}
