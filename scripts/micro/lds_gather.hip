// Developer micro-benchmark (GPU box): what a random table look-up costs a CU, in cycles per wave64 instruction per CU (all four SIMDs
// issuing, 4 waves per SIMD), for the access forms the Box-Muller step could use: ds_read_b128 / 2 x ds_read_b64 with per-lane random
// indices into a 513- or 129-entry table, the same with an 8-fold replicated bank-striped table, and ds_bpermute_b32 (crossbar only).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/lds_gather.hip -o scripts/micro/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048, CH = 4;
typedef double v2dd __attribute__((ext_vector_type(2)));

template <int ENTRIES, int MODE>  // MODE 0: b128 random; 1: two b64 (SoA); 2: b128 replicated x8 (lane & 7 picks the copy); 3: b128 same index in all lanes
__global__ void __launch_bounds__(256) k_gather(double* out, unsigned long long seed) {
    extern __shared__ double lds[];
    const int total = MODE == 2 ? ENTRIES * 8 * 2 : ENTRIES * 2;
    for (int i = threadIdx.x; i < total; i += 256) lds[i] = 1.0 + i;
    __syncthreads();
    unsigned long long s[CH];
    for (int c = 0; c < CH; ++c) s[c] = (seed + threadIdx.x * 977 + c) * 6364136223846793005ULL + 1442695040888963407ULL;
    double acc = 0.0;
    for (int it = 0; it < ITERS; ++it) {
        v2dd d[CH];
        double e0[CH], e1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            s[c] = s[c] * 6364136223846793005ULL + 1442695040888963407ULL;
            unsigned j = (unsigned)(((s[c] >> 33) * (unsigned long long)ENTRIES) >> 31);
            if (MODE == 3) j = __builtin_amdgcn_readfirstlane(j);
            if (MODE == 0 || MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(d[c]) : "v"(j * 16u));
            if (MODE == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(d[c]) : "v"((j * 8u + (threadIdx.x & 7u)) * 16u));
            if (MODE == 1) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(e0[c]) : "v"(j * 8u));
                asm volatile("ds_read_b64 %0, %1 offset:8192" : "=v"(e1[c]) : "v"(j * 8u));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += MODE == 1 ? e0[c] + e1[c] : d[c].x + d[c].y;
    }
    if (acc == 12345.678) out[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_bpermute(double* out, unsigned long long seed) {
    unsigned long long s[CH];
    for (int c = 0; c < CH; ++c) s[c] = (seed + threadIdx.x * 977 + c) * 6364136223846793005ULL + 1442695040888963407ULL;
    int tab = threadIdx.x * 3 + 1;
    int acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        int d[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            s[c] = s[c] * 6364136223846793005ULL + 1442695040888963407ULL;
            const unsigned j = (unsigned)(s[c] >> 58);
            asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(d[c]) : "v"(j * 4u), "v"(tab));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int c = 0; c < CH; ++c) acc += d[c];
    }
    if (acc == 12345678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_baseline(double* out, unsigned long long seed) {  // the index arithmetic alone
    unsigned long long s[CH];
    for (int c = 0; c < CH; ++c) s[c] = (seed + threadIdx.x * 977 + c) * 6364136223846793005ULL + 1442695040888963407ULL;
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            s[c] = s[c] * 6364136223846793005ULL + 1442695040888963407ULL;
            acc += (unsigned)(((s[c] >> 33) * 513ull) >> 31);
        }
    }
    if (acc == 12345678) out[threadIdx.x] = acc;
}

int main() {
    double* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clock_hz = p.clockRate * 1e3;
    auto run = [&](const char* name, auto kernel, size_t lds_bytes, int per_iter) {
        const int blocks = cus * 4;  // 4 blocks of 256 threads per CU: 4 waves per SIMD
        if (lds_bytes > 40960) { printf("%-34s skipped (LDS)\n", name); return; }
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds_bytes, 0, out, 1ULL);
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds_bytes, 0, out, (unsigned long long)(r + 2));
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        // 16 waves per CU, each ITERS * CH * per_iter instructions of the kind measured
        printf("%-34s %7.2f cycles per wave-instruction per CU   (%.1f us)\n", name, (double)best * 1e-3 * clock_hz / ((double)ITERS * CH * per_iter * 16), best * 1e3);
    };
    run("index arithmetic alone", k_baseline, 0, 1);
    run("ds_read_b128 random 513", k_gather<513, 0>, 513 * 16, 1);
    run("ds_read_b128 random 129", k_gather<129, 0>, 129 * 16, 1);
    run("ds_read_b128 uniform index", k_gather<513, 3>, 513 * 16, 1);
    run("2 x ds_read_b64 random 513 (SoA)", k_gather<513, 1>, 16384, 2);
    run("ds_read_b128 random 129 x8 striped", k_gather<129, 2>, 129 * 16 * 8, 1);
    run("ds_read_b128 random 257 x8 striped", k_gather<257, 2>, 257 * 16 * 8, 1);
    run("ds_bpermute_b32 random", k_bpermute, 0, 1);
    return 0;
}
