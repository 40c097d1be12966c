// Developer tool: latency of a software grid barrier + candidate exchange between P workgroups,
// (a) all on one XCD (blocks b % 8 == 0 of an 8P grid) and (b) spread over the XCDs.
// Every cross-block access is a relaxed agent-scope atomic (sc1), spins are bounded.
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__global__ void __launch_bounds__(512) k_bar(int stride, int P, int iters, unsigned* counter, double* cand, unsigned* xcc,
                                             int* err, double* out) {
    if (blockIdx.x % stride) return;
    const int me = blockIdx.x / stride;
    if (threadIdx.x == 0) xcc[me] = xcc_id();
    __shared__ double s_sum;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        if (threadIdx.x == 0) {
            __hip_atomic_store(&cand[par * 1024 + me], (double)(it + me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt: the store has been acknowledged
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)P * (unsigned)(it + 1);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 2000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (*(volatile int*)err) return;
        if (threadIdx.x < 64) {
            double s = 0.0;
            for (int b = threadIdx.x; b < P; b += 64) s += __hip_atomic_load(&cand[par * 1024 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (threadIdx.x == 0) s_sum = s;
        }
        __syncthreads();
        acc += s_sum;
    }
    if (threadIdx.x == 0) out[me] = acc;
}

int main() {
    unsigned* counter; double* cand; unsigned* xcc; int* err; double* out;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&cand, 2048 * 8)); CK(hipMalloc(&xcc, 1024 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int stride : {8, 1}) {
        for (int P : {1, 4, 16, 32}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(cand, 0, 2048 * 8));
                CK(hipEventRecord(e0));
                k_bar<<<P * stride, 512>>>(stride, P, iters, counter, cand, xcc, err, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int herr; std::vector<unsigned> hx(P); std::vector<double> ho(P);
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, P * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ho.data(), out, P * 8, hipMemcpyDeviceToHost));
                double want = 0; for (int it = 0; it < iters; ++it) for (int b = 0; b < P; ++b) want += it + b;
                bool ok = true; for (int b = 0; b < P; ++b) ok = ok && ho[b] == want;
                printf("stride=%d P=%2d: %.3f us/step err=%d sums_ok=%d xcc:", stride, P, ms * 1e3 / iters, herr, (int)ok);
                for (int b = 0; b < P && b < 16; ++b) printf(" %u", hx[b]);
                printf("\n");
            }
        }
    }
    return 0;
}
