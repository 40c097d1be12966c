// Developer tool: does a long-running (sleeping) kernel on another stream slow a CHAIN of small dependent kernels on this one?
// hipcc --offload-arch=gfx950 -O3 dispatch_policy3.hip -o dispatch_policy3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ unsigned cu_key() {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);
}
__global__ void __launch_bounds__(64) k_holder(const unsigned* flag, int held, unsigned* seated, int poll) {
    const unsigned key = cu_key();
    if (!((key & 0xff) == 0 && (int)(key >> 8) < held)) return;
    if (threadIdx.x == 0) atomicAdd(seated, 1u);
    const long long t0 = wall_clock64();
    if (poll == 1) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(127);
    if (poll == 0) while (wall_clock64() - t0 < 30000000LL) __builtin_amdgcn_s_sleep(127);   // 0.3 s, no memory traffic
    if (poll == 2) while (*(volatile const unsigned*)flag == 0u && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(127);
}
__global__ void __launch_bounds__(256) k_small(double* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
int main() {
    hipStream_t sh, ss;
    CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    unsigned* flag; CK(hipMalloc(&flag, 64));
    double* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k_holder, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    for (int poll : {1, 0, 2})
        for (int lds_kb : {81, 0})
            for (int held : {0, 1, 8}) {
                CK(hipMemset(flag, 0, 64));
                if (held) hipLaunchKernelGGL(k_holder, dim3(1024), dim3(64), (size_t)lds_kb * 1024, sh, flag, held, flag + 1, poll);
                hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, ss, p, 4096); CK(hipStreamSynchronize(ss));
                for (int grid : {16, 256}) {
                    auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, ss, p, grid * 256);
                    CK(hipStreamSynchronize(ss));
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    printf("poll %d holder LDS %d KiB on %d CUs: chain of 2000 kernels of %d WGs: %.2f us per kernel\n", poll, lds_kb, held, grid, us / 2000);
                }
                const unsigned one = 1; CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
                CK(hipDeviceSynchronize());
            }
    return 0;
}
