// Developer tool: what does a workgroup that does NOT fit the CU the dispatcher picked cost?  Holders (LDS- or VGPR-based) sit on 0/1/8 CUs;
// test kernels that cannot share a CU with a holder are timed: a 64-workgroup kernel of 3 us (launch latency) and a 4096-workgroup kernel
// of 20 us blocks (throughput).
// hipcc --offload-arch=gfx950 -O3 dispatch_policy2.hip -o dispatch_policy2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ unsigned cu_key() {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);
}
__device__ bool is_held(int held) { const unsigned key = cu_key(); return (key & 0xff) == 0 && (int)(key >> 8) < held; }
__global__ void __launch_bounds__(64) k_holder_lds(const unsigned* flag, int held, unsigned* seated) {
    if (!is_held(held)) return;
    if (threadIdx.x == 0) atomicAdd(seated, 1u);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(127);
}
// one wave per SIMD, each holding 320 VGPRs: a wave that needs more than 192 registers no longer fits that SIMD
__global__ void __launch_bounds__(256) k_holder_vgpr(const unsigned* flag, int held, unsigned* seated) {
    if (!is_held(held)) return;
    asm volatile("v_mov_b32 v250, 0\n\tv_accvgpr_write_b32 a60, v250" ::: "v250", "a60");
    if (threadIdx.x == 0) atomicAdd(seated, 1u);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(127);
}
__global__ void __launch_bounds__(256) k_test_lds(long long ticks) {  // launched with 132 KiB of LDS
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void __launch_bounds__(512) k_test_vgpr(long long ticks, double* sink) {  // 512 threads x ~150 registers: 2 waves per SIMD, 300 registers
    asm volatile("v_mov_b32 v149, 0" ::: "v149");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main() {
    hipStream_t sh, ss;
    CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    unsigned* flag; CK(hipMalloc(&flag, 64));
    CK(hipFuncSetAttribute((const void*)k_holder_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipFuncSetAttribute((const void*)k_test_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipFuncSetAttribute((const void*)k_test_vgpr, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind)       // 0: LDS holder, 1: VGPR holder
        for (int held : {0, 1, 8}) {
            CK(hipMemset(flag, 0, 64));
            if (held) {
                if (kind == 0) hipLaunchKernelGGL(k_holder_lds, dim3(1024), dim3(64), 81 * 1024, sh, flag, held, flag + 1);
                else hipLaunchKernelGGL(k_holder_vgpr, dim3(1024), dim3(256), 0, sh, flag, held, flag + 1);
            }
            // give the holders time to seat
            hipLaunchKernelGGL(k_test_lds, dim3(1), dim3(256), 0, ss, 20000LL); CK(hipStreamSynchronize(ss));
            for (int test = 0; test < 4; ++test) {
                // test 0: 64 WGs needing 132 KiB LDS (3 us); 1: 4096 WGs of 84 KiB LDS (20 us each); 2: 64 WGs of 512 threads x 150 VGPRs (3 us); 3: 4096 such WGs with 84 KiB (20 us)
                std::vector<float> lat;
                for (int rep = 0; rep < 12; ++rep) {
                    CK(hipEventRecord(e0, ss));
                    if (test == 0) hipLaunchKernelGGL(k_test_lds, dim3(64), dim3(256), 132 * 1024, ss, 300LL);
                    if (test == 1) hipLaunchKernelGGL(k_test_lds, dim3(4096), dim3(256), 84 * 1024, ss, 2000LL);
                    if (test == 2) hipLaunchKernelGGL(k_test_vgpr, dim3(64), dim3(512), 0, ss, 300LL, (double*)nullptr);
                    if (test == 3) hipLaunchKernelGGL(k_test_vgpr, dim3(4096), dim3(512), 84 * 1024, ss, 2000LL, (double*)nullptr);
                    CK(hipEventRecord(e1, ss));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); lat.push_back(ms * 1000);
                }
                std::sort(lat.begin(), lat.end());
                printf("holder %s on %d CUs, test %d: median %.1f us (min %.1f max %.1f)\n", kind ? "VGPR" : "LDS", held, test, lat[lat.size() / 2], lat[0], lat.back());
            }
            const unsigned one = 1; CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            unsigned seated = 0; CK(hipMemcpy(&seated, flag + 1, 4, hipMemcpyDeviceToHost));
            printf("   (seated %u)\n", seated);
        }
    return 0;
}
