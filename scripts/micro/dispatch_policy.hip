// Developer tool: where does the dispatcher place the workgroups of a small kernel when most CUs hold a big-LDS block and a few CUs are
// kept nearly empty by a sleeping "holder" workgroup?  (round 5: seat holders made every small kernel of the LU chain 5-10x slower.)
// hipcc --offload-arch=gfx950 -O3 dispatch_policy.hip -o dispatch_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ unsigned cu_key() {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);
}
// big block: stays `ticks` (100 MHz) on its CU unless that CU is "held" (id byte 0 of the first `held` XCDs): then it leaves at once
__global__ void __launch_bounds__(512) k_big(long long ticks, int held, int busy) {
    extern __shared__ double lds[];
    const unsigned key = cu_key();
    if ((key & 0xff) == 0 && (int)(key >> 8) < held) return;
    const long long t0 = wall_clock64();
    double acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (busy) { for (int i = 0; i < 64; ++i) acc = acc * 1.0000001 + 0.5; } else __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 12345.678) lds[0] = acc;
}
__global__ void __launch_bounds__(64) k_holder(const unsigned* flag, int held, unsigned* seated) {
    extern __shared__ double lds[];
    const unsigned key = cu_key();
    if (!((key & 0xff) == 0 && (int)(key >> 8) < held)) return;
    if (threadIdx.x == 0) atomicAdd(seated, 1u);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(127);
}
__global__ void __launch_bounds__(256) k_small(unsigned* where, long long* when, long long ticks) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) { where[blockIdx.x] = cu_key(); when[2 * blockIdx.x] = t0; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) when[2 * blockIdx.x + 1] = wall_clock64();
}
int main(int argc, char** argv) {
    hipStream_t sb, sh, ss;
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    unsigned *flag, *where; long long* when;
    const int nsmall = 64;
    CK(hipMalloc(&flag, 64)); CK(hipMalloc(&where, nsmall * 4)); CK(hipMalloc(&when, nsmall * 16));
    CK(hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute((const void*)k_holder, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int mode = 0; mode < 6; ++mode) {
        const int held = (mode == 0 || mode == 3) ? 0 : (mode == 1 || mode == 4 ? 1 : 8);
        const int with_holder = mode >= 3;
        CK(hipMemset(flag, 0, 64));
        if (with_holder && held) { hipLaunchKernelGGL(k_holder, dim3(1024), dim3(64), 81 * 1024, sh, flag, held, flag + 1); }
        CK(hipDeviceSynchronize() == hipSuccess || true ? hipSuccess : hipSuccess);
        // 2048 big blocks of 100 us each: 8 rounds on 256 CUs; the small kernels run in the middle
        hipLaunchKernelGGL(k_big, dim3(4096), dim3(512), 84 * 1024, sb, 10000LL, with_holder ? 0 : held, 1);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        // wait ~300 us on the host so that the big kernel is in steady state
        { hipEvent_t t; CK(hipEventCreate(&t)); }
        std::vector<float> lat;
        std::map<unsigned, int> hist;
        std::vector<unsigned> hw(nsmall); std::vector<long long> hwhen(2 * nsmall);
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipEventRecord(e0, ss));
            hipLaunchKernelGGL(k_small, dim3(nsmall), dim3(256), 512, ss, where, when, 300LL);  // 3 us of "work"
            CK(hipEventRecord(e1, ss));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); lat.push_back(ms * 1000);
            CK(hipMemcpy(hw.data(), where, nsmall * 4, hipMemcpyDeviceToHost));
            for (unsigned v : hw) hist[v]++;
        }
        const unsigned one = 1; CK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        unsigned seated = 0; CK(hipMemcpy(&seated, flag + 1, 4, hipMemcpyDeviceToHost));
        std::sort(lat.begin(), lat.end());
        int on_held = 0, total = 0, distinct = (int)hist.size(), mx = 0;
        for (auto& kv : hist) { total += kv.second; if ((kv.first & 0xff) == 0 && (int)(kv.first >> 8) < held) on_held += kv.second; mx = std::max(mx, kv.second); }
        printf("mode %d: held %d CUs by %s (seated %u): small kernel (64 WGs x 256 thr, 3 us) median %.1f us min %.1f max %.1f; %d of %d WGs on held CUs, %d distinct CUs, busiest CU %d\n",
               mode, held, with_holder ? "holder" : "big blocks leaving", seated, lat[lat.size() / 2], lat[0], lat.back(), on_held, total, distinct, mx);
    }
    return 0;
}
