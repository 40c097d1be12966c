// Developer tool: which physical CUs does a CU-masked stream use?  For each mask, launch many short blocks and count the
// distinct (XCC_ID, HW_ID.{se,sh,cu}) triples per XCD.
// hipcc --offload-arch=gfx950 -O3 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_probe(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
}
int main() {
    const int nblk = 8192;
    unsigned* out; CK(hipMalloc(&out, nblk * 8));
    std::vector<unsigned> h(nblk * 2);
    for (int mode = -1; mode < 7; ++mode) {
        uint32_t mask[8];
        for (int w = 0; w < 8; ++w) mask[w] = 0;
        for (unsigned i = 0; i < 256; ++i) {
            bool keep = true;
            if (mode == 0) keep = i < 192;
            if (mode == 1) keep = (i & 31u) < 24u;
            if (mode == 2) keep = i < 32;            // first 32 bits only
            if (mode == 3) keep = (i & 7u) == 0;     // every 8th bit
            if (mode == 4) keep = (i & 7u) != 0;     // everything but bits 0, 8, 16, ... (XCD 0 empty if bit i -> XCD i % 8)
            if (mode == 5) keep = (i & 7u) != 0 || i == 0;   // XCD 0 keeps one CU
            if (mode == 6) keep = (i & 7u) != 0 || i < 32;    // XCD 0 keeps four CUs
            if (keep) mask[i >> 5] |= 1u << (i & 31u);
        }
        hipStream_t s;
        if (mode < 0) CK(hipStreamCreate(&s)); else CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
        CK(hipMemsetAsync(out, 0xff, nblk * 8, s));
        hipLaunchKernelGGL(k_probe, dim3(nblk), dim3(64), 0, s, out, 2000);  // 20 us per block
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), out, nblk * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per;
        for (int b = 0; b < nblk; ++b) per[h[2 * b]].insert((h[2 * b + 1] >> 8) & 0xff);  // cu_id[11:8] sh[12] se[15:13]
        printf("mode %d:", mode);
        int total = 0;
        for (auto& kv : per) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  total %d\n", total);
        if (mode == 2 || mode == 3 || mode >= 4) { printf("   xcc0 ids:"); for (unsigned v : per[0]) printf(" %02x", v); printf("\n"); }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
