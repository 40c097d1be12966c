// Developer tool: per-step latency of the LU panel's candidate exchange (data-tagged 8-byte records, all-gather among P
// workgroups + one dependent 64-word read of the winner's row) for different cache-policy bits on the stores and loads,
// with the P workgroups on ONE XCD (blocks b % 8 == 0 of an 8P grid) or spread over the XCDs, idle or beside a
// streaming kernel.  Every word carries a freshness bit; spins are bounded; sums are checked.
// hipcc --offload-arch=gfx950 -O3 xcd_exchange.hip -o xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int SB>
__device__ __forceinline__ void st64(u64* p, u64 v) {
    if (SB == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (SB == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (SB == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (SB == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (SB == 4) asm volatile("global_atomic_swap_x2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");  // RMW at the coherence point, no return
    if (SB == 5) asm volatile("global_atomic_swap_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (SB == 6) asm volatile("global_store_dwordx2 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LB>
__device__ __forceinline__ u64 ld64(const u64* p) {
    u64 v;
    if (LB == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LB == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LB == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SB, int LB>
__global__ void __launch_bounds__(512) k_xchg(int stride, int P, int iters, u64* rec, u64* rows, unsigned* xcc, int* err, double* out) {
    if (blockIdx.x % stride) return;
    const int me = blockIdx.x / stride;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) xcc[me] = xcc_id();
    __shared__ double s_sum;
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        const u64 fresh = (u64)(((it >> 1) & 1) ^ 1) << 63;
        if (wv == 0) {
            // publish: 64 row words + the record
            st64<SB>(&rows[((size_t)par * 256 + me) * 64 + lane], ((u64)(it * 64 + lane + me) & 0xffffffffull) | fresh);
            if (lane == 0) st64<SB>(&rec[par * 256 + me], ((u64)(unsigned)(it + me)) | fresh);
            // gather all records
            u64 best = 0; int bb = 0, bad = 0;
            for (int b = lane; b < P; b += 64) {
                u64 w; int spins = 0;
                for (;;) {
                    w = ld64<LB>(&rec[par * 256 + b]);
                    if (((w ^ fresh) >> 63) == 0) break;
                    if (++spins > 300000) { bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                w &= ~((u64)1 << 63);
                if (w >= best) { best = w; bb = b; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const u64 ob = __shfl_down(best, off, 64); const int obb = __shfl_down(bb, off, 64);
                if (ob > best || (ob == best && obb > bb)) { best = ob; bb = obb; }
            }
            bb = __shfl(bb, 0, 64);
            // dependent read of the winner's row
            u64 w; int spins = 0;
            for (;;) {
                w = ld64<LB>(&rows[((size_t)par * 256 + bb) * 64 + lane]);
                if (((w ^ fresh) >> 63) == 0) break;
                if (++spins > 300000) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            double s = (double)(w & 0xffffffffull);
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (__any(bad)) { if (lane == 0) { s_bad = 1; *err = 1; } }
            if (lane == 0) s_sum = s;
        }
        __syncthreads();
        if (s_bad) return;
        acc += s_sum;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[me] = acc;
}

__global__ void __launch_bounds__(256) k_stream(const double4* __restrict__ a, double4* __restrict__ b, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            double4 v = a[i]; v.x += 1.0; b[i] = v;
        }
}

template <int SB, int LB>
static int run(const char* tag, bool loaded, hipStream_t s0, hipStream_t s1, u64* rec, u64* rows, unsigned* xcc, int* err, double* out,
               double4* big_a, double4* big_b, size_t big_n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 3000;
    for (int stride : {8, 1}) {
        for (int P : {2, 8, 32}) {
            CK(hipMemsetAsync(err, 0, 4, s0)); CK(hipMemsetAsync(rec, 0, 512 * 8, s0)); CK(hipMemsetAsync(rows, 0, 512 * 64 * 8, s0));
            CK(hipStreamSynchronize(s0));
            if (loaded) k_stream<<<1024, 256, 0, s1>>>(big_a, big_b, big_n, 6);
            CK(hipEventRecord(e0, s0));
            k_xchg<SB, LB><<<P * stride, 512, 100 * 1024, s0>>>(stride, P, iters, rec, rows, xcc, err, out);
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipDeviceSynchronize());
            int herr; std::vector<unsigned> hx(P); std::vector<double> ho(P);
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, P * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ho.data(), out, P * 8, hipMemcpyDeviceToHost));
            double want = 0;
            for (int it = 0; it < iters; ++it) for (int l = 0; l < 64; ++l) want += (double)((it * 64 + l + (P - 1)) & 0xffffffff);
            bool ok = true; for (int b = 0; b < P; ++b) ok = ok && ho[b] == want;
            int nx = 0; { bool seen[16] = {0}; for (int b = 0; b < P; ++b) if (!seen[hx[b]]) { seen[hx[b]] = true; ++nx; } }
            printf("%-28s %s stride=%d P=%2d: %.3f us/step err=%d ok=%d xcds=%d\n", tag, loaded ? "loaded" : "idle  ", stride, P, ms * 1e3 / iters, herr,
                   (int)ok, nx);
            fflush(stdout);
        }
    }
    return 0;
}

int main() {
    u64 *rec, *rows; unsigned* xcc; int* err; double* out;
    CK(hipMalloc(&rec, 512 * 8)); CK(hipMalloc(&rows, 512 * 64 * 8)); CK(hipMalloc(&xcc, 1024 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 1024 * 8));
    const size_t big_n = (size_t)1 << 26;  // 2 GiB per array
    double4 *ba, *bb; CK(hipMalloc(&ba, big_n * 32)); CK(hipMalloc(&bb, big_n * 32)); CK(hipMemset(ba, 0, big_n * 32));
    hipStream_t s0, s1; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
    CK(hipFuncSetAttribute((const void*)k_xchg<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void*)k_xchg<6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int loaded = 0; loaded < 2; ++loaded) {
        if (run<1, 1>("store sc1 / load sc1", loaded, s0, s1, rec, rows, xcc, err, out, ba, bb, big_n)) return 1;
        if (run<4, 1>("atomic swap sc1 / load sc1", loaded, s0, s1, rec, rows, xcc, err, out, ba, bb, big_n)) return 1;
        if (run<5, 1>("atomic swap / load sc1", loaded, s0, s1, rec, rows, xcc, err, out, ba, bb, big_n)) return 1;
        if (run<6, 1>("store nt sc1 / load sc1", loaded, s0, s1, rec, rows, xcc, err, out, ba, bb, big_n)) return 1;
        if (run<0, 1>("store plain / load sc1", loaded, s0, s1, rec, rows, xcc, err, out, ba, bb, big_n)) return 1;
    }
    return 0;
}
