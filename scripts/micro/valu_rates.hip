// Developer micro-benchmark (GPU box): issue cost of the VALU instructions the fp64 elementwise / RNG kernels are made of, in
// cycles per wave64 instruction per SIMD.  One wave per SIMD on every CU (1024 waves), 8 independent dependency chains per
// lane so the pipe - not the latency - is measured; cycles = elapsed * clock / instructions per SIMD (clock from s_memtime ratio).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_rates.hip -o scripts/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096, CHAINS = 8;

#define KERNEL(NAME, DECL, BODY, SINK)                                                        \
    __global__ void __launch_bounds__(64) NAME(double* out, unsigned long long seed) {          \
        DECL;                                                                                  \
        for (int it = 0; it < ITERS; ++it) {                                                   \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { BODY; }                         \
        }                                                                                      \
        double acc = 0.0;                                                                      \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) acc += (double)(SINK);               \
        if (acc == 12345.678) out[threadIdx.x] = acc;                                          \
    }

#define DDECL double x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = 1.0 + 1e-9 * (double)(threadIdx.x + c + (seed & 7))
#define UDECL unsigned u[CHAINS]; for (int c = 0; c < CHAINS; ++c) u[c] = (unsigned)(threadIdx.x * 2654435761u + c + seed)
#define LDECL unsigned long long l[CHAINS]; for (int c = 0; c < CHAINS; ++c) l[c] = seed * 6364136223846793005ULL + threadIdx.x + c

KERNEL(k_fma_f64, DDECL, asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_mul_f64, DDECL, asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_add_f64, DDECL, asm volatile("v_add_f64 %0, %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_rsq_f64, DDECL, asm volatile("v_rsq_f64 %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_rcp_f64, DDECL, asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_sqrt_f64, DDECL, asm volatile("v_sqrt_f64 %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_rndne_f64, DDECL, asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_ldexp_f64, DDECL, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[c])), x[c])
KERNEL(k_frexp_mant_f64, DDECL, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x[c])), x[c])
KERNEL(k_cvt_f64_u32, DDECL; UDECL, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[c]) : "v"(u[c])); u[c] += 1, x[c])
KERNEL(k_cvt_i32_f64, DDECL; UDECL, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[c]) : "v"(x[c])); x[c] += 1.0, u[c])
KERNEL(k_cvt_f32_f64, DDECL; float f[CHAINS], asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(x[c])); x[c] += 1.0, f[c])
KERNEL(k_mul_lo_u32, UDECL, asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[c])), u[c])
KERNEL(k_mul_hi_u32, UDECL, asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(u[c])), u[c])
KERNEL(k_mul_u32_u24, UDECL, asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(u[c])), u[c])
KERNEL(k_mad_u64_u32, LDECL; UDECL, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(l[c]) : "v"(u[c]) : "vcc"), l[c])
KERNEL(k_and_b32, UDECL, asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(u[c])), u[c])
KERNEL(k_add_u32, UDECL, asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[c])), u[c])
KERNEL(k_alignbit, UDECL, asm volatile("v_alignbit_b32 %0, %0, %0, 11" : "+v"(u[c])), u[c])
KERNEL(k_lshl_b64, LDECL, asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(l[c])), l[c])
KERNEL(k_fma_f32, float f[CHAINS]; for (int c = 0; c < CHAINS; ++c) f[c] = 1.0f + threadIdx.x, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c])), f[c])
KERNEL(k_cndmask, UDECL, asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[c]) : : ), u[c])
KERNEL(k_cndmask_sgpr, UDECL; unsigned long long mask = seed * 0x5555555555555555ULL,
       asm volatile("v_cndmask_b32_e64 %0, %0, %0, %1" : "+v"(u[c]) : "s"(mask)), u[c])
KERNEL(k_cndmask_2src, UDECL; unsigned w[CHAINS]; for (int c = 0; c < CHAINS; ++c) w[c] = u[c] * 3u; unsigned long long mask = seed * 0x5555555555555555ULL,
       asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(w[c]), "s"(mask)), u[c])
KERNEL(k_mov_b32, UDECL; unsigned w[CHAINS]; for (int c = 0; c < CHAINS; ++c) w[c] = u[c] * 3u,
       asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(w[c])); w[c] = u[c], u[c])
KERNEL(k_cmp_f64, DDECL, asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(x[c]) : "vcc"), x[c])
typedef double v2dd __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k_ds_read_b128(double* out, unsigned long long seed) {
    __shared__ v2dd lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) {
        lds[i].x = 1.0;
        lds[i].y = 2.0;
    }
    __syncthreads();
    unsigned u[CHAINS];
    v2dd d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) u[c] = (unsigned)(threadIdx.x * 2654435761u + c + seed);
    double acc = 0.0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(d[c]) : "v"((u[c] & 1023u) * 16u));
            u[c] += 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc += d[c].x;
    }
    if (acc == 12345.678) out[threadIdx.x] = acc;
}

int main() {
    double* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clock_hz = p.clockRate * 1e3;
    printf("device: %s, %d CUs, clockRate %.0f MHz, memoryClockRate %.0f MHz, bus %d bits, L2 %d KiB\n", p.name, cus, p.clockRate / 1e3,
           p.memoryClockRate / 1e3, p.memoryBusWidth, p.l2CacheSize / 1024);
    auto run = [&](const char* name, auto kernel) {
        printf("%-18s", name);
        for (int wps : {1, 2, 4}) {  // waves per SIMD: one wave alone issues a VALU instruction every ~5 cycles whatever its rate
            const int waves = cus * 4 * wps;
            hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, out, 1ULL);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, out, (unsigned long long)(r + 2));
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("  %d wave%s/SIMD: %6.2f", wps, wps > 1 ? "s" : " ", (double)best * 1e-3 * clock_hz / ((double)ITERS * CHAINS * wps));
        }
        printf("   cycles per wave64 instruction per SIMD\n");
    };
#define RUN(k) run(#k, k)
    RUN(k_fma_f64); RUN(k_mul_f64); RUN(k_add_f64); RUN(k_rsq_f64); RUN(k_rcp_f64); RUN(k_sqrt_f64); RUN(k_rndne_f64); RUN(k_ldexp_f64);
    RUN(k_frexp_mant_f64); RUN(k_cvt_f64_u32); RUN(k_cvt_i32_f64); RUN(k_cvt_f32_f64); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mul_u32_u24);
    RUN(k_mad_u64_u32); RUN(k_and_b32); RUN(k_add_u32); RUN(k_alignbit); RUN(k_lshl_b64); RUN(k_fma_f32); RUN(k_cndmask); RUN(k_cndmask_sgpr); RUN(k_cndmask_2src); RUN(k_mov_b32); RUN(k_cmp_f64);
    RUN(k_ds_read_b128);
    return 0;
}
