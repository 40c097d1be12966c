// Developer micro-benchmark (GPU box): cycle cost of the pieces of one wave-local partial-pivoting step (lu.hip gepp8_step) - the DPP
// wave maximum, the winner's row by v_readlane, the reciprocal chain, the elimination - one wave per SIMD, dependent iterations.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off scripts/micro/gepp_step.hip -o scripts/micro/gepp_step
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned umax32(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = umax32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
constexpr int ITERS = 2048;
// MODE bits: 1 wave max, 2 readlanes of the winner's 7 values + reciprocal, 4 reciprocal chain, 8 elimination
template <int MODE>
__global__ void __launch_bounds__(256) k_step(double* out, unsigned long long* cyc, double seed) {
    const int lane = threadIdx.x & 63;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + 0.001 * (threadIdx.x * 8 + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int wl = 5;
    for (int it = 0; it < ITERS; ++it) {
        unsigned key = ((unsigned)__double2hiint(fabs(v[0])) & ~0xffu) | (unsigned)(255 - lane);
        double rown = 1.0;
        if (MODE & 4) {
            rown = __builtin_amdgcn_rcp(v[0]);
            rown = __builtin_fma(rown, __builtin_fma(-v[0], rown, 1.0), rown);
            rown = __builtin_fma(rown, __builtin_fma(-v[0], rown, 1.0), rown);
        }
        if (MODE & 1) {
            const unsigned wm = wave_max_u32(key);
            wl = 255 - (int)(wm & 0xffu);
        } else {
            wl = (wl + 1) & 63;
        }
        double p[8], r = rown;
        if (MODE & 2) {
            for (int i = 1; i < 8; ++i) p[i] = readlane_f64(v[i], wl);
            r = readlane_f64(rown, wl);
        } else {
            for (int i = 1; i < 8; ++i) p[i] = 0.5 + i;
        }
        if (MODE & 8) {
            if (lane != wl) {
                const double f = v[0] * r;
                for (int i = 1; i < 8; ++i) v[i] = __builtin_fma(-f, p[i], v[i]);
                v[0] = v[1] + 1.0;
            }
        } else {
            v[0] += p[1] * 1e-9 + r * 1e-9;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += v[i];
    if (acc == 1.2345) out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc; CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 64));
    auto run = [&](const char* name, auto kern) {
        unsigned long long h = 0;
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, out, cyc, 1.0 + r);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-62s %7.1f cycles per step (s_memtime / 100 MHz ticks x 24 if the counter is the 100 MHz one: raw %llu)\n", name, (double)h / ITERS, h);
    };
    run("nothing but the loop", k_step<0>);
    run("wave max (6 DPP + readlane 63)", k_step<1>);
    run("16 v_readlane with a rotating lane", k_step<2>);
    run("wave max + 16 readlanes from the winner", k_step<3>);
    run("reciprocal chain (rcp + 4 fma)", k_step<4>);
    run("elimination (mul + 7 fma, exec-masked)", k_step<8>);
    run("wave max + readlanes + reciprocal", k_step<7>);
    run("the whole step", k_step<15>);
    return 0;
}
