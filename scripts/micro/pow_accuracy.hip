// Developer tool (GPU box): error of rm_pow_tab (pow_tab.h, the gamma step of image_normalize) and of its predecessor rm_pow_pos
// (skel_common.h) against the host's 80-bit powl over random bases in (0, 4), 2^-40 .. 2^2 and around 1 and exponents
// {0.45, 1.8, 2.2, 2.4}: max and mean error in units of 2^-53 relative (rm_pow_pos: also against its bound 0.45 |g ln w| + 1.5), and the
// time of each over the same array.  Build: hipcc --offload-arch=gfx950 -O3 -I runmat_amd/csrc scripts/micro/pow_accuracy.hip -o scripts/micro/pow_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "skel_common.h"
#include "pow_tab.h"

template <bool TAB>
__global__ void __launch_bounds__(256) k_pow(const double* x, double g, double* y, size_t n) {
    __shared__ __attribute__((aligned(16))) double s_pow[rmhip::kPowLdsDoubles];
    rmhip::PowTables tb{};
    if (TAB) tb = rmhip::pow_stage_tables(s_pow, threadIdx.x, 256);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = TAB ? rmhip::rm_pow_tab(x[i], g, tb) : rm_pow_pos(x[i], g);
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 4000000;
    std::vector<double> x(n), y(n);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(s >> 11) * 0x1.0p-53;
        // a third uniform in (0, 4), a third log-uniform over 2^-40 .. 2^2, a third near 1
        x[i] = i % 3 == 0 ? 4.0 * u + 1e-300 : (i % 3 == 1 ? std::exp2(-40.0 + 42.0 * u) : 1.0 + (u - 0.5) * 0.125);
    }
    double *dx, *dy;
    hipMalloc((void**)&dx, n * 8);
    hipMalloc((void**)&dy, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int tab = 1; tab >= 0; --tab)
        for (double g : {0.45, 1.8, 2.2, 2.4}) {
            const dim3 grid((unsigned)((n + 255) / 256));
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (tab) hipLaunchKernelGGL(k_pow<true>, grid, dim3(256), 0, 0, dx, g, dy, n);
                else hipLaunchKernelGGL(k_pow<false>, grid, dim3(256), 0, 0, dx, g, dy, n);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
            double worst = 0.0, sum = 0.0, worst_ratio = 0.0;
            for (size_t i = 0; i < n; ++i) {
                const long double want = powl((long double)x[i], (long double)g);
                const double err = (double)(fabsl((long double)y[i] - want) / want) * 0x1.0p+53;
                const double bound = 0.45 * std::fabs(g * std::log(x[i])) + 1.5;
                worst = err > worst ? err : worst;
                worst_ratio = err / bound > worst_ratio ? err / bound : worst_ratio;
                sum += err;
            }
            printf("%s gamma %.2f: max error %.3f x 2^-53, mean %.4f, max error / rm_pow_pos bound %.3f, %.1f us for %zu elements\n",
                   tab ? "rm_pow_tab" : "rm_pow_pos", g, worst, sum / (double)n, worst_ratio, ms * 1e3, n);
        }
    return 0;
}
