// offload_calibrate.cpp -- native calibration sample for RunMat's auto-offload thresholds, in the file format the reference loads.
//
// `apply_auto_offload_calibration_from_file` (crates/runmat-accelerate/src/native_auto.rs:478-) reads
//   { "auto_offload_calibration": { "runs", "cpu_time_ms": {elementwise, reduction, matmul},
//                                   "units": {elementwise, reduction, matmul_flops},
//                                   "provider": {name, vendor, backend, device_id} } }
// (`CalibrationFile` / `CalibrationSample` / `CalibrationTimes` / `CalibrationUnits` / `CalibrationProviderInfo`, :330-390) and turns
// cpu_time / units into seconds per element / per flop of the CPU path (:425-467); `provider` must match the registered provider's
// `device_info_struct()` (:490-).  This tool produces that sample on the box it runs on: the CPU side is the oracle's restatement of
// the reference CPU path (oracle/oracle.c, one host core - the reference is single-threaded), the provider block comes from
// rmhip_device_info.  A second object, "rmhip_break_even" (ignored by the reference's parser: no deny_unknown_fields), lists the GPU
// side measured through the C ABI - no Python / ctypes in the timed loops - and the sizes from which the device wins with resident
// operands, i.e. what RUNMAT_ACCEL_THRESHOLD_* should be set to for this backend.
//
// TEST / BENCH INFRASTRUCTURE: it links the CPU oracle, which the product never does.
// Build (done by __graft_entry__.build()):
//   g++ -O2 -std=c++17 -Iinclude tests/tools/offload_calibrate.cpp -Lrunmat_amd/csrc -lrmhip -Loracle -loracle -Wl,-rpath,... -o tests/tools/offload_calibrate
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rmhip.h"

extern "C" {
int orc_unary(int op, const double* x, size_t n, double* out);
int orc_binary(int op, const double* a, const size_t* sa, size_t ra, const double* b, const size_t* sb, size_t rb, double* out, size_t* out_shape,
               size_t* out_rank);
int orc_sum(const double* x, const size_t* shape, size_t rank, const int* reduce_mask, int nan_mode, int mean, double* out);
int orc_matmul(const double* a, size_t arows, size_t acols, const double* b, size_t brows, size_t bcols, double* out);
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double cpu_ms(F f, int runs) {  // best of `runs` (the reference's suite reports one time per category)
    f();
    double best = 1e300;
    for (int r = 0; r < runs; ++r) {
        const double t0 = now_ms();
        f();
        const double t = now_ms() - t0;
        if (t < best) best = t;
    }
    return best;
}

#define CK(x)                                                                    \
    do {                                                                         \
        int rc_ = (x);                                                           \
        if (rc_) {                                                               \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rmhip_last_error()); \
            return rc_ == RMHIP_ERR_NO_DEVICE ? 2 : 1;                           \
        }                                                                        \
    } while (0)

template <class F>
static int gpu_us(rmhip_ctx* ctx, int reps, F call, double* out_us) {
    for (int i = 0; i < 5; ++i) {
        rmhip_buf o = 0;
        CK(call(&o));
        CK(rmhip_free(ctx, o));
    }
    CK(rmhip_synchronize(ctx));
    const double t0 = now_ms();
    for (int i = 0; i < reps; ++i) {
        rmhip_buf o = 0;
        CK(call(&o));
        CK(rmhip_free(ctx, o));
    }
    CK(rmhip_synchronize(ctx));
    *out_us = (now_ms() - t0) * 1e3 / reps;
    return 0;
}

static std::string json_escape(const char* s) {
    std::string o;
    for (; *s; ++s) {
        if (*s == '"' || *s == '\\') o += '\\';
        o += *s;
    }
    return o;
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? std::atoi(argv[1]) : 5;
    // ---- CPU side: the categories of the reference's calibration suite on its own sizes of the order of the defaults' regime
    const size_t n_elem = (size_t)1 << 22, n_red = (size_t)1 << 22, mm = 256;
    std::vector<double> x(n_elem), y(n_elem), z(n_elem);
    for (size_t i = 0; i < n_elem; ++i) {
        x[i] = -3.0 + 6.0 * (double)i / (double)n_elem;
        y[i] = 1.0 + (double)i / (double)n_elem;
    }
    const size_t shp[2] = {n_elem, 1};
    size_t oshape[8], orank = 0;
    const double t_elem = cpu_ms([&] { orc_binary(0 /* add */, x.data(), shp, 2, y.data(), shp, 2, z.data(), oshape, &orank); }, runs);
    const size_t rshp[2] = {n_red, 1};
    const int mask[2] = {1, 1};
    double sum_out = 0.0;
    const double t_red = cpu_ms([&] { orc_sum(x.data(), rshp, 2, mask, 0, 0, &sum_out); }, runs);
    std::vector<double> A(mm * mm), B(mm * mm), Cm(mm * mm);
    for (size_t i = 0; i < mm * mm; ++i) {
        A[i] = std::sin((double)i);
        B[i] = std::cos((double)i);
    }
    const double t_mm = cpu_ms([&] { orc_matmul(A.data(), mm, mm, B.data(), mm, mm, Cm.data()); }, runs);

    // ---- provider block + GPU side
    rmhip_ctx* ctx = nullptr;
    CK(rmhip_init(0, &ctx));
    rmhip_device_info_t info;
    CK(rmhip_device_info(ctx, &info));
    std::string sweeps;
    size_t be_unary = 0, be_binary = 0, be_red = 0, be_mm = 0;
    for (int e = 8; e <= 22; e += 2) {
        const size_t n = (size_t)1 << e;
        const size_t s2[2] = {n, 1};
        rmhip_buf ha = 0, hb = 0;
        CK(rmhip_upload(ctx, x.data(), s2, 2, &ha));
        CK(rmhip_upload(ctx, y.data(), s2, 2, &hb));
        double g_un, g_bin, g_red;
        const int reps = n < ((size_t)1 << 18) ? 400 : 60;
        if (gpu_us(ctx, reps, [&](rmhip_buf* o) { return rmhip_unary(ctx, RMHIP_SIN, ha, o); }, &g_un)) return 1;
        if (gpu_us(ctx, reps, [&](rmhip_buf* o) { return rmhip_binary(ctx, RMHIP_ADD, ha, hb, o); }, &g_bin)) return 1;
        if (gpu_us(ctx, reps, [&](rmhip_buf* o) { return rmhip_reduce(ctx, RMHIP_RSUM, ha, -1, 0, o); }, &g_red)) return 1;
        const double c_un = cpu_ms([&] { orc_unary(0 /* sin */, x.data(), n, z.data()); }, 3) * 1e3;
        const double c_bin = cpu_ms([&] { orc_binary(0, x.data(), s2, 2, y.data(), s2, 2, z.data(), oshape, &orank); }, 3) * 1e3;
        const double c_red = cpu_ms([&] { orc_sum(x.data(), s2, 2, mask, 0, 0, &sum_out); }, 3) * 1e3;
        if (!be_unary && g_un < c_un) be_unary = n;
        if (!be_binary && g_bin < c_bin) be_binary = n;
        if (!be_red && g_red < c_red) be_red = n;
        char line[512];
        std::snprintf(line, sizeof line,
                      "%s{\"n\": %zu, \"unary_sin\": {\"gpu_us\": %.2f, \"cpu_us\": %.2f}, \"binary_add\": {\"gpu_us\": %.2f, \"cpu_us\": %.2f}, "
                      "\"reduce_sum\": {\"gpu_us\": %.2f, \"cpu_us\": %.2f}}",
                      sweeps.empty() ? "" : ", ", n, g_un, c_un, g_bin, c_bin, g_red, c_red);
        sweeps += line;
        CK(rmhip_free(ctx, ha));
        CK(rmhip_free(ctx, hb));
    }
    std::string mms;
    for (size_t n : {8, 16, 32, 48, 64, 96, 128, 192, 256}) {
        const size_t s2[2] = {n, n};
        rmhip_buf ha = 0, hb = 0;
        CK(rmhip_upload(ctx, A.data(), s2, 2, &ha));
        CK(rmhip_upload(ctx, B.data(), s2, 2, &hb));
        double g;
        if (gpu_us(ctx, 200, [&](rmhip_buf* o) { return rmhip_matmul(ctx, ha, hb, o); }, &g)) return 1;
        const double cpu = cpu_ms([&] { orc_matmul(A.data(), n, n, B.data(), n, n, Cm.data()); }, 3) * 1e3;
        if (!be_mm && g < cpu) be_mm = n * n * n;
        char line[256];
        std::snprintf(line, sizeof line, "%s{\"n\": %zu, \"flops\": %zu, \"gpu_us\": %.2f, \"cpu_us\": %.2f}", mms.empty() ? "" : ", ", n, n * n * n, g, cpu);
        mms += line;
        CK(rmhip_free(ctx, ha));
        CK(rmhip_free(ctx, hb));
    }
    std::printf(
        "{\"auto_offload_calibration\": {\"runs\": %d, \"cpu_time_ms\": {\"elementwise\": %.6f, \"reduction\": %.6f, \"matmul\": %.6f}, "
        "\"units\": {\"elementwise\": %zu, \"reduction\": %zu, \"matmul_flops\": %.1f}, "
        "\"provider\": {\"name\": \"%s\", \"vendor\": \"%s\", \"backend\": \"%s\", \"device_id\": %d}}, "
        "\"rmhip_break_even\": {\"note\": \"operands resident on the device; GPU side through the C ABI from C++ (no ctypes), CPU side = oracle/oracle.c "
        "on one host core; ignored by RunMat's CalibrationFile parser\", \"elementwise_sweep\": [%s], \"matmul_sweep\": [%s], "
        "\"recommended_env\": {\"RUNMAT_ACCEL_THRESHOLD_UNARY\": %zu, \"RUNMAT_ACCEL_THRESHOLD_ELEMWISE\": %zu, \"RUNMAT_ACCEL_THRESHOLD_REDUCTION\": %zu, "
        "\"RUNMAT_ACCEL_THRESHOLD_MATMUL\": %zu}}}\n",
        runs, t_elem, t_red, t_mm, n_elem, n_red, 2.0 * (double)mm * mm * mm, json_escape(info.name).c_str(), json_escape(info.vendor).c_str(),
        json_escape(info.backend).c_str(), info.device_ordinal, sweeps.c_str(), mms.c_str(), be_unary, be_binary, be_red, be_mm);
    rmhip_shutdown(ctx);
    return 0;
}
