// call_overhead.cpp -- host-side cost of one provider call, measured natively through the C ABI (no Python / ctypes in
// the loop).  The reference's fusion executor issues one `fused_elementwise` per fusion group (crates/runmat-accelerate/src/
// fusion_exec.rs:366-371) and frees the temporaries it uploaded (:415-419); for small tensors this per-call cost, not the
// kernel, is what a script sees (BASELINE.json configs[0] is the 1024x1024 elementwise-math chain).
//   enqueue   = wall time of the call loop divided by the calls (what the calling thread pays per call)
//   drained   = the same including the final synchronize (what the stream needs per call)
// Build: g++ -O2 -std=c++17 -Iinclude examples/call_overhead.cpp -Lrunmat_amd/csrc -lrmhip -Wl,-rpath,$PWD/runmat_amd/csrc
// Usage: call_overhead <requests dir> [n]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rmhip.h"

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static std::string slurp(const std::string& path) {
    std::ifstream f(path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
#define CK(x)                                                                          \
    do {                                                                               \
        int rc_ = (x);                                                                 \
        if (rc_) {                                                                     \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rmhip_last_error());       \
            return rc_ == RMHIP_ERR_NO_DEVICE ? 2 : 1;                                 \
        }                                                                              \
    } while (0)

template <class F>
static int rate(rmhip_ctx* ctx, const char* tag, int reps, F call) {
    for (int i = 0; i < 50; ++i) {
        rmhip_buf o = 0;
        CK(call(&o));
        CK(rmhip_free(ctx, o));
    }
    CK(rmhip_synchronize(ctx));
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) {
        rmhip_buf o = 0;
        CK(call(&o));
        CK(rmhip_free(ctx, o));
    }
    const double t1 = now_us();
    CK(rmhip_synchronize(ctx));
    const double t2 = now_us();
    std::printf("{\"call\": \"%s\", \"reps\": %d, \"enqueue_us\": %.2f, \"drained_us\": %.2f}\n", tag, reps, (t1 - t0) / reps, (t2 - t0) / reps);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "examples/requests";
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5000;
    const std::string sma = slurp(dir + "/sin_mul_add_f64.wgsl"), chain = slurp(dir + "/elementwise_math_f64.wgsl");
    if (sma.empty() || chain.empty()) {
        std::fprintf(stderr, "request files not found under %s\n", dir.c_str());
        return 1;
    }
    rmhip_ctx* ctx = nullptr;
    CK(rmhip_init(0, &ctx));
    for (size_t n : {(size_t)8, (size_t)1024}) {
        const size_t shape[2] = {n, n}, one[2] = {1, 1};
        rmhip_buf a, b, c, consts[5];
        CK(rmhip_fill_uniform(ctx, 1, -3.0, 3.0, shape, 2, &a));
        CK(rmhip_fill_uniform(ctx, 2, -1.0, 1.0, shape, 2, &b));
        CK(rmhip_fill_uniform(ctx, 3, -1.0, 1.0, shape, 2, &c));
        const double cv[5] = {10.0, 4.0, 0.25, 2.0, 0.1};
        for (int i = 0; i < 5; ++i) CK(rmhip_fill(ctx, cv[i], one, 2, &consts[i]));
        char tag[96];
        std::snprintf(tag, sizeof tag, "unary_sin %zux%zu", n, n);
        if (rate(ctx, tag, reps, [&](rmhip_buf* o) { return rmhip_unary(ctx, RMHIP_SIN, a, o); })) return 1;
        std::snprintf(tag, sizeof tag, "elem_add %zux%zu", n, n);
        if (rate(ctx, tag, reps, [&](rmhip_buf* o) { return rmhip_binary(ctx, RMHIP_ADD, a, b, o); })) return 1;
        std::snprintf(tag, sizeof tag, "fused sin(A).*B+C %zux%zu", n, n);
        const rmhip_buf in3[3] = {a, b, c};
        if (rate(ctx, tag, reps, [&](rmhip_buf* o) { return rmhip_fused_elementwise(ctx, sma.c_str(), in3, 3, shape, 2, n * n, 1, o); })) return 1;
        std::snprintf(tag, sizeof tag, "fused elementwise-math chain (14 ops) %zux%zu", n, n);
        const rmhip_buf in6[6] = {a, consts[0], consts[1], consts[2], consts[3], consts[4]};
        if (rate(ctx, tag, reps, [&](rmhip_buf* o) { return rmhip_fused_elementwise(ctx, chain.c_str(), in6, 6, shape, 2, n * n, 1, o); })) return 1;
        std::snprintf(tag, sizeof tag, "reduce_sum %zux%zu", n, n);
        if (rate(ctx, tag, reps, [&](rmhip_buf* o) { return rmhip_reduce(ctx, RMHIP_RSUM, a, -1, 0, o); })) return 1;
    }
    rmhip_shutdown(ctx);
    return 0;
}
