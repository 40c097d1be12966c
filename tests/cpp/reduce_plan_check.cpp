// Host-logic check of the reduction launch geometry (runmat_amd/csrc/reduce_plan.h): every shape must get a valid plan
// whose grid covers all slices and whose splits cover the reduced extent exactly once.  No GPU needed.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "reduce_plan.h"

using namespace rmhip;

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            std::fprintf(stderr, "FAIL %s: ", #cond);      \
            std::fprintf(stderr, __VA_ARGS__);             \
            std::fputc('\n', stderr);                      \
            ++failures;                                    \
        }                                                  \
    } while (0)

int main() {
    const uint64_t pres[] = {1, 2, 3, 7, 64, 255, 256, 257, 1000, 8192, 100000};
    const uint64_t reds[] = {1, 2, 5, 63, 64, 255, 2048, 4095, 4096, 8191, 8192, 12288, 16384, 100001, 67108864};
    const uint64_t posts[] = {1, 2, 3, 40, 8192, 65535, 65536, 70000};
    for (unsigned eb : {8u, 4u})
        for (uint64_t pre : pres)
            for (uint64_t red : reds)
                for (uint64_t post : posts) {
                    if ((double)pre * (double)red * (double)post > 1e12) continue;
                    const ReducePlan p = plan_reduction(pre, red, post, 256, eb);
                    if (!p.valid) {  // only geometries beyond the grid limits may be refused
                        CHECK(pre > 1 && post > 65535, "pre=%llu red=%llu post=%llu refused", (unsigned long long)pre,
                              (unsigned long long)red, (unsigned long long)post);
                        continue;
                    }
                    CHECK(p.nslices == pre * post, "nslices");
                    CHECK(p.nsplit >= 1 && p.nsplit <= 65535, "nsplit=%llu", (unsigned long long)p.nsplit);
                    CHECK(p.gx >= 1 && p.gy >= 1 && p.gz >= 1 && p.gy <= 65535 && p.gz <= 65535, "grid");
                    if (p.contiguous) {
                        CHECK(pre == 1, "kernel A needs pre == 1");
                        CHECK(p.tx == 256 || p.tx == 1024, "block size %d", p.tx);
                        CHECK((p.tx == 1024) == (red * eb >= 65536), "block size follows the slice's bytes");
                        CHECK(p.gx == p.nsplit && (uint64_t)p.gy * p.gz >= post, "kernel A grid covers the slices");
                        // block-aligned chunks cover [0, red) without overlap
                        uint64_t chunk = (red + p.nsplit - 1) / p.nsplit;
                        chunk = (chunk + p.tx - 1) / p.tx * p.tx;
                        CHECK(chunk * p.nsplit >= red, "chunks cover red");
                        CHECK(p.nsplit == 1 || chunk * (p.nsplit - 1) < red + chunk, "no empty interior split");
                    } else {
                        CHECK(p.tx >= 1 && p.tx <= 256 && (p.tx & (p.tx - 1)) == 0, "tx power of two");
                        CHECK((uint64_t)p.gx * p.tx >= pre, "kernel B covers pre");
                        CHECK(p.gy == p.nsplit && p.gz == post, "kernel B grid");
                        const uint64_t chunk = (red + p.nsplit - 1) / p.nsplit;
                        CHECK(chunk * p.nsplit >= red, "chunks cover red");
                    }
                }
    // sum(x, 'all') of the headline tensor spreads over the whole chip; one slice per block once there are enough slices
    const ReducePlan all = plan_reduction(1, 67108864ull, 1, 256);
    CHECK(all.contiguous && all.nsplit == 2048 && all.tx == 1024, "sum all: nsplit=%llu", (unsigned long long)all.nsplit);
    const ReducePlan cols = plan_reduction(1, 8192, 8192, 256);
    CHECK(cols.nsplit == 1 && cols.tx == 1024, "sum over columns, f64");
    const ReducePlan cols32 = plan_reduction(1, 8192, 8192, 256, 4);
    CHECK(cols32.nsplit == 1 && cols32.tx == 256, "sum over columns, f32 storage");
    const ReducePlan rows = plan_reduction(8192, 8192, 1, 256);
    // 64 chunks of 128 columns would each span 8 MiB - a multiple of 256 KiB, all chunks on the same memory channels at any moment -
    // so the plan takes 65 chunks of 127 (dealias_nsplit)
    CHECK(!rows.contiguous && rows.tx == 256 && rows.gx == 32 && rows.nsplit == 65, "sum over rows: gx=%u nsplit=%llu", rows.gx,
          (unsigned long long)rows.nsplit);
    const ReducePlan rows_odd = plan_reduction(8192, 8000, 1, 256);  // 125 columns per chunk: left alone
    CHECK(rows_odd.nsplit == 64, "sum over rows, 8000 columns: nsplit=%llu", (unsigned long long)rows_odd.nsplit);
    CHECK(dealias_nsplit(8192, 64, 65536, 512) == 65 && dealias_nsplit(8192, 1, 65536, 512) == 1 && dealias_nsplit(8192, 64, 63000, 512) == 64,
          "dealias_nsplit");
    const ReducePlan none = plan_reduction(0, 5, 1, 256);
    CHECK(!none.valid, "empty output");
    if (failures) return 1;
    std::puts("reduce plan ok");
    return 0;
}
