// Host-logic check of the in-place repmat view addressing (runmat_amd/csrc/host_shape.h refined_strides + collapse): for
// every output element of a broadcasting launch whose operands may be lazy repmat views, the refined / collapsed strides
// must address exactly the base element the tiling rule defines - element (c0, c1, ...) of a view is base element
// (c0 % b0, c1 % b1, ...) (crates/runmat-accelerate/src/simple_provider.rs:2225-2238) - and a plain operand the element
// MATLAB broadcasting defines (common/broadcast.rs:8-140).  Brute force, no GPU.
#include <cstdio>
#include <cstdlib>

#include "host_shape.h"

using namespace rmhip;
typedef std::vector<size_t> Shape;

static int failures = 0, conflicts = 0, cases = 0;

static void check_case(const std::vector<OperandDims>& ops, const Shape& out) {
    const size_t rank = out.size();
    std::vector<uint64_t> rshape;
    std::vector<std::vector<uint64_t>> strides;
    size_t bad = 0;
    const int rc = refined_strides(ops, out.data(), rank, &rshape, &strides, &bad);
    if (rc == 2) {  // two views tile one dimension differently: the caller materialises one (checked separately below)
        ++conflicts;
        return;
    }
    if (rc != 0) {
        std::fprintf(stderr, "FAIL: refined_strides rejected operand %zu\n", bad);
        ++failures;
        return;
    }
    ++cases;
    collapse(&rshape, &strides);
    uint64_t total = 1, ctotal = 1;
    for (size_t e : out) total *= e;
    for (uint64_t e : rshape) ctotal *= e;
    if (total != ctotal) {
        std::fprintf(stderr, "FAIL: refined element count %llu != %llu\n", (unsigned long long)ctotal, (unsigned long long)total);
        ++failures;
        return;
    }
    std::vector<size_t> c(rank, 0);
    for (uint64_t lin = 0; lin < total; ++lin) {
        uint64_t rem = lin;
        std::vector<uint64_t> cc(rshape.size());
        for (size_t d = 0; d < rshape.size(); ++d) {
            cc[d] = rem % rshape[d];
            rem /= rshape[d];
        }
        for (size_t k = 0; k < ops.size(); ++k) {
            uint64_t got = 0;
            for (size_t d = 0; d < rshape.size(); ++d) got += cc[d] * strides[k][d];
            // definition: front-pad the operand to `rank`; a view reduces each coordinate modulo its base extent
            const Shape& shp = ops[k].shape;
            const Shape& base = ops[k].base.empty() ? ops[k].shape : ops[k].base;
            const size_t pad = rank - shp.size();
            uint64_t want = 0, mul = 1;
            for (size_t d = 0; d < shp.size(); ++d) {
                const size_t coord = shp[d] == 1 ? 0 : c[d + pad];
                want += (base[d] ? coord % base[d] : 0) * mul;
                mul *= base[d];
            }
            if (got != want) {
                std::fprintf(stderr, "FAIL: operand %zu, output element %llu: source %llu, expected %llu\n", k, (unsigned long long)lin,
                             (unsigned long long)got, (unsigned long long)want);
                ++failures;
                return;
            }
        }
        for (size_t d = 0; d < rank; ++d) {
            if (++c[d] < out[d]) break;
            c[d] = 0;
        }
    }
}

int main() {
    // the callers' cases (times.rs:501-543): 4x1 .* 1x3 - each operand expanded with repmat to 4x3, then elem_mul
    check_case({{{4, 3}, {4, 1}}, {{4, 3}, {1, 3}}}, {4, 3});
    check_case({{{2, 3}, {}}, {{2, 3}, {2, 1}}}, {2, 3});
    check_case({{{40, 30}, {1, 1}}, {{40, 30}, {}}}, {40, 30});           // gpuScalar .* gpuMatrix
    check_case({{{7, 6, 5}, {7, 1, 5}}, {{7, 6, 5}, {1, 6, 1}}}, {7, 6, 5});
    // true tilings: repmat([1 3; 2 4], 2, 3) against a dense 4x6, and against a broadcast row
    check_case({{{4, 6}, {2, 2}}, {{4, 6}, {}}}, {4, 6});
    check_case({{{4, 6}, {2, 2}}, {{1, 6}, {}}}, {4, 6});
    check_case({{{2, 3, 6}, {1, 3, 2}}, {{2, 3, 6}, {}}}, {2, 3, 6});     // repmat_high_dim_numeric (repmat.rs:811-845)
    check_case({{{6}, {2}}, {{6}, {3}}}, {6});                            // conflicting tilings -> rc 2
    check_case({{{6}, {2}}, {{6}, {2}}}, {6});                            // equal tilings share the split
    unsigned seed = 2024;
    auto next = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int it = 0; it < 1500; ++it) {
        const size_t rank = 1 + next() % 4;
        Shape out(rank);
        std::vector<OperandDims> ops(1 + next() % 3);
        // build the output as base * reps so that tilings exist
        Shape b0(rank), r0(rank);
        for (size_t d = 0; d < rank; ++d) {
            b0[d] = 1 + next() % 3;
            r0[d] = 1 + next() % 3;
            out[d] = b0[d] * r0[d];
        }
        for (auto& op : ops) {
            const size_t r = 1 + next() % rank;  // may be shorter: front-padded
            op.shape.assign(r, 1);
            const bool view = next() & 1;
            if (view) op.base.assign(r, 1);
            for (size_t d = 0; d < r; ++d) {
                const size_t od = d + rank - r;
                if (next() % 4 == 0) continue;  // extent 1: broadcast
                op.shape[d] = out[od];
                if (view) {
                    const unsigned pick = next() % 3;
                    op.base[d] = pick == 0 ? out[od] : pick == 1 ? 1 : b0[od];  // untiled / replicated element / tiled
                }
            }
        }
        check_case(ops, out);
    }
    if (failures) return 1;
    if (cases < 500 || conflicts < 1) {
        std::fprintf(stderr, "FAIL: sweep too thin (%d cases, %d conflicts)\n", cases, conflicts);
        return 1;
    }
    std::puts("repmat view ok");
    return 0;
}
