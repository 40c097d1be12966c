// Host-logic check of the broadcast preparation (runmat_amd/csrc/host_shape.h): padded strides + dimension collapsing
// must address, for every output element, exactly the operand element MATLAB broadcasting defines (front-padded
// shapes, extent-1 dims repeat; crates/runmat-runtime/src/builtins/common/broadcast.rs:8-140).  Brute force, no GPU.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "host_shape.h"

using namespace rmhip;
typedef std::vector<size_t> Shape;

static int failures = 0;

// definition: column-major linear index of the operand element feeding output coordinate `c`
static uint64_t defined_source(const Shape& in, const Shape& out, const std::vector<size_t>& c) {
    const size_t pad = out.size() - in.size();
    uint64_t idx = 0, mul = 1;
    for (size_t d = 0; d < in.size(); ++d) {
        const size_t coord = in[d] == 1 ? 0 : c[d + pad];
        idx += coord * mul;
        mul *= in[d];
    }
    return idx;
}

static void check_case(const std::vector<Shape>& ins, const Shape& out, bool expect_ok) {
    const size_t rank = out.size();
    std::vector<std::vector<uint64_t>> strides(ins.size());
    bool ok = true;
    for (size_t k = 0; k < ins.size(); ++k) ok = ok && padded_strides(ins[k], out.data(), rank, &strides[k]);
    if (ok != expect_ok) {
        std::fprintf(stderr, "FAIL: broadcast acceptance mismatch (got %d, want %d)\n", (int)ok, (int)expect_ok);
        ++failures;
        return;
    }
    if (!ok) return;
    std::vector<uint64_t> cshape(out.begin(), out.end());
    std::vector<std::vector<uint64_t>> cstr = strides;
    collapse(&cshape, &cstr);
    uint64_t total = 1, ctotal = 1;
    for (size_t e : out) total *= e;
    for (uint64_t e : cshape) ctotal *= e;
    if (total != ctotal) {
        std::fprintf(stderr, "FAIL: collapsed element count %llu != %llu\n", (unsigned long long)ctotal, (unsigned long long)total);
        ++failures;
        return;
    }
    std::vector<size_t> c(rank, 0);
    for (uint64_t lin = 0; lin < total; ++lin) {
        // coordinates in the collapsed space
        uint64_t rem = lin;
        std::vector<uint64_t> cc(cshape.size());
        for (size_t d = 0; d < cshape.size(); ++d) {
            cc[d] = rem % cshape[d];
            rem /= cshape[d];
        }
        for (size_t k = 0; k < ins.size(); ++k) {
            // operands of higher rank than the request are squeezed the way padded_strides does it
            Shape in = ins[k];
            while (in.size() > rank && !in.empty() && in.back() == 1) in.pop_back();
            while (in.size() > rank && !in.empty() && in.front() == 1) in.erase(in.begin());
            uint64_t got = 0;
            for (size_t d = 0; d < cshape.size(); ++d) got += cc[d] * cstr[k][d];
            const uint64_t want = defined_source(in, out, c);
            if (got != want) {
                std::fprintf(stderr, "FAIL: operand %zu, output element %llu: source %llu, expected %llu\n", k,
                             (unsigned long long)lin, (unsigned long long)got, (unsigned long long)want);
                ++failures;
                return;
            }
        }
        for (size_t d = 0; d < rank; ++d) {  // next output coordinate, dim 0 fastest
            if (++c[d] < out[d]) break;
            c[d] = 0;
        }
    }
}

int main() {
    check_case({{4, 1}, {1, 3}}, {4, 3}, true);                 // graph.rs:252-271
    check_case({{2, 3}, {2, 1}}, {2, 3}, true);
    check_case({{5, 7}, {5, 7}, {1, 1}}, {5, 7}, true);         // same-shape + scalar: collapses to one dim
    check_case({{6, 5, 4}, {6, 1, 4}, {1, 5, 1}}, {6, 5, 4}, true);
    check_case({{3}, {2, 3}}, {2, 3}, true);                    // front-padding: [3] is [1, 3]
    check_case({{7, 1}, {7}}, {7}, true);                       // trailing singleton of a higher-rank operand is squeezed
    check_case({{1, 1, 5}, {4, 5}}, {4, 5}, true);              // leading singletons too
    check_case({{2, 1, 3, 1, 2}, {1, 4, 1, 5, 1}}, {2, 4, 3, 5, 2}, true);
    check_case({{1, 1}, {1, 1}}, {1, 1}, true);
    check_case({{8, 1, 1}, {1, 1, 9}, {8, 6, 9}}, {8, 6, 9}, true);
    check_case({{3, 2}, {2, 3}}, {3, 2}, false);                // not broadcastable
    check_case({{2, 3, 4}}, {3, 4}, false);                     // genuine higher rank
    // pseudo-random sweep: every dim independently full / singleton per operand
    unsigned seed = 12345;
    auto next = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int it = 0; it < 400; ++it) {
        const size_t rank = 1 + next() % 5;
        Shape out(rank);
        for (size_t d = 0; d < rank; ++d) out[d] = 1 + next() % 5;
        std::vector<Shape> ins(1 + next() % 3);
        for (auto& in : ins) {
            const size_t r = 1 + next() % rank;  // may be shorter: front-padded
            in.assign(r, 1);
            for (size_t d = 0; d < r; ++d) in[d] = (next() & 1) ? out[d + rank - r] : 1;
        }
        check_case(ins, out, true);
    }
    if (failures) return 1;
    std::puts("broadcast prep ok");
    return 0;
}
