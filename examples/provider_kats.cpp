// provider_kats.cpp -- the reference's provider KATs written against the C++ host-side mirror
// (include/rmhip_provider.hpp).  Mirrors crates/runmat-runtime-integration-tests/tests/gpu.rs:28-60
// (plus / times on [1 2 3 4], [5 6 7 8]) and mtimes.rs:688-706 (column-major matmul round trip).
// Build: g++ -std=c++17 -Iinclude examples/provider_kats.cpp -Lrunmat_amd/csrc -lrmhip -Wl,-rpath,$PWD/runmat_amd/csrc
// Exit code 0 = all KATs pass; 2 = no gfx950 device (the provider has no CPU fallback); 1 = mismatch.
#include <cstdio>

#include "rmhip_provider.hpp"

static bool eq(const std::vector<double>& a, std::initializer_list<double> b) {
    return a == std::vector<double>(b);
}

int main() {
    try {
        rmhip::HipProvider p(0);
        auto a = p.upload({1, 2, 3, 4}, {2, 2});
        auto b = p.upload({5, 6, 7, 8}, {2, 2});
        bool ok = eq(p.download(p.elem_add(a, b)).data, {6, 8, 10, 12});
        ok = ok && eq(p.download(p.elem_mul(a, b)).data, {5, 12, 21, 32});
        auto b2 = p.upload({5, 7, 6, 8}, {2, 2});
        ok = ok && eq(p.download(p.matmul(a, b2)).data, {26, 38, 30, 44});
        auto s = p.reduce_sum(a);
        ok = ok && s.shape == std::vector<size_t>({1, 1}) && eq(p.download(s).data, {10});
        try {  // inner dimension mismatch is a soft error (mtimes.rs:628-637)
            p.matmul(a, p.upload({1, 2, 3}, {3, 1}));
            ok = false;
        } catch (const rmhip::ProviderError& e) {
            ok = ok && e.code == RMHIP_ERR_SHAPE;
        }
        std::printf(ok ? "provider KATs ok\n" : "provider KATs FAILED\n");
        return ok ? 0 : 1;
    } catch (const rmhip::ProviderError& e) {
        std::fprintf(stderr, "provider unavailable: %s\n", e.what());
        return e.code == RMHIP_ERR_NO_DEVICE ? 2 : 1;
    }
}
