// provider_kats.cpp -- the reference's provider KATs written against the C++ host-side mirror
// (include/rmhip_provider.hpp).  Mirrors crates/runmat-runtime-integration-tests/tests/gpu.rs:28-60
// (plus / times on [1 2 3 4], [5 6 7 8]) and mtimes.rs:688-706 (column-major matmul round trip), plus one KAT per
// widened entry point (transpose view, syrk, linsolve, mldivide, stochastic_evolution, comparisons).
// Build: g++ -std=c++17 -Iinclude examples/provider_kats.cpp -Lrunmat_amd/csrc -lrmhip -Wl,-rpath,$PWD/runmat_amd/csrc
// Exit code 0 = all KATs pass; 2 = no gfx950 device (the provider has no CPU fallback); 1 = mismatch.
#include <cmath>
#include <cstdio>
#include <string>

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
        auto near = [](const std::vector<double>& v, std::initializer_list<double> w, double tol) {
            if (v.size() != w.size()) return false;
            size_t i = 0;
            for (double x : w)
                if (std::fabs(v[i++] - x) > tol) return false;
            return true;
        };
        // transpose view consumed in place: [1 3; 2 4]' * [5 6; 7 8]   (lib.rs:2532, matmul on a view)
        ok = ok && near(p.download(p.matmul(p.transpose(a), b2)).data, {19, 43, 22, 50}, 1e-12);
        ok = ok && eq(p.download(p.transpose(a)).data, {1, 3, 2, 4});
        // syrk: accelerate/tests/syrk.rs (A' * A), here on [1 3; 2 4]
        ok = ok && near(p.download(p.syrk(a)).data, {5, 11, 11, 25}, 1e-12);
        // linsolve LT hint: linsolve.rs:1224-1247
        rmhip_linsolve_options_t lt{};
        lt.lower = 1;
        lt.need_rcond = 1;
        auto sol = p.linsolve(p.upload({3, -1, 4, 0, 2, 1, 0, 0, 5}, {3, 3}), p.upload({9, 1, 19}, {3, 1}), lt);
        ok = ok && near(p.download(sol.solution).data, {3, 2, 1}, 1e-12) && sol.reciprocal_condition == 2.0 / 5.0;
        // mldivide: mldivide.rs:662-680 ([1 2; 3 4] \ [5; 6] = [-4; 4.5])
        ok = ok && near(p.download(p.mldivide(p.upload({1, 3, 2, 4}, {2, 2}), p.upload({5, 6}, {2, 1}))).data, {-4, 4.5}, 1e-12);
        // stochastic_evolution with zero scale: accelerate/tests/stochastic_evolution.rs:17-47
        ok = ok && near(p.download(p.stochastic_evolution(p.upload({1, 2, 3}, {3, 1}), 0.05, 0.0, 4)).data,
                        {std::exp(0.2), 2 * std::exp(0.2), 3 * std::exp(0.2)}, 1e-9);
        // comparisons: 1.0 / 0.0 tensors
        ok = ok && eq(p.download(p.elem_lt(a, b)).data, {1, 1, 1, 1}) && eq(p.download(p.elem_eq(a, a)).data, {1, 1, 1, 1});
        // reduce_moments_nd (lib.rs:2770-2778): E[x] and E[x^2] of [1 3; 2 4] over the rows
        auto mom = p.reduce_moments_nd(a, {0});
        ok = ok && near(p.download(mom.first).data, {1.5, 3.5}, 1e-15) && near(p.download(mom.second).data, {2.5, 12.5}, 1e-15);
        {
            // the unfused implicit-expansion sequence of times.rs:501-543: repmat each operand, elem_mul, free the expansions
            auto col = p.upload({1, 2, 3, 4}, {4, 1});
            auto row = p.upload({10, 20, 30}, {1, 3});
            auto ce = p.repmat(col, {1, 3}), re = p.repmat(row, {4, 1});
            auto prod = p.elem_mul(ce, re);
            p.free(ce);
            p.free(re);
            ok = ok && prod.shape == std::vector<size_t>{4, 3} &&
                 eq(p.download(prod).data, {10, 20, 30, 40, 20, 40, 60, 80, 30, 60, 90, 120});
            // repmat.rs:1044-1060 repmat_gpu_roundtrip; permute.rs:784-792; linspace.rs:495-512; read_scalar / gather_linear
            ok = ok && eq(p.download(p.repmat(p.upload({1, 2}, {2, 1}), {2})).data, {1, 2, 1, 2, 1, 2, 1, 2});
            ok = ok && eq(p.download(p.permute(p.upload({1, 99, 3, 4}, {2, 2}), {1, 0})).data, {1, 3, 99, 4});
            ok = ok && near(p.download(p.linspace(0.0, 1.0, 5)).data, {0.0, 0.25, 0.5, 0.75, 1.0}, 1e-12);
            ok = ok && p.read_scalar(prod, 5) == 40.0 && eq(p.download(p.gather_linear(prod, {11, 0, 6}, {3, 1})).data, {120, 10, 60});
            ok = ok && eq(p.download(p.zeros_like(prod)).data, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
            const double nan = std::nan("");
            ok = ok && eq(p.download(p.map_nan_to_zero(p.upload({1, nan, 3}, {3, 1}))).data, {1, 0, 3}) &&
                 eq(p.download(p.not_nan_mask(p.upload({1, nan, 3}, {3, 1}))).data, {1, 0, 1});
        }
        {
            // ProviderPrecision::F32 (lib.rs:815-818): host views stay f64, storage is f32, results round once
            rmhip::HipProvider q(0, 32);
            ok = ok && std::string(q.precision()) == "F32";
            auto x = q.upload({0.1, 0.2, 0.3, 16777217.0}, {2, 2});
            ok = ok && eq(q.download(x).data, {(double)0.1f, (double)0.2f, (double)0.3f, 16777216.0});
            auto y = q.elem_add(x, x);  // f64 arithmetic on the f32 values, one rounding
            ok = ok && eq(q.download(y).data, {(double)(float)(2.0 * (double)0.1f), (double)(float)(2.0 * (double)0.2f),
                                               (double)(float)(2.0 * (double)0.3f), 33554432.0});
            auto ai = q.upload({1, 2, 3, 4}, {2, 2});
            auto bi = q.upload({5, 7, 6, 8}, {2, 2});
            ok = ok && eq(q.download(q.matmul(ai, bi)).data, {26, 38, 30, 44});  // f32 matrix cores: integers stay exact
        }
        std::printf(ok ? "provider KATs ok\n" : "provider KATs FAILED\n");
        return ok ? 0 : 1;
    } catch (const rmhip::ProviderError& e) {
        std::fprintf(stderr, "provider unavailable: %s\n", e.what());
        return e.code == RMHIP_ERR_NO_DEVICE ? 2 : 1;
    }
}
