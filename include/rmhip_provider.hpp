// rmhip_provider.hpp -- C++ host side above the C ABI (include/rmhip.h): `rmhip::HipProvider`
// mirrors the reference's `trait AccelProvider` (crates/runmat-accelerate-api/src/lib.rs:1386-3151)
// for the dense-array hot path with the same method names, argument meaning and error behaviour:
// every provider `Err` becomes a `rmhip::ProviderError` (callers fall back to the CPU builtin,
// mtimes.rs:212-216); every op returns a NEW handle; inputs are never mutated.
// Header-only; link with -lrmhip.  The Rust equivalent is shim/hip_provider.rs.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <vector>

#include "rmhip.h"

namespace rmhip {

struct ProviderError : std::runtime_error {
    int code;
    ProviderError(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

// lib.rs:260-264
struct GpuTensorHandle {
    std::vector<size_t> shape;
    uint32_t device_id = 0;
    uint64_t buffer_id = 0;
    size_t numel() const {
        size_t n = 1;
        for (size_t d : shape) n *= d;
        return n;
    }
};

// lib.rs:3362-3372: column-major f64 host data
struct HostTensorView {
    const double* data;
    const size_t* shape;
    size_t rank;
};
struct HostTensorOwned {  // lib.rs:3362-3366
    std::vector<double> data;
    std::vector<size_t> shape;
    bool complex_interleaved = false;  // `storage == GpuTensorStorage::ComplexInterleaved`: data holds (re, im) pairs
};

// lib.rs:865-890
struct ReductionFlavor {
    enum Kind { Sum = 0, Mean = 1, CustomScale = 2 } kind = Sum;
    double scale = 1.0;
    static ReductionFlavor sum() { return {Sum, 1.0}; }
    static ReductionFlavor mean() { return {Mean, 1.0}; }
    static ReductionFlavor custom(double s) { return {CustomScale, s}; }
};

// lib.rs:649-698
struct ProviderLuResult {
    GpuTensorHandle combined, lower, upper, perm_matrix, perm_vector;
};

class HipProvider {
public:
    // precision_bits: 64 or 32 (ProviderPrecision, lib.rs:815-818), fixed for the provider's lifetime
    explicit HipProvider(int device_ordinal = 0, int precision_bits = 64) : precision_bits_(precision_bits) {
        static uint32_t next_device_id = 1;  // next_device_id(), lib.rs:3279
        check(rmhip_init(device_ordinal, &ctx_));
        if (precision_bits != 64) check(rmhip_set_precision(ctx_, precision_bits));
        device_id_ = next_device_id++;
    }
    ~HipProvider() {
        if (ctx_) rmhip_shutdown(ctx_);
    }
    HipProvider(const HipProvider&) = delete;
    HipProvider& operator=(const HipProvider&) = delete;

    uint32_t device_id() const { return device_id_; }
    const char* precision() const { return precision_bits_ == 32 ? "F32" : "F64"; }  // ProviderPrecision, lib.rs:815-818
    rmhip_device_info_t device_info_struct() const {
        rmhip_device_info_t info;
        check(rmhip_device_info(ctx_, &info));
        return info;
    }
    rmhip_telemetry_t telemetry_snapshot() const {
        rmhip_telemetry_t t;
        check(rmhip_telemetry(ctx_, &t));
        return t;
    }
    std::string device_info() const {  // lib.rs:1390: the one-line description
        const rmhip_device_info_t i = device_info_struct();
        return std::string(i.name) + " (" + i.arch + ", " + i.backend + ")";
    }
    uint32_t default_reduction_workgroup_size() const { return device_info_struct().reduction_workgroup_size; }  // lib.rs:3048
    uint32_t two_pass_threshold() const { return device_info_struct().two_pass_threshold; }                      // lib.rs:3053
    std::pair<uint64_t, uint64_t> fused_cache_counters() const {  // (hits, misses), lib.rs:3014
        const rmhip_telemetry_t t = telemetry_snapshot();
        return {t.fusion_cache_hits, t.fusion_cache_misses};
    }
    void reset_telemetry() const { check(rmhip_reset_telemetry(ctx_)); }  // lib.rs:3046
    rmhip_lu_stats_t lu_stats() const {  // solve-path factorisations, pivot-growth refactorisations, time-outs
        rmhip_lu_stats_t st;
        check(rmhip_lu_stats(ctx_, &st));
        return st;
    }

    // ---- memory (lib.rs:1387-1389, 1468-1522) ----
    GpuTensorHandle upload(const HostTensorView& host) const {
        uint64_t id = 0;
        check(rmhip_upload(ctx_, host.data, host.shape, host.rank, &id));
        return make(id, std::vector<size_t>(host.shape, host.shape + host.rank));
    }
    GpuTensorHandle upload(const std::vector<double>& data, const std::vector<size_t>& shape) const {
        return upload(HostTensorView{data.data(), shape.data(), shape.size()});
    }
    HostTensorOwned download(const GpuTensorHandle& h) const {
        const bool cplx = is_complex(h);
        HostTensorOwned out{std::vector<double>(h.numel() * (cplx ? 2 : 1)), h.shape, cplx};
        check(rmhip_download(ctx_, own(h), out.data.data(), out.data.size()));
        return out;
    }
    bool is_complex(const GpuTensorHandle& h) const {  // `handle_storage`, lib.rs:588-594
        int r = 0;
        check(rmhip_storage(ctx_, own(h), &r));
        return r != 0;
    }
    void free(const GpuTensorHandle& h) const { check(rmhip_free(ctx_, own(h))); }
    GpuTensorHandle fill(const std::vector<size_t>& shape, double value) const {
        uint64_t id = 0;
        check(rmhip_fill(ctx_, value, shape.data(), shape.size(), &id));
        return make(id, shape);
    }
    GpuTensorHandle zeros(const std::vector<size_t>& shape) const { return fill(shape, 0.0); }
    GpuTensorHandle ones(const std::vector<size_t>& shape) const { return fill(shape, 1.0); }
    GpuTensorHandle fill_like(const GpuTensorHandle& prototype, double value) const {  // lib.rs:1524
        uint64_t id = 0;
        check(rmhip_fill_like(ctx_, own(prototype), value, &id));
        return make(id, prototype.shape);
    }
    GpuTensorHandle zeros_like(const GpuTensorHandle& prototype) const { return fill_like(prototype, 0.0); }  // lib.rs:1497
    GpuTensorHandle ones_like(const GpuTensorHandle& prototype) const { return fill_like(prototype, 1.0); }   // lib.rs:1547
    // lib.rs:2676-2684: the SAME buffer id with a new shape
    GpuTensorHandle reshape(const GpuTensorHandle& h, const std::vector<size_t>& shape) const {
        uint64_t id = 0;
        check(rmhip_reshape(ctx_, own(h), shape.data(), shape.size(), &id));
        return make(id, shape);
    }

    // ---- shape / indexing hooks (lib.rs:2689, 2579, 1463, 1423-1445, 1887) ----
    GpuTensorHandle repmat(const GpuTensorHandle& a, const std::vector<size_t>& reps) const {  // a view; elem_* read it in place
        uint64_t id = 0;
        check(rmhip_repmat(ctx_, own(a), reps.data(), reps.size(), &id));
        return with_shape(id);
    }
    GpuTensorHandle permute(const GpuTensorHandle& a, const std::vector<size_t>& order_zero_based) const {
        uint64_t id = 0;
        check(rmhip_permute(ctx_, own(a), order_zero_based.data(), order_zero_based.size(), &id));
        return with_shape(id);
    }
    double read_scalar(const GpuTensorHandle& h, size_t linear_index) const {
        double v = 0.0;
        check(rmhip_read_scalar(ctx_, own(h), linear_index, &v));
        return v;
    }
    GpuTensorHandle gather_linear(const GpuTensorHandle& source, const std::vector<uint32_t>& indices,
                                  const std::vector<size_t>& output_shape) const {
        uint64_t id = 0;
        check(rmhip_gather_linear(ctx_, own(source), indices.data(), indices.size(), output_shape.data(), output_shape.size(), &id));
        return make(id, output_shape);
    }
    void scatter_linear(const GpuTensorHandle& target, const std::vector<uint32_t>& indices, const GpuTensorHandle& values) const {
        check(rmhip_scatter_linear(ctx_, own(target), indices.data(), indices.size(), own(values)));
    }
    GpuTensorHandle eye(const std::vector<size_t>& shape) const {  // lib.rs:1552
        uint64_t id = 0;
        check(rmhip_eye(ctx_, shape.data(), shape.size(), &id));
        return with_shape(id);
    }
    GpuTensorHandle eye_like(const GpuTensorHandle& prototype) const { return eye(prototype.shape); }  // lib.rs:1557
    GpuTensorHandle flip(const GpuTensorHandle& a, const std::vector<size_t>& axes_zero_based) const {  // lib.rs:2586
        uint64_t id = 0;
        check(rmhip_flip(ctx_, own(a), axes_zero_based.data(), axes_zero_based.size(), &id));
        return make(id, a.shape);
    }
    GpuTensorHandle circshift(const GpuTensorHandle& a, const std::vector<long long>& shifts) const {  // lib.rs:2589
        uint64_t id = 0;
        check(rmhip_circshift(ctx_, own(a), shifts.data(), shifts.size(), &id));
        return make(id, a.shape);
    }
    GpuTensorHandle tril(const GpuTensorHandle& a, long long offset = 0) const {  // lib.rs:1635
        uint64_t id = 0;
        check(rmhip_tri(ctx_, own(a), 0, offset, &id));
        return make(id, a.shape);
    }
    GpuTensorHandle triu(const GpuTensorHandle& a, long long offset = 0) const {  // lib.rs:1644
        uint64_t id = 0;
        check(rmhip_tri(ctx_, own(a), 1, offset, &id));
        return make(id, a.shape);
    }
    GpuTensorHandle cat(size_t dim_one_based, const std::vector<GpuTensorHandle>& inputs) const {  // lib.rs:2686
        std::vector<uint64_t> ids = ids_of(inputs);
        uint64_t id = 0;
        check(rmhip_cat(ctx_, dim_one_based, ids.data(), ids.size(), &id));
        return with_shape(id);
    }
    GpuTensorHandle linspace(double start, double stop, size_t count) const {
        uint64_t id = 0;
        check(rmhip_linspace(ctx_, start, stop, count, &id));
        return make(id, {1, count});
    }

    // ---- fused kernels (lib.rs:2946-3008) ----
    GpuTensorHandle fused_elementwise(const std::string& shader, const std::vector<GpuTensorHandle>& inputs,
                                      const std::vector<size_t>& output_shape, size_t len) const {
        return fused_elementwise_multi(shader, inputs, output_shape, len, 1)[0];
    }
    std::vector<GpuTensorHandle> fused_elementwise_multi(const std::string& shader,
                                                         const std::vector<GpuTensorHandle>& inputs,
                                                         const std::vector<size_t>& output_shape, size_t len,
                                                         size_t num_outputs) const {
        std::vector<uint64_t> ids = ids_of(inputs), outs(num_outputs, 0);
        check(rmhip_fused_elementwise(ctx_, shader.c_str(), ids.data(), ids.size(), output_shape.data(),
                                      output_shape.size(), len, num_outputs, outs.data()));
        std::vector<GpuTensorHandle> r;
        for (uint64_t id : outs) r.push_back(make(id, output_shape));
        return r;
    }
    GpuTensorHandle fused_reduction(const std::string& shader, const std::vector<GpuTensorHandle>& inputs,
                                    const std::vector<size_t>& output_shape, size_t reduce_len, size_t num_slices,
                                    uint32_t workgroup_size, ReductionFlavor flavor) const {
        std::vector<uint64_t> ids = ids_of(inputs);
        uint64_t out = 0;
        check(rmhip_fused_reduction(ctx_, shader.c_str(), ids.data(), ids.size(), output_shape.data(), output_shape.size(),
                                    reduce_len, num_slices, workgroup_size, (int)flavor.kind, flavor.scale, &out));
        return make(out, output_shape);
    }

    // ---- per-op hooks (lib.rs:1890-1938, 2077-2355) ----
    GpuTensorHandle elem_add(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_ADD, a, b); }
    GpuTensorHandle elem_sub(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_SUB, a, b); }
    GpuTensorHandle elem_mul(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_MUL, a, b); }
    GpuTensorHandle elem_div(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_DIV, a, b); }
    GpuTensorHandle elem_pow(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_POW, a, b); }
    GpuTensorHandle elem_max(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_MAX, a, b); }
    GpuTensorHandle elem_min(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_MIN, a, b); }
    GpuTensorHandle elem_hypot(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_HYPOT, a, b); }
    GpuTensorHandle elem_atan2(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_ATAN2, a, b); }
    // comparisons / logicals (lib.rs:1939-2068): 1.0 / 0.0 tensors
    GpuTensorHandle elem_eq(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_EQ, a, b); }
    GpuTensorHandle elem_ne(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_NE, a, b); }
    GpuTensorHandle elem_lt(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_LT, a, b); }
    GpuTensorHandle elem_le(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_LE, a, b); }
    GpuTensorHandle elem_gt(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_GT, a, b); }
    GpuTensorHandle elem_ge(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_GE, a, b); }
    GpuTensorHandle logical_and(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_AND, a, b); }
    GpuTensorHandle logical_or(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_OR, a, b); }
    GpuTensorHandle logical_xor(const GpuTensorHandle& a, const GpuTensorHandle& b) const { return binary(RMHIP_XOR, a, b); }
    GpuTensorHandle logical_not(const GpuTensorHandle& a) const { return unary(RMHIP_NOT, a); }
    GpuTensorHandle binary(rmhip_binary_op op, const GpuTensorHandle& a, const GpuTensorHandle& b) const {
        uint64_t out = 0;
        check(rmhip_binary(ctx_, op, own(a), own(b), &out));
        return with_shape(out);
    }
    GpuTensorHandle unary(rmhip_unary_op op, const GpuTensorHandle& a) const {
        uint64_t out = 0;
        check(rmhip_unary(ctx_, op, own(a), &out));
        return make(out, a.shape);
    }
    GpuTensorHandle unary_sin(const GpuTensorHandle& a) const { return unary(RMHIP_SIN, a); }
    GpuTensorHandle unary_cos(const GpuTensorHandle& a) const { return unary(RMHIP_COS, a); }
    GpuTensorHandle unary_exp(const GpuTensorHandle& a) const { return unary(RMHIP_EXP, a); }
    GpuTensorHandle unary_log(const GpuTensorHandle& a) const { return unary(RMHIP_LOG, a); }
    GpuTensorHandle unary_sqrt(const GpuTensorHandle& a) const { return unary(RMHIP_SQRT, a); }
    GpuTensorHandle unary_abs(const GpuTensorHandle& a) const { return unary(RMHIP_ABS, a); }
    GpuTensorHandle unary_tanh(const GpuTensorHandle& a) const { return unary(RMHIP_TANH, a); }
    GpuTensorHandle unary_gamma(const GpuTensorHandle& a) const { return unary(RMHIP_GAMMA, a); }        // lib.rs:2089
    GpuTensorHandle unary_gammaln(const GpuTensorHandle& a) const { return unary(RMHIP_GAMMALN, a); }    // lib.rs:2095
    GpuTensorHandle unary_erfcinv(const GpuTensorHandle& a) const { return unary(RMHIP_ERFCINV, a); }    // lib.rs:2107
    GpuTensorHandle unary_factorial(const GpuTensorHandle& a) const { return unary(RMHIP_FACTORIAL, a); }  // lib.rs:2113
    GpuTensorHandle unary_nextpow2(const GpuTensorHandle& a) const { return unary(RMHIP_NEXTPOW2, a); }  // lib.rs:2319
#define RMHIP_UNARY_HOOK(name, code) \
    GpuTensorHandle name(const GpuTensorHandle& a) const { return unary(code, a); }
    RMHIP_UNARY_HOOK(unary_tan, RMHIP_TAN) RMHIP_UNARY_HOOK(unary_asin, RMHIP_ASIN) RMHIP_UNARY_HOOK(unary_acos, RMHIP_ACOS)
    RMHIP_UNARY_HOOK(unary_atan, RMHIP_ATAN) RMHIP_UNARY_HOOK(unary_sinh, RMHIP_SINH) RMHIP_UNARY_HOOK(unary_cosh, RMHIP_COSH)
    RMHIP_UNARY_HOOK(unary_asinh, RMHIP_ASINH) RMHIP_UNARY_HOOK(unary_acosh, RMHIP_ACOSH) RMHIP_UNARY_HOOK(unary_atanh, RMHIP_ATANH)
    RMHIP_UNARY_HOOK(unary_expm1, RMHIP_EXPM1) RMHIP_UNARY_HOOK(unary_log2, RMHIP_LOG2) RMHIP_UNARY_HOOK(unary_log10, RMHIP_LOG10)
    RMHIP_UNARY_HOOK(unary_log1p, RMHIP_LOG1P) RMHIP_UNARY_HOOK(unary_sign, RMHIP_SIGN) RMHIP_UNARY_HOOK(unary_floor, RMHIP_FLOOR)
    RMHIP_UNARY_HOOK(unary_ceil, RMHIP_CEIL) RMHIP_UNARY_HOOK(unary_round, RMHIP_ROUND) RMHIP_UNARY_HOOK(unary_fix, RMHIP_FIX)
    RMHIP_UNARY_HOOK(unary_pow2, RMHIP_EXP2) RMHIP_UNARY_HOOK(unary_heaviside, RMHIP_HEAVISIDE) RMHIP_UNARY_HOOK(unary_single, RMHIP_SINGLE)
    RMHIP_UNARY_HOOK(unary_double, RMHIP_DOUBLE) RMHIP_UNARY_HOOK(unary_erf, RMHIP_ERF) RMHIP_UNARY_HOOK(unary_sinc, RMHIP_SINC)
    RMHIP_UNARY_HOOK(logical_isnan, RMHIP_ISNAN) RMHIP_UNARY_HOOK(logical_isinf, RMHIP_ISINF) RMHIP_UNARY_HOOK(logical_isfinite, RMHIP_ISFINITE)
    RMHIP_UNARY_HOOK(map_nan_to_zero, RMHIP_NAN_TO_ZERO) RMHIP_UNARY_HOOK(not_nan_mask, RMHIP_NOT_NAN)  // lib.rs:2980-2988
#undef RMHIP_UNARY_HOOK
    GpuTensorHandle scalar(rmhip_scalar_op op, const GpuTensorHandle& a, double s) const {
        uint64_t out = 0;
        check(rmhip_scalar(ctx_, op, own(a), s, &out));
        return make(out, a.shape);
    }
    GpuTensorHandle scalar_add(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SADD, a, s); }
    GpuTensorHandle scalar_sub(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SSUB, a, s); }
    GpuTensorHandle scalar_mul(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SMUL, a, s); }
    GpuTensorHandle scalar_div(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SDIV, a, s); }
    GpuTensorHandle scalar_rsub(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SRSUB, a, s); }
    GpuTensorHandle scalar_rdiv(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SRDIV, a, s); }
    GpuTensorHandle scalar_max(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SMAX, a, s); }
    GpuTensorHandle scalar_min(const GpuTensorHandle& a, double s) const { return scalar(RMHIP_SMIN, a, s); }

    // ---- reductions (lib.rs:2709-2883) ----
    GpuTensorHandle reduce(rmhip_reduce_op op, const GpuTensorHandle& a, int dim, bool omitnan = false) const {
        uint64_t out = 0;
        check(rmhip_reduce(ctx_, op, own(a), dim, omitnan ? 1 : 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle reduce_sum(const GpuTensorHandle& a) const { return reduce(RMHIP_RSUM, a, -1); }
    GpuTensorHandle reduce_sum_dim(const GpuTensorHandle& a, size_t dim) const { return reduce(RMHIP_RSUM, a, (int)dim); }
    GpuTensorHandle reduce_mean(const GpuTensorHandle& a) const { return reduce(RMHIP_RMEAN, a, -1); }
    GpuTensorHandle reduce_mean_dim(const GpuTensorHandle& a, size_t dim) const { return reduce(RMHIP_RMEAN, a, (int)dim); }
    GpuTensorHandle reduce_mean_nd(const GpuTensorHandle& a, const std::vector<size_t>& dims_zero_based) const {
        uint64_t out = 0;
        check(rmhip_reduce_nd(ctx_, RMHIP_RMEAN, own(a), dims_zero_based.data(), dims_zero_based.size(), 0, &out));
        return with_shape(out);
    }
    // lib.rs:2770-2778 -> ProviderMoments2 { mean, ex2 } (lib.rs:1317-1320)
    std::pair<GpuTensorHandle, GpuTensorHandle> reduce_moments_nd(const GpuTensorHandle& a, const std::vector<size_t>& dims_zero_based) const {
        uint64_t mean = 0, ex2 = 0;
        check(rmhip_reduce_moments_nd(ctx_, own(a), dims_zero_based.data(), dims_zero_based.size(), &mean, &ex2));
        return {with_shape(mean), with_shape(ex2)};
    }
    GpuTensorHandle reduce_prod(const GpuTensorHandle& a) const { return reduce(RMHIP_RPROD, a, -1); }                        // lib.rs:2743
    GpuTensorHandle reduce_prod_dim(const GpuTensorHandle& a, size_t dim) const { return reduce(RMHIP_RPROD, a, (int)dim); }  // lib.rs:2749
    GpuTensorHandle dot(const GpuTensorHandle& a, const GpuTensorHandle& b, int dim = -1) const {  // lib.rs:2722; dim < 0: first non-singleton
        uint64_t id = 0;
        check(rmhip_dot(ctx_, own(a), own(b), dim, &id));
        return with_shape(id);
    }
    GpuTensorHandle reduce_min(const GpuTensorHandle& a) const { return reduce(RMHIP_RMIN, a, -1); }
    GpuTensorHandle reduce_max(const GpuTensorHandle& a) const { return reduce(RMHIP_RMAX, a, -1); }
    // lib.rs:2864-2883 -> ReduceDimResult { values, indices } (lib.rs:513-517); CPU semantics (first occurrence, first NaN wins)
    struct ReduceDimResult {
        GpuTensorHandle values, indices;
    };
    struct SortResult {  // lib.rs:1085-1088
        HostTensorOwned values, indices;
    };
    ReduceDimResult reduce_min_dim(const GpuTensorHandle& a, size_t dim) const { return minmax_dim(RMHIP_RMIN, a, dim); }
    ReduceDimResult reduce_max_dim(const GpuTensorHandle& a, size_t dim) const { return minmax_dim(RMHIP_RMAX, a, dim); }
    // lib.rs:2786-2802 (normalization: 0 sample / 1 population; nan_mode: 0 include / 1 omit)
    GpuTensorHandle reduce_std(const GpuTensorHandle& a, int normalization, int nan_mode) const { return reduce_std_dim_(a, -1, normalization, nan_mode); }
    GpuTensorHandle reduce_std_dim(const GpuTensorHandle& a, size_t dim, int normalization, int nan_mode) const {
        return reduce_std_dim_(a, (int)dim, normalization, nan_mode);
    }
    // lib.rs:2730-2742, 2803-2850
    GpuTensorHandle reduce_nnz(const GpuTensorHandle& a) const { return truth(RMHIP_TNNZ, a, -1, false); }
    GpuTensorHandle reduce_nnz_dim(const GpuTensorHandle& a, size_t dim) const { return truth(RMHIP_TNNZ, a, (int)dim, false); }
    GpuTensorHandle reduce_any(const GpuTensorHandle& a, bool omit_nan) const { return truth(RMHIP_TANY, a, -1, omit_nan); }
    GpuTensorHandle reduce_any_dim(const GpuTensorHandle& a, size_t dim, bool omit_nan) const { return truth(RMHIP_TANY, a, (int)dim, omit_nan); }
    GpuTensorHandle reduce_all(const GpuTensorHandle& a, bool omit_nan) const { return truth(RMHIP_TALL, a, -1, omit_nan); }
    GpuTensorHandle reduce_all_dim(const GpuTensorHandle& a, size_t dim, bool omit_nan) const { return truth(RMHIP_TALL, a, (int)dim, omit_nan); }
    // lib.rs:2884-2891, 2908-2915 (reverse: ProviderScanDirection::Reverse; nan_mode: ProviderNanMode)
    GpuTensorHandle cumsum_scan(const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const { return cumulative(0, a, dim, reverse, nan_mode); }
    GpuTensorHandle cumprod_scan(const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const { return cumulative(1, a, dim, reverse, nan_mode); }
    // lib.rs:2918-2935 (ProviderCumminResult / ProviderCummaxResult = {values, indices}; nan_mode: 0 include / 1 omit)
    ReduceDimResult cummin_scan(const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const { return cumextreme(0, a, dim, reverse, nan_mode); }
    ReduceDimResult cummax_scan(const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const { return cumextreme(1, a, dim, reverse, nan_mode); }
    ReduceDimResult cumextreme(int is_max, const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const {
        uint64_t v = 0, i = 0;
        check(rmhip_cumextreme(ctx_, is_max, own(a), (int)dim, reverse ? 1 : 0, nan_mode, &v, &i));
        return {with_shape(v), with_shape(i)};
    }
    // lib.rs:2596-2603 (the reference's output order; column_major = true for the column-major array of the differences)
    GpuTensorHandle diff_dim(const GpuTensorHandle& a, size_t order, size_t dim, bool column_major = false) const {
        uint64_t out = 0;
        check(rmhip_diff_dim(ctx_, own(a), order, (int)dim, column_major ? 1 : 0, &out));
        return with_shape(out);
    }
    // lib.rs:2358-2366: SortResult carries host tensors (lib.rs:1085-1088); descend = SortOrder::Descend, by_abs = SortComparison::Abs
    SortResult sort_dim(const GpuTensorHandle& a, size_t dim, bool descend, bool by_abs) const {
        uint64_t v = 0, i = 0;
        check(rmhip_sort_dim(ctx_, own(a), (int)dim, descend ? 1 : 0, by_abs ? 1 : 0, &v, &i));
        const GpuTensorHandle hv = with_shape(v), hi = with_shape(i);
        SortResult r{download(hv), download(hi)};
        free(hv);
        free(hi);
        return r;
    }
    // lib.rs:2367-2374: columns as (zero-based index, descend) pairs
    SortResult sort_rows(const GpuTensorHandle& a, const std::vector<std::pair<size_t, bool>>& columns, bool by_abs) const {
        std::vector<size_t> idx;
        std::vector<int> desc;
        for (const auto& cspec : columns) idx.push_back(cspec.first), desc.push_back(cspec.second ? 1 : 0);
        uint64_t v = 0, i = 0;
        check(rmhip_sort_rows(ctx_, own(a), idx.data(), desc.data(), columns.size(), by_abs ? 1 : 0, &v, &i));
        const GpuTensorHandle hv = with_shape(v), hi = with_shape(i);
        SortResult r{download(hv), download(hi)};
        free(hv);
        free(hi);
        return r;
    }
    // lib.rs:2937-2944 -> ProviderFindResult (lib.rs:623-628); limit < 0 = None; last = FindDirection::Last
    struct FindResult {
        GpuTensorHandle linear, rows, cols, values;
    };
    FindResult find(const GpuTensorHandle& a, long long limit_or_neg, bool last) const {
        uint64_t l = 0, r = 0, c = 0, v = 0;
        check(rmhip_find(ctx_, own(a), limit_or_neg, last ? 1 : 0, &l, &r, &c, &v));
        return {with_shape(l), with_shape(r), with_shape(c), with_shape(v)};
    }
    // lib.rs:2833-2845
    GpuTensorHandle reduce_median(const GpuTensorHandle& a) const { return median_(a, -1); }
    GpuTensorHandle reduce_median_dim(const GpuTensorHandle& a, size_t dim) const { return median_(a, (int)dim); }
    GpuTensorHandle median_(const GpuTensorHandle& a, int dim) const {
        uint64_t out = 0;
        check(rmhip_reduce_median(ctx_, own(a), dim, &out));
        return with_shape(out);
    }
    ReduceDimResult minmax_dim(int op, const GpuTensorHandle& a, size_t dim) const {
        uint64_t v = 0, i = 0;
        check(rmhip_reduce_minmax_dim(ctx_, op, own(a), (int)dim, 0, &v, &i));
        return {with_shape(v), with_shape(i)};
    }
    GpuTensorHandle reduce_std_dim_(const GpuTensorHandle& a, int dim, int normalization, int nan_mode) const {
        uint64_t out = 0;
        check(rmhip_reduce_std(ctx_, own(a), dim, normalization, nan_mode, &out));
        return with_shape(out);
    }
    GpuTensorHandle truth(int op, const GpuTensorHandle& a, int dim, bool omit_nan) const {
        uint64_t out = 0;
        check(rmhip_reduce_truth(ctx_, op, own(a), dim, omit_nan ? 1 : 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle cumulative(int op, const GpuTensorHandle& a, size_t dim, bool reverse, int nan_mode) const {
        uint64_t out = 0;
        check(rmhip_cumulative(ctx_, op, own(a), (int)dim, reverse ? 1 : 0, nan_mode, &out));
        return with_shape(out);
    }

    // ---- linear algebra (lib.rs:2375, 2477-2500) ----
    GpuTensorHandle matmul(const GpuTensorHandle& a, const GpuTensorHandle& b) const {
        uint64_t out = 0;
        check(rmhip_matmul(ctx_, own(a), own(b), &out));
        return with_shape(out);
    }
    GpuTensorHandle matmul_epilogue(const GpuTensorHandle& a, const GpuTensorHandle& b, const rmhip_matmul_epilogue_t& ep) const {  // lib.rs:2394
        uint64_t id = 0;
        check(rmhip_matmul_epilogue(ctx_, own(a), own(b), &ep, &id));
        return make(id, {a.shape[0], b.shape[1]});
    }
    GpuTensorHandle mldivide(const GpuTensorHandle& lhs, const GpuTensorHandle& rhs) const {
        uint64_t out = 0;
        check(rmhip_mldivide(ctx_, own(lhs), own(rhs), &out));
        return with_shape(out);
    }
    // lib.rs:2502-2508 -> ProviderCholResult { factor, info } (lib.rs:658-662); throws for a matrix the host path must judge
    struct CholResult {
        GpuTensorHandle factor;
        unsigned info;
    };
    CholResult chol(const GpuTensorHandle& a, bool lower) const {
        uint64_t out = 0;
        unsigned info = 0;
        check(rmhip_chol(ctx_, own(a), lower ? 1 : 0, &out, &info));
        return {with_shape(out), info};
    }
    // lib.rs:2430-2436 (ProviderInvOptions is empty)
    GpuTensorHandle inv(const GpuTensorHandle& matrix) const {
        uint64_t out = 0;
        check(rmhip_inv(ctx_, own(matrix), &out));
        return with_shape(out);
    }
    GpuTensorHandle mrdivide(const GpuTensorHandle& lhs, const GpuTensorHandle& rhs) const {
        uint64_t out = 0;
        check(rmhip_mrdivide(ctx_, own(lhs), own(rhs), &out));
        return with_shape(out);
    }
    // ProviderLinsolveOptions / ProviderLinsolveResult (lib.rs:679-697, 2422-2429)
    struct LinsolveResult {
        GpuTensorHandle solution;
        double reciprocal_condition;
    };
    LinsolveResult linsolve(const GpuTensorHandle& lhs, const GpuTensorHandle& rhs, const rmhip_linsolve_options_t& opts) const {
        uint64_t out = 0;
        double rcond = 0.0;
        check(rmhip_linsolve(ctx_, own(lhs), own(rhs), &opts, &out, &rcond));
        return {with_shape(out), rcond};
    }
    GpuTensorHandle matmul_power_step(const GpuTensorHandle& lhs, const GpuTensorHandle& rhs, double epsilon) const {  // lib.rs:2414
        uint64_t out = 0;
        check(rmhip_matmul_power_step(ctx_, own(lhs), own(rhs), epsilon, &out));
        return with_shape(out);
    }
    GpuTensorHandle image_normalize(const GpuTensorHandle& input, const rmhip_image_normalize_t& desc) const {  // lib.rs:2407
        uint64_t out = 0;
        check(rmhip_image_normalize(ctx_, own(input), &desc, &out));
        return make(out, input.shape);
    }
    GpuTensorHandle covariance(const GpuTensorHandle& matrix, bool biased) const {  // lib.rs:1857 (dense, unweighted)
        uint64_t out = 0;
        check(rmhip_covariance(ctx_, own(matrix), biased ? 1 : 0, &out));
        return with_shape(out);
    }
    std::pair<GpuTensorHandle, GpuTensorHandle> covariance_to_correlation(const GpuTensorHandle& m) const {  // lib.rs:1876: (correlation, sigma)
        uint64_t corr = 0, sig = 0;
        check(rmhip_covariance_to_correlation(ctx_, own(m), &corr, &sig));
        return {with_shape(corr), with_shape(sig)};
    }
    // lib.rs:2437-2470; tolerance == nullptr: the default rule; cond norm: 0 Two (served), 1 One, 2 Inf, 3 Fro
    GpuTensorHandle rank(const GpuTensorHandle& m, const double* tolerance = nullptr) const {
        uint64_t out = 0;
        check(rmhip_rank(ctx_, own(m), tolerance ? 1 : 0, tolerance ? *tolerance : 0.0, &out));
        return with_shape(out);
    }
    GpuTensorHandle cond(const GpuTensorHandle& m, int norm = 0) const {
        uint64_t out = 0;
        check(rmhip_cond(ctx_, own(m), norm, &out));
        return with_shape(out);
    }
    GpuTensorHandle rcond(const GpuTensorHandle& m) const {  // lib.rs:2471
        uint64_t out = 0;
        check(rmhip_rcond(ctx_, own(m), &out));
        return with_shape(out);
    }
    GpuTensorHandle pinv(const GpuTensorHandle& m, const double* tolerance = nullptr) const {
        uint64_t out = 0;
        check(rmhip_pinv(ctx_, own(m), tolerance ? 1 : 0, tolerance ? *tolerance : 0.0, &out));
        return with_shape(out);
    }
    GpuTensorHandle peaks(size_t n) const {  // lib.rs:1781
        uint64_t out = 0;
        check(rmhip_peaks(ctx_, n, 0, 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle peaks_xy(const GpuTensorHandle& x, const GpuTensorHandle& y) const {  // lib.rs:1787
        uint64_t out = 0;
        check(rmhip_peaks(ctx_, 0, own(x), own(y), &out));
        return with_shape(out);
    }
    GpuTensorHandle corrcoef(const GpuTensorHandle& matrix, bool biased, int rows_mode = 0) const {  // lib.rs:1867 (rows_mode: 0 All, 1 Complete, 2 Pairwise)
        uint64_t out = 0;
        check(rmhip_corrcoef(ctx_, own(matrix), biased ? 1 : 0, rows_mode, &out));
        return with_shape(out);
    }
    GpuTensorHandle diag_extract(const GpuTensorHandle& matrix, long long offset) const {  // lib.rs:1625
        uint64_t out = 0;
        check(rmhip_diag_extract(ctx_, own(matrix), offset, &out));
        return with_shape(out);
    }
    GpuTensorHandle syrk(const GpuTensorHandle& a) const {  // lib.rs:2383: A' * A
        uint64_t out = 0;
        check(rmhip_syrk(ctx_, own(a), &out));
        return with_shape(out);
    }
    GpuTensorHandle transpose(const GpuTensorHandle& a) const {
        uint64_t out = 0;
        check(rmhip_transpose(ctx_, own(a), &out));
        return with_shape(out);
    }
    ProviderLuResult lu(const GpuTensorHandle& a) const {
        uint64_t ids[5] = {0, 0, 0, 0, 0};
        check(rmhip_lu(ctx_, own(a), ids));
        return {with_shape(ids[0]), with_shape(ids[1]), with_shape(ids[2]), with_shape(ids[3]), with_shape(ids[4])};
    }

    // ---- RNG (lib.rs:1713-1728, 1759-1772) ----
    GpuTensorHandle stochastic_evolution(const GpuTensorHandle& state, double drift, double scale, uint32_t steps) const {
        uint64_t out = 0;
        check(rmhip_stochastic_evolution(ctx_, own(state), drift, scale, steps, &out));
        return make(out, state.shape);
    }
    void set_rng_state(uint64_t state) const { check(rmhip_set_rng_state(ctx_, state)); }
    // storage-less `random_normal` handles generated in registers by the consuming fused kernel (rmhip.h: rmhip_set_lazy_random)
    void set_lazy_random(bool enabled, size_t min_numel = 0) const { check(rmhip_set_lazy_random(ctx_, enabled ? 1 : 0, min_numel)); }
    GpuTensorHandle random_uniform(const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_uniform(ctx_, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    GpuTensorHandle random_normal(const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_normal(ctx_, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    // lib.rs:1567-1569, 3064-3112, 2325-2331, 2197-2240, 2055 (index_ops.hip)
    std::vector<GpuTensorHandle> ndgrid(const std::vector<GpuTensorHandle>& axes, const std::vector<size_t>& output_shape, size_t output_count) const {
        std::vector<uint64_t> ids, outs(output_count ? output_count : 1);
        for (const auto& a : axes) ids.push_back(own(a));
        check(rmhip_ndgrid(ctx_, ids.data(), ids.size(), output_shape.data(), output_shape.size(), output_count, outs.data()));
        std::vector<GpuTensorHandle> r;
        for (size_t i = 0; i < output_count; ++i) r.push_back(with_shape(outs[i]));
        return r;
    }
    GpuTensorHandle sub2ind(const std::vector<size_t>& dims, const std::vector<size_t>& strides, const std::vector<GpuTensorHandle>& inputs,
                            const std::vector<bool>& scalar_mask, size_t len, const std::vector<size_t>& output_shape) const {
        std::vector<uint64_t> ids;
        std::vector<unsigned char> mask;
        for (const auto& h : inputs) ids.push_back(own(h));
        for (bool m : scalar_mask) mask.push_back(m ? 1 : 0);
        if (dims.size() != strides.size() || dims.size() != ids.size() || dims.size() != mask.size()) throw std::runtime_error("sub2ind: expected one subscript per dimension");
        uint64_t out = 0;
        check(rmhip_sub2ind(ctx_, dims.data(), strides.data(), ids.data(), mask.data(), dims.size(), len, output_shape.data(), output_shape.size(), &out));
        return with_shape(out);
    }
    bool supports_ind2sub() const { return true; }
    std::vector<GpuTensorHandle> ind2sub(const std::vector<size_t>& dims, const std::vector<size_t>& strides, const GpuTensorHandle& indices, size_t total,
                                         size_t len, const std::vector<size_t>& output_shape) const {
        std::vector<uint64_t> outs(dims.size() ? dims.size() : 1);
        check(rmhip_ind2sub(ctx_, dims.data(), strides.data(), dims.size(), own(indices), total, len, output_shape.data(), output_shape.size(), outs.data()));
        std::vector<GpuTensorHandle> r;
        for (size_t i = 0; i < dims.size(); ++i) r.push_back(with_shape(outs[i]));
        return r;
    }
    GpuTensorHandle scatter_column(const GpuTensorHandle& m, size_t col, const GpuTensorHandle& v) const {
        uint64_t out = 0;
        check(rmhip_scatter_line(ctx_, own(m), 1, col, own(v), &out));
        return with_shape(out);
    }
    GpuTensorHandle scatter_row(const GpuTensorHandle& m, size_t row, const GpuTensorHandle& v) const {
        uint64_t out = 0;
        check(rmhip_scatter_line(ctx_, own(m), 0, row, own(v), &out));
        return with_shape(out);
    }
    GpuTensorHandle pow2_scale(const GpuTensorHandle& m, const GpuTensorHandle& e) const {
        uint64_t out = 0;
        check(rmhip_pow2_scale(ctx_, own(m), own(e), &out));
        return with_shape(out);
    }
    GpuTensorHandle round_digits(const GpuTensorHandle& a, int digits, bool significant) const {
        uint64_t out = 0;
        check(rmhip_round_digits(ctx_, own(a), digits, significant ? 1 : 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle real_part_(int part, const GpuTensorHandle& a) const {
        uint64_t out = 0;
        check(rmhip_real_part(ctx_, part, own(a), &out));
        return with_shape(out);
    }
    GpuTensorHandle unary_real(const GpuTensorHandle& a) const { return real_part_(0, a); }
    GpuTensorHandle unary_imag(const GpuTensorHandle& a) const { return real_part_(1, a); }
    GpuTensorHandle unary_conj(const GpuTensorHandle& a) const { return real_part_(2, a); }
    GpuTensorHandle unary_angle(const GpuTensorHandle& a) const { return real_part_(3, a); }
    bool logical_isreal(const GpuTensorHandle& a) const {
        int r = 0;
        check(rmhip_isreal(ctx_, own(a), &r));
        return r != 0;
    }
    // lib.rs:1600-1623, 2697-2708, 2604-2620, 3115-3124 (misc_ops.hip)
    GpuTensorHandle diag_from_vector(const GpuTensorHandle& v, long long offset) const {
        uint64_t out = 0;
        check(rmhip_diag_from_vector(ctx_, own(v), offset, -1, -1, &out));
        return with_shape(out);
    }
    GpuTensorHandle diag_from_vector_sized(const GpuTensorHandle& v, long long offset, size_t rows, size_t cols) const {
        uint64_t out = 0;
        check(rmhip_diag_from_vector(ctx_, own(v), offset, (long long)rows, (long long)cols, &out));
        return with_shape(out);
    }
    GpuTensorHandle kron(const GpuTensorHandle& a, const GpuTensorHandle& b) const {
        uint64_t out = 0;
        check(rmhip_kron(ctx_, own(a), own(b), &out));
        return with_shape(out);
    }
    // dim: ONE-based as the trait's Option<usize>; 0 = None (the first dimension of extent 3)
    GpuTensorHandle cross(const GpuTensorHandle& lhs, const GpuTensorHandle& rhs, size_t dim_one_based_or_0 = 0) const {
        uint64_t out = 0;
        check(rmhip_cross(ctx_, own(lhs), own(rhs), (int)dim_one_based_or_0, &out));
        return with_shape(out);
    }
    GpuTensorHandle gradient_dim(const GpuTensorHandle& a, size_t dim, double spacing) const {
        uint64_t out = 0;
        check(rmhip_gradient_dim(ctx_, own(a), (int)dim, spacing, 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle gradient_dim_with_coordinates(const GpuTensorHandle& a, size_t dim, const GpuTensorHandle& coordinates) const {
        uint64_t out = 0;
        check(rmhip_gradient_dim(ctx_, own(a), (int)dim, 1.0, own(coordinates), &out));
        return with_shape(out);
    }
    // lib.rs:2893-2908; spacing_kind as ProviderTrapezoidSpacing: 0 Unit, 1 Scalar, 2 ScalarHandle, 3 Vector, 4 Tensor
    GpuTensorHandle trapz_dim(const GpuTensorHandle& a, size_t dim, int spacing_kind = 0, double scalar = 1.0, const GpuTensorHandle* spacing = nullptr) const {
        uint64_t out = 0;
        check(rmhip_trapz_dim(ctx_, own(a), (int)dim, 0, spacing_kind, scalar, spacing ? own(*spacing) : 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle cumtrapz_dim(const GpuTensorHandle& a, size_t dim, int spacing_kind = 0, double scalar = 1.0, const GpuTensorHandle* spacing = nullptr) const {
        uint64_t out = 0;
        check(rmhip_trapz_dim(ctx_, own(a), (int)dim, 1, spacing_kind, scalar, spacing ? own(*spacing) : 0, &out));
        return with_shape(out);
    }
    // lib.rs:2451-2457; order as ProviderNormOrder: 1 One, 2 Two, 3 Inf, 4 NegInf, 5 Zero, 6 Fro, 7 Nuc, 8 P(p)
    GpuTensorHandle norm(const GpuTensorHandle& t, int order, double p = 2.0) const {
        uint64_t out = 0;
        check(rmhip_norm(ctx_, own(t), order, p, &out));
        return with_shape(out);
    }
    bool issymmetric(const GpuTensorHandle& m, bool skew, double tolerance) const {
        int r = 0;
        check(rmhip_issymmetric(ctx_, own(m), skew ? 1 : 0, tolerance, &r));
        return r != 0;
    }
    // lib.rs:2645-2651 (elements; rows: not served) -> host tensors [count, 1], [count, 1], [numel, 1]
    struct UniqueResult {
        HostTensorOwned values, ia, ic;
    };
    UniqueResult unique(const GpuTensorHandle& a, bool stable, bool last_occurrence) const {
        const size_t n = a.numel();
        UniqueResult r;
        r.values.data.resize(n), r.ia.data.resize(n), r.ic.data.resize(n);
        size_t count = 0;
        check(rmhip_unique(ctx_, own(a), stable ? 1 : 0, last_occurrence ? 1 : 0, &count, r.values.data.data(), r.ia.data.data(), r.ic.data.data()));
        r.values.data.resize(count), r.ia.data.resize(count);
        r.values.shape = {count, 1}, r.ia.shape = {count, 1}, r.ic.shape = {n, 1};
        return r;
    }
    struct UnionResult {  // lib.rs:1140-1146
        HostTensorOwned values, ia, ib;
    };
    UnionResult set_union(const GpuTensorHandle& a, const GpuTensorHandle& b, bool stable) const {  // `union` is a keyword here
        UnionResult r;
        r.values.data.resize(a.numel() + b.numel()), r.ia.data.resize(a.numel()), r.ib.data.resize(b.numel());
        size_t n = 0, na = 0, nb = 0;
        check(rmhip_union(ctx_, own(a), own(b), stable ? 1 : 0, &n, r.values.data.data(), &na, r.ia.data.data(), &nb, r.ib.data.data()));
        r.values.data.resize(n), r.ia.data.resize(na), r.ib.data.resize(nb);
        r.values.shape = {n, 1}, r.ia.shape = {na, 1}, r.ib.shape = {nb, 1};
        return r;
    }
    struct SetdiffResult {  // lib.rs:1249-1254
        HostTensorOwned values, ia;
    };
    SetdiffResult setdiff(const GpuTensorHandle& a, const GpuTensorHandle& b, bool stable) const {
        SetdiffResult r;
        r.values.data.resize(a.numel()), r.ia.data.resize(a.numel());
        size_t n = 0;
        check(rmhip_setdiff(ctx_, own(a), own(b), stable ? 1 : 0, &n, r.values.data.data(), r.ia.data.data()));
        r.values.data.resize(n), r.ia.data.resize(n);
        r.values.shape = {n, 1}, r.ia.shape = {n, 1};
        return r;
    }
    struct IsMemberResult {  // lib.rs:1262-1274
        std::vector<unsigned char> mask;
        HostTensorOwned loc;
        std::vector<size_t> shape;
    };
    IsMemberResult ismember(const GpuTensorHandle& a, const GpuTensorHandle& b) const {
        IsMemberResult r;
        r.mask.resize(a.numel()), r.loc.data.resize(a.numel());
        r.shape = a.shape, r.loc.shape = a.shape;
        check(rmhip_ismember(ctx_, own(a), own(b), r.mask.data(), r.loc.data.data()));
        return r;
    }
    // lib.rs:1809-1817; padding 0 constant / 1 replicate / 2 symmetric / 3 circular; shape 0 same / 1 full / 2 valid
    GpuTensorHandle imfilter(const GpuTensorHandle& image, const GpuTensorHandle& kernel, int padding, double constant_value, int shape, bool convolution) const {
        uint64_t out = 0;
        check(rmhip_imfilter(ctx_, own(image), own(kernel), padding, constant_value, shape, convolution ? 1 : 0, &out));
        return with_shape(out);
    }
    // lib.rs:2458-2463; extrapolation: 0 NaN, 1 extrapolate, 2 the value
    GpuTensorHandle interp1(const GpuTensorHandle& x, const GpuTensorHandle& y, const GpuTensorHandle& xq, size_t sample_len, size_t series_count, size_t query_len,
                            const std::vector<size_t>& output_shape, bool nearest, int extrapolation, double value) const {
        uint64_t out = 0;
        check(rmhip_interp1(ctx_, own(x), own(y), own(xq), sample_len, series_count, query_len, output_shape.data(), output_shape.size(), nearest ? 1 : 0, extrapolation, value,
                            &out));
        return with_shape(out);
    }
    struct IirFilterResult {  // lib.rs:1308-1314
        GpuTensorHandle output, final_state;
    };
    IirFilterResult iir_filter(const GpuTensorHandle& b, const GpuTensorHandle& a, const GpuTensorHandle& x, size_t dim, const GpuTensorHandle* zi,
                               bool unit_denominator) const {  // lib.rs:2551-2559
        uint64_t out = 0, fin = 0;
        check(rmhip_iir_filter(ctx_, own(b), own(a), own(x), (int)dim, zi ? own(*zi) : 0, unit_denominator ? 1 : 0, &out, &fin));
        return {with_shape(out), with_shape(fin)};
    }
    // lib.rs:1652-1660; mu == nullptr: no centring / scaling, else {mean, scale}
    GpuTensorHandle polyval(const GpuTensorHandle& coefficients, const GpuTensorHandle& points, const double* mu = nullptr) const {
        uint64_t out = 0;
        check(rmhip_polyval(ctx_, own(coefficients), own(points), mu ? 1 : 0, mu ? mu[0] : 0.0, mu ? mu[1] : 1.0, &out));
        return with_shape(out);
    }
    // lib.rs:1674-1710: polynomial derivative (single / product rule / quotient rule: {numerator, denominator}) and integral
    GpuTensorHandle polyder_single(const GpuTensorHandle& polynomial) const {
        uint64_t out = 0;
        check(rmhip_polyder(ctx_, own(polynomial), 0, 0, &out, nullptr));
        return with_shape(out);
    }
    GpuTensorHandle polyder_product(const GpuTensorHandle& p, const GpuTensorHandle& q) const {
        uint64_t out = 0;
        check(rmhip_polyder(ctx_, own(p), own(q), 0, &out, nullptr));
        return with_shape(out);
    }
    std::pair<GpuTensorHandle, GpuTensorHandle> polyder_quotient(const GpuTensorHandle& u, const GpuTensorHandle& v) const {
        uint64_t num = 0, den = 0;
        check(rmhip_polyder(ctx_, own(u), own(v), 1, &num, &den));
        return {with_shape(num), with_shape(den)};
    }
    GpuTensorHandle polyint(const GpuTensorHandle& polynomial, double constant) const {
        uint64_t out = 0;
        check(rmhip_polyint(ctx_, own(polynomial), constant, &out));
        return with_shape(out);
    }
    // lib.rs:1561-1564: two or three host axes -> X, Y[, Z]
    std::vector<GpuTensorHandle> meshgrid(const std::vector<std::vector<double>>& axes) const {
        if (axes.size() != 2 && axes.size() != 3) throw ProviderError(RMHIP_ERR_INVALID, "meshgrid: provider expects two or three axes");
        uint64_t outs[3] = {0, 0, 0};
        check(rmhip_meshgrid(ctx_, axes[0].data(), axes[0].size(), axes[1].data(), axes[1].size(), axes.size() == 3 ? axes[2].data() : nullptr,
                             axes.size() == 3 ? axes[2].size() : 0, outs));
        std::vector<GpuTensorHandle> r;
        for (size_t i = 0; i < axes.size(); ++i) r.push_back(with_shape(outs[i]));
        return r;
    }
    // lib.rs:1472-1489 (complex == false: `zeros`)
    GpuTensorHandle zeros_with_storage(const std::vector<size_t>& shape, bool complex) const {
        if (!complex) return zeros(shape);
        uint64_t out = 0;
        check(rmhip_zeros_complex(ctx_, shape.data(), shape.size(), &out));
        return with_shape(out);
    }
    // lib.rs:2535-2550; mode: 0 full, 1 same, 2 valid (`ProviderConvMode`); column: `ProviderConvOrientation::Column`
    GpuTensorHandle conv1d(const GpuTensorHandle& signal, const GpuTensorHandle& kernel, int mode, bool column) const {
        uint64_t out = 0;
        check(rmhip_conv1d(ctx_, own(signal), own(kernel), mode, column ? 1 : 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle conv2d(const GpuTensorHandle& signal, const GpuTensorHandle& kernel, int mode) const {
        uint64_t out = 0;
        check(rmhip_conv2d(ctx_, own(signal), own(kernel), mode, &out));
        return with_shape(out);
    }
    // lib.rs:2852-2857; op 0 sum 1 mean 2 prod 3 min 4 max 5 median 6 std 7 var; endpoints 0 shrink 1 discard 2 fill(fill)
    GpuTensorHandle moving_window(const GpuTensorHandle& input, const std::vector<size_t>& output_shape, size_t dim, size_t before, size_t after, int op,
                                  int endpoints, double fill, bool nan_omit, bool population) const {
        uint64_t out = 0;
        check(rmhip_moving_window(ctx_, own(input), (int)dim, before, after, op, endpoints, fill, nan_omit ? 1 : 0, population ? 1 : 0, output_shape.data(),
                                  output_shape.size(), &out));
        return with_shape(out);
    }
    // lib.rs:1797-1807
    GpuTensorHandle hann_window(size_t len, bool periodic) const { return window(0, len, periodic); }
    GpuTensorHandle hamming_window(size_t len, bool periodic) const { return window(1, len, periodic); }
    GpuTensorHandle blackman_window(size_t len, bool periodic) const { return window(2, len, periodic); }
    GpuTensorHandle window(int kind, size_t len, bool periodic) const {
        uint64_t out = 0;
        check(rmhip_window(ctx_, kind, len, periodic ? 1 : 0, &out));
        return with_shape(out);
    }
    // lib.rs:2622-2644: transforms along zero-based `dim`, padded / truncated to `len` (-1: the extent) -> complex-interleaved tensors
    GpuTensorHandle fft_dim(const GpuTensorHandle& a, long long len, size_t dim) const {
        uint64_t out = 0;
        check(rmhip_fft_dim(ctx_, own(a), len, (int)dim, 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle ifft_dim(const GpuTensorHandle& a, long long len, size_t dim) const {
        uint64_t out = 0;
        check(rmhip_fft_dim(ctx_, own(a), len, (int)dim, 1, &out));
        return with_shape(out);
    }
    GpuTensorHandle signal_hilbert(const GpuTensorHandle& a, long long len, size_t dim) const {  // lib.rs:2572
        uint64_t out = 0;
        check(rmhip_hilbert(ctx_, own(a), len, (int)dim, &out));
        return with_shape(out);
    }
    GpuTensorHandle fft_extract_real(const GpuTensorHandle& a) const {
        uint64_t out = 0;
        check(rmhip_complex_real(ctx_, own(a), &out));
        return with_shape(out);
    }
    GpuTensorHandle complex_from_real(const GpuTensorHandle& re) const {  // lib.rs:1940-1947
        uint64_t out = 0;
        check(rmhip_complex(ctx_, own(re), 0, &out));
        return with_shape(out);
    }
    GpuTensorHandle complex_from_real_imag(const GpuTensorHandle& re, const GpuTensorHandle& im) const {  // lib.rs:1949-1959
        uint64_t out = 0;
        check(rmhip_complex(ctx_, own(re), own(im), &out));
        return with_shape(out);
    }
    bool ishermitian(const GpuTensorHandle& m, bool skew, double tolerance) const {  // lib.rs:3126-3138
        int r = 0;
        check(rmhip_ishermitian(ctx_, own(m), skew ? 1 : 0, tolerance, &r));
        return r != 0;
    }
    std::pair<uint32_t, uint32_t> bandwidth(const GpuTensorHandle& m) const {  // lib.rs:3140-3143: (lower, upper)
        unsigned lo = 0, up = 0;
        check(rmhip_bandwidth(ctx_, own(m), &lo, &up));
        return {lo, up};
    }
    // lib.rs:1718-1757, 1820-1839: the prototype forms and the scaled / transformed draws of the same stream
    GpuTensorHandle random_uniform_like(const GpuTensorHandle& prototype) const { return random_uniform(prototype.shape); }
    GpuTensorHandle random_normal_like(const GpuTensorHandle& prototype) const { return random_normal(prototype.shape); }
    GpuTensorHandle random_unifrnd(double a, double b, const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_unifrnd(ctx_, a, b, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    GpuTensorHandle random_exponential(double mu, const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_exponential(ctx_, mu, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    GpuTensorHandle random_normrnd(double mu, double sigma, const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_normrnd(ctx_, mu, sigma, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    GpuTensorHandle random_integer_range(long long lower, long long upper, const std::vector<size_t>& shape) const {
        uint64_t out = 0;
        check(rmhip_random_integer_range(ctx_, lower, upper, shape.data(), shape.size(), &out));
        return make(out, shape);
    }
    GpuTensorHandle random_integer_like(const GpuTensorHandle& prototype, long long lower, long long upper) const {
        return random_integer_range(lower, upper, prototype.shape);
    }

    // ---- multi-GPU collectives (include/rmhip.h "multi-GPU collectives"; one process per GPU, no trait counterpart) ----
    static std::array<unsigned char, RMHIP_COMM_ID_BYTES> comm_unique_id(bool rccl = true) {
        std::array<unsigned char, RMHIP_COMM_ID_BYTES> id{};
        check(rmhip_comm_unique_id(rccl ? RMHIP_COMM_RCCL : RMHIP_COMM_HOST_SHM, id.data()));
        return id;
    }
    void comm_init(const std::array<unsigned char, RMHIP_COMM_ID_BYTES>& id, int rank, int world) const {
        check(rmhip_comm_init(ctx_, id.data(), rank, world));
    }
    void comm_destroy() const { check(rmhip_comm_destroy(ctx_)); }
    std::pair<int, int> comm_rank() const {
        int r = 0, w = 1;
        check(rmhip_comm_rank(ctx_, &r, &w));
        return {r, w};
    }
    void comm_barrier() const { check(rmhip_comm_barrier(ctx_)); }
    void comm_bcast(const rmhip_view_t& block, int root, bool asynchronous = false) const {
        check(rmhip_comm_bcast(ctx_, &block, root, asynchronous ? 1 : 0));
    }
    void comm_wait() const { check(rmhip_comm_wait(ctx_)); }
    GpuTensorHandle comm_allgather_f64(const GpuTensorHandle& local) const {
        uint64_t out = 0;
        check(rmhip_comm_allgather_f64(ctx_, own(local), &out));
        return with_shape(out);
    }
    GpuTensorHandle comm_allgather_rows(const GpuTensorHandle& local, size_t rows_total, size_t granule = 128) const {
        uint64_t out = 0;
        check(rmhip_comm_allgather_rows(ctx_, own(local), rows_total, granule, &out));
        return with_shape(out);
    }

    rmhip_ctx* raw() const { return ctx_; }

private:
    static void check(int rc) {
        if (rc != RMHIP_OK) throw ProviderError(rc, rmhip_last_error());
    }
    GpuTensorHandle make(uint64_t id, std::vector<size_t> shape) const {
        GpuTensorHandle h;
        h.shape = std::move(shape);
        h.device_id = device_id_;
        h.buffer_id = id;
        return h;
    }
    GpuTensorHandle with_shape(uint64_t id) const {
        size_t rank = 16, shape[16];
        check(rmhip_shape(ctx_, id, &rank, shape));
        return make(id, std::vector<size_t>(shape, shape + rank));
    }
    uint64_t own(const GpuTensorHandle& h) const {
        if (h.device_id != device_id_)  // foreign handles are an error (io.rs:269-275)
            throw ProviderError(RMHIP_ERR_INVALID, "handle belongs to another device");
        return h.buffer_id;
    }
    std::vector<uint64_t> ids_of(const std::vector<GpuTensorHandle>& hs) const {
        std::vector<uint64_t> ids;
        for (const auto& h : hs) ids.push_back(own(h));
        return ids;
    }
    rmhip_ctx* ctx_ = nullptr;
    int precision_bits_ = 64;
    uint32_t device_id_ = 0;
};

}  // namespace rmhip
