// rmhip_auto_offload.hpp -- host-side mirror of RunMat's auto-offload decision, header-only C++17.
//
// In RunMat the CPU-versus-GPU choice is caller policy: `NativeAutoOffload` (crates/runmat-accelerate/src/native_auto.rs) decides per
// builtin call from element / flop thresholds, residency, the active fusion group, a small-batch guard and - when a GPU profile is
// present - a linear cost model, and it refines its CPU cost coefficients from a calibration file.  A maintainer who wires rmhip in
// keeps that code; this header restates the DECISION so that the backend's own tools and tests speak the same language: the calibrator
// (tests/tools/offload_calibrate.cpp) writes what `load_calibration_sample` below reads back, the break-even sizes it measures become
// thresholds through `apply_env_overrides`, and the KATs (examples/auto_offload_kats.cpp, run on the CPU by tests/test_auto_offload.py)
// pin every branch of the rules.  No GPU, no librmhip needed: the decision is pure host arithmetic.
//
//   Thresholds                 native_auto.rs:55-82   (defaults 4096 / 4096 / 256 elements, 1e6 flops; CPU costs :26-28; small batch :29-30)
//   apply_env_overrides        :1416-1449             (RUNMAT_ACCEL_THRESHOLD_{UNARY,ELEMWISE,REDUCTION,MATMUL,ALL}, RUNMAT_ACCEL_SMALL_BATCH_*)
//   Reason / reason_name       :105-114               (serialised kebab-case)
//   LinearModel, fit           :1937-1955, 2044-2075
//   ProfileCostModel, load     :1921-2042, 2081-2125  (reports by category; matmul samples in m*k*n; `RUNMAT_ACCEL_PROFILE` file)
//   CalibrationSample, load    :330-417               (`suite.auto_offload_calibration` wins over the top-level section)
//   apply_calibration_sample   :419-476               (ms / units -> seconds per element / flop; only real changes count)
//   Planner::evaluate_*        :923-1118, small_batch_guard :824-839, batch dimension :570-583
//   precision policy           crates/runmat-accelerate/src/precision.rs:22-81 (what may be promoted to an F64 / F32 provider)
#pragma once

#include <cmath>
#include <cstdint>
#include <cerrno>
#include <cstdlib>
#include <functional>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rmhip {
namespace auto_offload {

struct Thresholds {
    size_t unary_min_elems = 4096;
    size_t binary_min_elems = 4096;
    size_t reduction_min_elems = 256;
    size_t matmul_min_flops = 1000000;  // roughly 100 x 100 x 100
    double cpu_elem_per_elem = 1.0e-7;
    double cpu_reduction_per_elem = 1.2e-7;
    double cpu_matmul_per_flop = 2.5e-11;
    size_t small_batch_max_dim = 8;
    size_t small_batch_min_elems = 1048576;
};

// `lookup(name)` returns the variable's text or nullptr (std::getenv by default).  Values that do not parse as an unsigned integer are
// ignored, like `env_usize`.  Returns whether anything was applied.
inline bool apply_env_overrides(Thresholds& t, const std::function<const char*(const char*)>& lookup = [](const char* k) { return (const char*)std::getenv(k); }) {
    auto get = [&](const char* key, size_t* out) {
        const char* v = lookup(key);
        if (!v || !*v) return false;
        // Rust's usize parser: ASCII digits with an optional leading '+', nothing else - no sign, no white space of any kind
        // (strtoull alone would skip leading tabs / newlines), and a value that does not fit is an error, not a saturated maximum
        const char* d = *v == '+' ? v + 1 : v;
        if (*d < '0' || *d > '9') return false;
        char* end = nullptr;
        errno = 0;
        const unsigned long long x = std::strtoull(d, &end, 10);
        if (end == d || *end != '\0' || errno == ERANGE) return false;
        *out = (size_t)x;
        return true;
    };
    bool applied = false;
    size_t v = 0;
    if (get("RUNMAT_ACCEL_THRESHOLD_UNARY", &v)) t.unary_min_elems = v, applied = true;
    if (get("RUNMAT_ACCEL_THRESHOLD_ELEMWISE", &v)) t.binary_min_elems = v, applied = true;
    if (get("RUNMAT_ACCEL_THRESHOLD_REDUCTION", &v)) t.reduction_min_elems = v, applied = true;
    if (get("RUNMAT_ACCEL_THRESHOLD_MATMUL", &v)) t.matmul_min_flops = v, applied = true;
    if (get("RUNMAT_ACCEL_THRESHOLD_ALL", &v)) {  // after the specific ones: it wins over them, and leaves the matmul threshold alone
        t.unary_min_elems = t.binary_min_elems = t.reduction_min_elems = v;
        applied = true;
    }
    if (get("RUNMAT_ACCEL_SMALL_BATCH_MAX_DIM", &v)) t.small_batch_max_dim = v, applied = true;
    if (get("RUNMAT_ACCEL_SMALL_BATCH_MIN_ELEMS", &v)) t.small_batch_min_elems = v, applied = true;
    return applied;
}

enum class Reason { FusionOverride, Residency, SmallBatchGuard, ProfileModel, Threshold, Disabled };
inline const char* reason_name(Reason r) {
    switch (r) {
        case Reason::FusionOverride: return "fusion-override";
        case Reason::Residency: return "residency";
        case Reason::SmallBatchGuard: return "small-batch-guard";
        case Reason::ProfileModel: return "profile-model";
        case Reason::Threshold: return "threshold";
        default: return "disabled";
    }
}

struct Decision {
    bool gpu = false;
    Reason reason = Reason::Threshold;
    std::optional<double> cpu_secs, gpu_secs;
    std::optional<size_t> threshold, batch;
};

// ---- the GPU-side cost model fitted from profile reports ------------------------------------------------------------------------
struct LinearModel {
    double slope = 0.0, intercept = 0.0;
    std::optional<double> estimate(double x) const {  // seconds
        if (!std::isfinite(slope) || slope <= 0.0) return std::nullopt;
        const double total = intercept + slope * x;
        if (std::isfinite(total) && total > 0.0) return total;
        return std::nullopt;
    }
};

inline std::optional<LinearModel> fit_linear_model(const std::vector<std::pair<double, double>>& samples) {
    if (samples.empty()) return std::nullopt;
    if (samples.size() == 1) {  // one point: a line through the origin
        if (samples[0].first > 0.0) return LinearModel{std::max(samples[0].second / samples[0].first, 0.0), 0.0};
        return std::nullopt;
    }
    double sx = 0, sy = 0, sxx = 0, sxy = 0;
    for (const auto& p : samples) {
        sx += p.first;
        sy += p.second;
        sxx += p.first * p.first;
        sxy += p.first * p.second;
    }
    const double n = (double)samples.size(), denom = n * sxx - sx * sx;
    if (std::fabs(denom) < std::numeric_limits<double>::epsilon()) return std::nullopt;
    const double slope = (n * sxy - sx * sy) / denom;
    double intercept = sy / n - slope * (sx / n);
    if (intercept < 0.0) intercept = 0.0;  // a launch cost cannot be negative
    if (!std::isfinite(slope) || slope <= 0.0) return std::nullopt;
    return LinearModel{slope, intercept};
}

struct ProfileReport {  // one entry of a GPU profile file: category, the operand shapes, the average total time
    std::string category;
    std::vector<std::vector<size_t>> input_shapes;
    double avg_total_ms = 0.0;
};

struct ProfileCostModel {
    std::optional<LinearModel> elem, reduction, transpose, matmul;
    static ProfileCostModel from_reports(const std::vector<ProfileReport>& reports) {
        std::vector<std::pair<double, double>> e, r, t, m;
        for (const ProfileReport& rep : reports) {
            const double secs = rep.avg_total_ms / 1000.0;
            if (rep.category == "elementwise" || rep.category == "reduction" || rep.category == "transpose") {
                if (rep.input_shapes.empty()) continue;
                size_t elems = 1;
                for (size_t d : rep.input_shapes[0]) elems *= d;
                if (elems == 0) continue;
                (rep.category == "elementwise" ? e : rep.category == "reduction" ? r : t).emplace_back((double)elems, secs);
            } else if (rep.category == "matmul") {
                if (rep.input_shapes.size() < 2 || rep.input_shapes[0].size() != 2 || rep.input_shapes[1].size() != 2) continue;
                const size_t mm = rep.input_shapes[0][0], kk = rep.input_shapes[0][1], nn = rep.input_shapes[1][1];
                if (mm != 0 && kk > std::numeric_limits<size_t>::max() / mm) continue;  // the reference's checked_mul
                const size_t mk = mm * kk;
                if (mk != 0 && nn > std::numeric_limits<size_t>::max() / mk) continue;
                m.emplace_back((double)(mk * nn), secs);  // m * k * n (not doubled): the unit the matmul threshold is compared in
            }
        }
        return ProfileCostModel{fit_linear_model(e), fit_linear_model(r), fit_linear_model(t), fit_linear_model(m)};
    }
};

// ---- calibration file ----------------------------------------------------------------------------------------------------------
struct CalibrationProvider {
    std::string name, vendor;
    std::optional<std::string> backend;
    uint32_t device_id = 0;
};
struct CalibrationSample {
    size_t runs = 0;
    double cpu_ms_elementwise = 0.0, cpu_ms_reduction = 0.0, cpu_ms_matmul = 0.0;
    double units_elementwise = 0.0, units_reduction = 0.0, units_matmul_flops = 0.0;
    std::optional<CalibrationProvider> provider;
    bool provider_conflict = false;
};

namespace detail {
// A reader for the small JSON subset calibration files use (objects, arrays, strings without exotic escapes, numbers, true / false /
// null); unknown members are skipped - the reference's serde structs do not deny unknown fields either.
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* get(const std::string& key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};
struct JsonParser {
    const std::string& s;
    size_t i = 0;
    explicit JsonParser(const std::string& text) : s(text) {}
    [[noreturn]] void bad(const char* what) const { throw std::runtime_error(std::string("failed to parse calibration file: ") + what + " at offset " + std::to_string(i)); }
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
    }
    Json value() {
        ws();
        if (i >= s.size()) bad("unexpected end");
        Json j;
        const char c = s[i];
        if (c == '{') {
            j.kind = Json::Obj;
            ++i;
            ws();
            if (i < s.size() && s[i] == '}') return ++i, j;
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') bad("expected a member name");
                std::string key = string();
                ws();
                if (i >= s.size() || s[i] != ':') bad("expected ':'");
                ++i;
                j.obj.emplace_back(std::move(key), value());
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == '}') return ++i, j;
                bad("expected ',' or '}'");
            }
        }
        if (c == '[') {
            j.kind = Json::Arr;
            ++i;
            ws();
            if (i < s.size() && s[i] == ']') return ++i, j;
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ']') return ++i, j;
                bad("expected ',' or ']'");
            }
        }
        if (c == '"') {
            j.kind = Json::Str;
            j.str = string();
            return j;
        }
        if (s.compare(i, 4, "true") == 0) return i += 4, j.kind = Json::Bool, j.b = true, j;
        if (s.compare(i, 5, "false") == 0) return i += 5, j.kind = Json::Bool, j;
        if (s.compare(i, 4, "null") == 0) return i += 4, j;
        char* end = nullptr;
        j.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) bad("unexpected character");
        i = (size_t)(end - s.c_str());
        j.kind = Json::Num;
        return j;
    }
    std::string string() {
        std::string out;
        ++i;  // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                if (++i >= s.size()) bad("unterminated escape");
                const char e = s[i];
                if (e == 'n') out += '\n';
                else if (e == 't') out += '\t';
                else if (e == 'u') {  // names and vendors are ASCII in practice: keep the escape as it stands
                    out += "\\u";
                } else out += e;
                ++i;
            } else out += s[i++];
        }
        if (i >= s.size()) bad("unterminated string");
        ++i;
        return out;
    }
};
// a JSON number as a count: negative, NaN and out-of-range values clamp instead of hitting the undefined double -> integer conversion
inline size_t to_size(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709549568.0) return (size_t)-1;
    return (size_t)v;
}
inline double num_or(const Json* j, const char* key, double dflt) {
    const Json* v = j ? j->get(key) : nullptr;
    return v && v->kind == Json::Num ? v->num : dflt;
}
}  // namespace detail

// The sample of a calibration file's text: `suite.auto_offload_calibration` when present, else the top-level `auto_offload_calibration`;
// a file with neither is an error ("calibration file does not contain an auto_offload_calibration section").  Missing members default
// to zero, like the reference's `#[serde(default)]`.
inline CalibrationSample load_calibration_sample(const std::string& json_text) {
    detail::JsonParser p(json_text);
    const detail::Json root = p.value();
    const detail::Json* sec = nullptr;
    if (const detail::Json* suite = root.get("suite")) sec = suite->get("auto_offload_calibration");
    if (!sec || sec->kind != detail::Json::Obj) sec = root.get("auto_offload_calibration");
    if (!sec || sec->kind != detail::Json::Obj) throw std::runtime_error("calibration file does not contain an auto_offload_calibration section");
    CalibrationSample s;
    s.runs = detail::to_size(detail::num_or(sec, "runs", 0.0));
    const detail::Json* t = sec->get("cpu_time_ms");
    s.cpu_ms_elementwise = detail::num_or(t, "elementwise", 0.0);
    s.cpu_ms_reduction = detail::num_or(t, "reduction", 0.0);
    s.cpu_ms_matmul = detail::num_or(t, "matmul", 0.0);
    const detail::Json* u = sec->get("units");
    s.units_elementwise = detail::num_or(u, "elementwise", 0.0);
    s.units_reduction = detail::num_or(u, "reduction", 0.0);
    s.units_matmul_flops = detail::num_or(u, "matmul_flops", 0.0);
    if (const detail::Json* pr = sec->get("provider"); pr && pr->kind == detail::Json::Obj) {
        CalibrationProvider cp;
        if (const detail::Json* v = pr->get("name"); v && v->kind == detail::Json::Str) cp.name = v->str;
        if (const detail::Json* v = pr->get("vendor"); v && v->kind == detail::Json::Str) cp.vendor = v->str;
        if (const detail::Json* v = pr->get("backend"); v && v->kind == detail::Json::Str) cp.backend = v->str;
        cp.device_id = (uint32_t)std::min<size_t>(detail::to_size(detail::num_or(pr, "device_id", 0.0)), 0xffffffffu);
        s.provider = cp;
    }
    if (const detail::Json* v = sec->get("provider_conflict"); v && v->kind == detail::Json::Bool) s.provider_conflict = v->b;
    return s;
}

// A GPU profile file (`RUNMAT_ACCEL_PROFILE`, native_auto.rs:2081-2125): a JSON array of reports {category, input_shapes, total_ms{avg_ms}}
// (:1921-1935); reports without a category or a total are skipped.
inline std::vector<ProfileReport> load_profile_reports(const std::string& json_text) {
    detail::JsonParser p(json_text);
    const detail::Json root = p.value();
    if (root.kind != detail::Json::Arr) throw std::runtime_error("GPU profile: expected an array of reports");
    std::vector<ProfileReport> out;
    for (const detail::Json& r : root.arr) {
        const detail::Json* cat = r.get("category");
        const detail::Json* tot = r.get("total_ms");
        if (!cat || cat->kind != detail::Json::Str || !tot) continue;
        ProfileReport rep;
        rep.category = cat->str;
        rep.avg_total_ms = detail::num_or(tot, "avg_ms", 0.0);
        if (const detail::Json* shapes = r.get("input_shapes"); shapes && shapes->kind == detail::Json::Arr)
            for (const detail::Json& sh : shapes->arr) {
                std::vector<size_t> dims;
                if (sh.kind == detail::Json::Arr)
                    for (const detail::Json& d : sh.arr) dims.push_back(d.kind == detail::Json::Num && d.num >= 0.0 ? (size_t)d.num : 0);
                rep.input_shapes.push_back(std::move(dims));
            }
        out.push_back(std::move(rep));
    }
    return out;
}

struct CalibrationDelta {  // before / after of every coefficient the sample changed
    std::optional<std::pair<double, double>> cpu_elem_per_elem, cpu_reduction_per_elem, cpu_matmul_per_flop;
};

// cpu_time_ms / units -> seconds per element (per flop); a coefficient moves only when the sample has both numbers, the quotient is
// finite and positive, and it differs from the current value by more than machine epsilon.  Returns whether anything changed (the
// reference reports "calibration sample did not produce coefficient updates" otherwise); zero `runs` is the caller's error to raise.
inline bool apply_calibration_sample(Thresholds& t, const CalibrationSample& s, CalibrationDelta* delta = nullptr) {
    bool changed = false;
    auto upd = [&](double ms, double units, double* slot, std::optional<std::pair<double, double>>* d) {
        if (!(units > 0.0 && ms > 0.0)) return;
        const double per_unit = (ms / 1000.0) / units;
        if (!std::isfinite(per_unit) || per_unit <= 0.0 || !(std::fabs(*slot - per_unit) > std::numeric_limits<double>::epsilon())) return;
        if (d) *d = std::make_pair(*slot, per_unit);
        *slot = per_unit;
        changed = true;
    };
    CalibrationDelta local;
    CalibrationDelta* d = delta ? delta : &local;
    upd(s.cpu_ms_elementwise, s.units_elementwise, &t.cpu_elem_per_elem, &d->cpu_elem_per_elem);
    upd(s.cpu_ms_reduction, s.units_reduction, &t.cpu_reduction_per_elem, &d->cpu_reduction_per_elem);
    upd(s.cpu_ms_matmul, s.units_matmul_flops, &t.cpu_matmul_per_flop, &d->cpu_matmul_per_flop);
    return changed;
}

// does the sample's provider block describe this device?  (`apply_auto_offload_calibration_from_file` only warns on a mismatch)
inline bool provider_matches(const CalibrationProvider& p, const std::string& name, const std::string& vendor, const std::optional<std::string>& backend,
                             uint32_t device_id) {
    return p.name == name && p.vendor == vendor && p.backend == backend && p.device_id == device_id;
}

// ---- precision policy of a promotion (crates/runmat-accelerate/src/precision.rs) ----------------------------------------------------
// Before a host tensor is promoted the caller checks that the provider can run its logical dtype: an F64 provider
// (`rmhip_set_precision(ctx, 64)`, the default) takes f64 and f32 data, an F32 provider only f32 - unless the user allowed the implicit
// downcast of doubles with RUNMAT_ALLOW_PRECISION_DOWNCAST (parsed like the reference's `parse_bool`); integer classes never go.
enum class NumericDType { F64, F32, U8, U16, U32 };
inline std::optional<bool> parse_bool(const std::string& text) {  // precision.rs:22-28
    size_t b = 0, e = text.size();
    while (b < e && (text[b] == ' ' || text[b] == '\t' || text[b] == '\n' || text[b] == '\r')) ++b;
    while (e > b && (text[e - 1] == ' ' || text[e - 1] == '\t' || text[e - 1] == '\n' || text[e - 1] == '\r')) --e;
    std::string v;
    for (size_t i = b; i < e; ++i) v += (char)(text[i] >= 'A' && text[i] <= 'Z' ? text[i] - 'A' + 'a' : text[i]);
    if (v == "1" || v == "true" || v == "yes" || v == "on") return true;
    if (v == "0" || v == "false" || v == "no" || v == "off") return false;
    return std::nullopt;
}
// `provider_bits` = rmhip_buffer_bits / the context's precision: 64 or 32
inline bool provider_supports_dtype(int provider_bits, NumericDType dtype) {  // precision.rs:40-46
    switch (dtype) {
        case NumericDType::F32: return true;
        case NumericDType::F64: return provider_bits == 64;
        default: return false;
    }
}
// empty string = go ahead (possibly with the one-time downcast warning, *downcast set); otherwise the reference's refusal text
inline std::string ensure_provider_supports_dtype(int provider_bits, NumericDType dtype, bool allow_downcast, bool* downcast = nullptr) {  // precision.rs:53-81
    if (downcast) *downcast = false;
    if (provider_supports_dtype(provider_bits, dtype)) return std::string();
    if (dtype == NumericDType::F64 && allow_downcast) {
        if (downcast) *downcast = true;
        return std::string();
    }
    switch (dtype) {
        case NumericDType::F64: return "active provider does not advertise f64 kernels; refusing implicit downcast";
        case NumericDType::F32: return "active provider does not support f32 kernels";
        case NumericDType::U8: return "active provider does not support uint8 kernels";
        case NumericDType::U16: return "active provider does not support uint16 kernels";
        default: return "active provider does not support uint32 kernels";
    }
}

// ---- the decision ----------------------------------------------------------------------------------------------------------------
enum class UnaryOp { Generic, Transpose };
enum class Fusion { None, ElementwiseOrReductionSupported, Other };  // the active fusion group, as far as the decision looks at it

// the batch extent the small-batch guard looks at: the LAST extent of an operand of rank >= 3 (the smallest over the operands)
inline std::optional<size_t> batch_dimension(const std::vector<std::vector<size_t>>& operand_shapes) {
    std::optional<size_t> best;
    for (const auto& shape : operand_shapes)
        if (shape.size() >= 3 && (!best || shape.back() < *best)) best = shape.back();
    return best;
}

class Planner {
public:
    Thresholds thresholds;
    bool enabled = true;
    std::optional<ProfileCostModel> profile;

    static std::optional<double> cpu_estimate(double per_unit, size_t units) {
        if (std::isfinite(per_unit) && per_unit > 0.0) return per_unit * (double)units;
        return std::nullopt;
    }
    // many elements in few slabs (a 1e6-element array whose last extent is <= 8): the per-slab launches would dominate
    bool small_batch_guard(size_t elements, std::optional<size_t> batch) const {
        if (!enabled || !batch || *batch == 0) return false;
        return thresholds.small_batch_max_dim > 0 && thresholds.small_batch_min_elems > 0 && *batch <= thresholds.small_batch_max_dim &&
               elements >= thresholds.small_batch_min_elems;
    }
    // order of the rules: residency, fusion, small-batch guard, profile model, threshold
    Decision evaluate_elementwise(size_t elements, bool any_operand_resident, std::optional<size_t> batch = std::nullopt, Fusion fusion = Fusion::None) const {
        Decision d;
        d.cpu_secs = cpu_estimate(thresholds.cpu_elem_per_elem, elements);
        d.threshold = thresholds.binary_min_elems;
        d.batch = batch;
        if (!enabled) return d.reason = Reason::Disabled, d;
        if (any_operand_resident) return d.gpu = true, d.reason = Reason::Residency, d;  // keep a chain where its data already is
        if (fusion == Fusion::ElementwiseOrReductionSupported) return d.gpu = true, d.reason = Reason::FusionOverride, d;
        if (small_batch_guard(elements, batch)) return d.reason = Reason::SmallBatchGuard, d;
        if (profile)
            if (const auto g = profile->elem ? profile->elem->estimate((double)elements) : std::nullopt) return by_model(d, *g);
        return d.gpu = elements >= thresholds.binary_min_elems, d;
    }
    Decision evaluate_unary(size_t elements, UnaryOp op, bool operand_resident, std::optional<size_t> batch = std::nullopt) const {
        Decision d;
        d.cpu_secs = cpu_estimate(thresholds.cpu_elem_per_elem, elements);
        d.threshold = thresholds.unary_min_elems;
        d.batch = batch;
        if (!enabled) return d.reason = Reason::Disabled, d;
        if (operand_resident) return d.gpu = true, d.reason = Reason::Residency, d;
        if (op == UnaryOp::Generic && small_batch_guard(elements, batch)) return d.reason = Reason::SmallBatchGuard, d;
        if (profile) {
            const auto& lm = op == UnaryOp::Transpose ? profile->transpose : profile->elem;
            if (const auto g = lm ? lm->estimate((double)elements) : std::nullopt) return by_model(d, *g);
        }
        return d.gpu = elements >= thresholds.unary_min_elems, d;
    }
    Decision evaluate_reduction(size_t elements) const {
        Decision d;
        d.cpu_secs = cpu_estimate(thresholds.cpu_reduction_per_elem, elements);
        d.threshold = thresholds.reduction_min_elems;
        if (!enabled) return d.reason = Reason::Disabled, d;
        if (profile)
            if (const auto g = profile->reduction ? profile->reduction->estimate((double)elements) : std::nullopt) return by_model(d, *g);
        return d.gpu = elements >= thresholds.reduction_min_elems, d;
    }
    // `flops` = m * k * n, the unit of the threshold and of the profile's matmul samples
    Decision evaluate_matmul(size_t flops) const {
        Decision d;
        d.cpu_secs = cpu_estimate(thresholds.cpu_matmul_per_flop, flops);
        d.threshold = thresholds.matmul_min_flops;
        if (!enabled) return d.reason = Reason::Disabled, d;
        if (profile)
            if (const auto g = profile->matmul ? profile->matmul->estimate((double)flops) : std::nullopt) return by_model(d, *g);
        return d.gpu = flops >= thresholds.matmul_min_flops, d;
    }

private:
    static Decision by_model(Decision d, double gpu_secs) {  // the device has to win by 5 %
        d.gpu_secs = gpu_secs;
        d.reason = Reason::ProfileModel;
        d.gpu = gpu_secs * 0.95 < d.cpu_secs.value_or(std::numeric_limits<double>::infinity());
        return d;
    }
};

}  // namespace auto_offload
}  // namespace rmhip
