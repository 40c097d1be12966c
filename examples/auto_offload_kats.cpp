// auto_offload_kats.cpp -- known-answer tests of the auto-offload decision mirror (include/rmhip_auto_offload.hpp) against the rules of
// crates/runmat-accelerate/src/native_auto.rs (cited per block).  Pure host code: runs on the CPU (tests/test_auto_offload.py).
// Usage: auto_offload_kats [calibration.json]                       - with a file: also load it and print the coefficients it yields
//        auto_offload_kats --profile profile.json [calibration.json] - fit the GPU cost models of a profile file and print decisions
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "rmhip_auto_offload.hpp"

using namespace rmhip::auto_offload;

static int failures = 0;
#define CHECK(...)                                                           \
    do {                                                                     \
        if (!(__VA_ARGS__)) {                                                       \
            std::fprintf(stderr, "FAILED %s:%d  %s\n", __FILE__, __LINE__, #__VA_ARGS__); \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

static bool near(double a, double b, double rel = 1e-12) { return std::fabs(a - b) <= rel * std::fmax(std::fabs(a), std::fabs(b)); }

int main(int argc, char** argv) {
    // ---- defaults (native_auto.rs:26-30, 55-82) ----
    Thresholds d;
    CHECK(d.unary_min_elems == 4096 && d.binary_min_elems == 4096 && d.reduction_min_elems == 256 && d.matmul_min_flops == 1000000);
    CHECK(d.cpu_elem_per_elem == 1.0e-7 && d.cpu_reduction_per_elem == 1.2e-7 && d.cpu_matmul_per_flop == 2.5e-11);
    CHECK(d.small_batch_max_dim == 8 && d.small_batch_min_elems == 1048576);

    // ---- thresholds decide when nothing else speaks (:989-996, 1018-1026, 1048-1057, 1109-1117) ----
    Planner p;
    CHECK(!p.evaluate_elementwise(4095, false).gpu && p.evaluate_elementwise(4096, false).gpu);
    CHECK(p.evaluate_elementwise(4096, false).reason == Reason::Threshold && *p.evaluate_elementwise(10, false).threshold == 4096);
    CHECK(!p.evaluate_unary(4095, UnaryOp::Generic, false).gpu && p.evaluate_unary(4096, UnaryOp::Transpose, false).gpu);
    CHECK(!p.evaluate_reduction(255).gpu && p.evaluate_reduction(256).gpu && *p.evaluate_reduction(1).threshold == 256);
    CHECK(!p.evaluate_matmul(999999).gpu && p.evaluate_matmul(1000000).gpu);  // 100^3: the documented "roughly 100x100x100"
    CHECK(near(*p.evaluate_elementwise(1000, false).cpu_secs, 1.0e-4) && near(*p.evaluate_reduction(1000).cpu_secs, 1.2e-4) &&
          near(*p.evaluate_matmul(1000000).cpu_secs, 2.5e-5));
    CHECK(!p.evaluate_elementwise(10, false).gpu_secs);

    // ---- residency beats everything, then the fusion group, then the small-batch guard (:929-973, 1064-1085) ----
    CHECK(p.evaluate_elementwise(1, true).gpu && p.evaluate_elementwise(1, true).reason == Reason::Residency);
    CHECK(p.evaluate_unary(1, UnaryOp::Generic, true).reason == Reason::Residency);
    CHECK(p.evaluate_elementwise(1, false, std::nullopt, Fusion::ElementwiseOrReductionSupported).gpu);
    CHECK(p.evaluate_elementwise(1, false, std::nullopt, Fusion::ElementwiseOrReductionSupported).reason == Reason::FusionOverride);
    CHECK(!p.evaluate_elementwise(1, false, std::nullopt, Fusion::Other).gpu);
    CHECK(p.evaluate_elementwise(1, true, 4, Fusion::ElementwiseOrReductionSupported).reason == Reason::Residency);
    // 2^20 elements in <= 8 slabs: CPU although far above the element threshold; a resident operand or a fusion group still wins
    Decision g = p.evaluate_elementwise(1048576, false, 8);
    CHECK(!g.gpu && g.reason == Reason::SmallBatchGuard && *g.batch == 8);
    CHECK(p.evaluate_elementwise(1048576, false, 9).gpu && p.evaluate_elementwise(1048575, false, 8).gpu);
    CHECK(p.evaluate_elementwise(1048576, true, 8).reason == Reason::Residency);
    CHECK(p.evaluate_elementwise(1048576, false, 8, Fusion::ElementwiseOrReductionSupported).reason == Reason::FusionOverride);
    CHECK(p.evaluate_unary(1048576, UnaryOp::Generic, false, 3).reason == Reason::SmallBatchGuard);
    CHECK(p.evaluate_unary(1048576, UnaryOp::Transpose, false, 3).gpu);  // the guard is for generic unary ops only
    CHECK(!p.small_batch_guard(1048576, std::nullopt) && !p.small_batch_guard(1048576, 0) && p.small_batch_guard(1048576, 1));
    // the batch extent is the LAST extent of an operand of rank >= 3, the smallest over the operands (:570-583)
    CHECK(!batch_dimension({{1024, 1024}}) && *batch_dimension({{512, 512, 4}}) == 4 && *batch_dimension({{8, 8, 16}, {8, 8, 2}, {64, 64}}) == 2);

    // ---- disabled (:841-843: promotion returns its operands unchanged) ----
    Planner off;
    off.enabled = false;
    CHECK(!off.evaluate_elementwise(1u << 30, true).gpu && off.evaluate_elementwise(1u << 30, true).reason == Reason::Disabled);
    CHECK(!off.evaluate_matmul(1u << 30).gpu && !off.small_batch_guard(1048576, 1));
    CHECK(std::strcmp(reason_name(Reason::SmallBatchGuard), "small-batch-guard") == 0 && std::strcmp(reason_name(Reason::FusionOverride), "fusion-override") == 0 &&
          std::strcmp(reason_name(Reason::ProfileModel), "profile-model") == 0 && std::strcmp(reason_name(Reason::Residency), "residency") == 0);

    // ---- linear models (:1937-1955, 2044-2075) ----
    CHECK(!fit_linear_model({}));
    CHECK(near(fit_linear_model({{1000.0, 2.0e-3}})->slope, 2.0e-6) && fit_linear_model({{1000.0, 2.0e-3}})->intercept == 0.0);
    CHECK(!fit_linear_model({{0.0, 1.0}}));
    auto lm = fit_linear_model({{1000.0, 5.0e-6 + 1000 * 1.0e-9}, {3000.0, 5.0e-6 + 3000 * 1.0e-9}, {9000.0, 5.0e-6 + 9000 * 1.0e-9}});
    CHECK(lm && near(lm->slope, 1.0e-9, 1e-9) && near(lm->intercept, 5.0e-6, 1e-9));
    auto neg = fit_linear_model({{1000.0, 1.0e-6}, {2000.0, 3.0e-6}});  // the line crosses zero above the origin: the intercept is clamped
    CHECK(neg && neg->intercept == 0.0 && near(neg->slope, 2.0e-9));
    CHECK(!fit_linear_model({{1000.0, 1.0}, {1000.0, 2.0}}));           // no spread in x
    CHECK(!fit_linear_model({{1000.0, 2.0}, {2000.0, 1.0}}));           // falling: not a cost model
    CHECK(!LinearModel{0.0, 1.0}.estimate(10.0) && !LinearModel{-1.0, 1.0}.estimate(10.0) && near(*LinearModel{2.0, 1.0}.estimate(10.0), 21.0));

    // ---- profile reports -> model -> decision (:965-2014, 975-987): the device must win by 5 % ----
    std::vector<ProfileReport> reps = {
        {"elementwise", {{1000, 1}}, 0.006}, {"elementwise", {{1000, 1000}}, 0.010},       // 5.996 us + 4.004e-12 s / element
        {"reduction", {{100, 100}}, 0.008},                                                  // one sample: through the origin
        {"transpose", {{0, 5}}, 1.0},                                                        // empty operand: skipped
        {"matmul", {{100, 50}, {50, 20}}, 0.020}, {"matmul", {{1000, 1000}, {1000, 1000}}, 0.220},
        {"matmul", {{4, 4, 4}, {4, 4}}, 1.0},                                                // not 2-D: skipped
        {"fft", {{16}}, 1.0}};
    ProfileCostModel m = ProfileCostModel::from_reports(reps);
    CHECK(m.elem && m.reduction && !m.transpose && m.matmul);
    CHECK(near(m.reduction->slope, 8.0e-6 / 1.0e4) && m.reduction->intercept == 0.0);
    CHECK(near(*m.matmul->estimate(1.0e9), 2.2e-4, 1e-9) && near(*m.matmul->estimate(1.0e5), 2.0e-5, 1e-6));
    Planner q;
    q.profile = m;
    Decision e1 = q.evaluate_elementwise(100, false);   // ~6 us on the device against 10 us of CPU estimate
    CHECK(e1.reason == Reason::ProfileModel && e1.gpu && e1.gpu_secs && near(*e1.gpu_secs, 5.996e-6 + 100 * 4.004004e-12, 1e-3));
    CHECK(!q.evaluate_elementwise(50, false).gpu);        // 6 us against 5 us
    q.thresholds.cpu_elem_per_elem = 6.0e-8;              // 100 elements: 6 us of CPU against 0.95 * 6 us: the device still wins ...
    CHECK(q.evaluate_elementwise(100, false).gpu);
    q.thresholds.cpu_elem_per_elem = 5.6e-8;              // ... 5.6 us against 5.7 us: it does not
    CHECK(!q.evaluate_elementwise(100, false).gpu);
    CHECK(q.evaluate_unary(1u << 20, UnaryOp::Transpose, false).reason == Reason::Threshold);  // no transpose model: the threshold rule
    CHECK(q.evaluate_unary(1u << 20, UnaryOp::Generic, false).reason == Reason::ProfileModel);
    CHECK(q.evaluate_reduction(10000).reason == Reason::ProfileModel && q.evaluate_matmul(1000000).reason == Reason::ProfileModel);
    CHECK(q.evaluate_elementwise(5, true).reason == Reason::Residency);  // the model does not override residency
    q.thresholds.cpu_matmul_per_flop = std::nan("");      // no CPU estimate: the device is taken to win
    CHECK(q.evaluate_matmul(1000).gpu && !q.evaluate_matmul(1000).cpu_secs);

    const std::vector<ProfileReport> parsed = load_profile_reports(R"([{"category": "elementwise", "input_shapes": [[1000, 1]], "total_ms": {"avg_ms": 0.006, "p95_ms": 1}},
        {"category": "matmul", "input_shapes": [[100, 50], [50, 20]], "total_ms": {"avg_ms": 0.02}}, {"input_shapes": []}, {"category": "reduction", "total_ms": {}}])");
    CHECK(parsed.size() == 3 && parsed[0].category == "elementwise" && parsed[0].input_shapes[0][0] == 1000 && parsed[0].avg_total_ms == 0.006 &&
          parsed[1].input_shapes.size() == 2 && parsed[1].input_shapes[1][1] == 20 && parsed[2].avg_total_ms == 0.0 && parsed[2].input_shapes.empty());

    // ---- precision policy (precision.rs:22-81) ----
    CHECK(*parse_bool(" Yes\n") && *parse_bool("ON") && !*parse_bool("off") && !*parse_bool("0") && !parse_bool("maybe") && !parse_bool(""));
    CHECK(provider_supports_dtype(64, NumericDType::F64) && provider_supports_dtype(64, NumericDType::F32) && provider_supports_dtype(32, NumericDType::F32) &&
          !provider_supports_dtype(32, NumericDType::F64) && !provider_supports_dtype(64, NumericDType::U8));
    bool down = true;
    CHECK(ensure_provider_supports_dtype(64, NumericDType::F64, false, &down).empty() && !down);
    CHECK(ensure_provider_supports_dtype(32, NumericDType::F64, false) == "active provider does not advertise f64 kernels; refusing implicit downcast");
    CHECK(ensure_provider_supports_dtype(32, NumericDType::F64, true, &down).empty() && down);  // RUNMAT_ALLOW_PRECISION_DOWNCAST
    CHECK(ensure_provider_supports_dtype(64, NumericDType::U16, true) == "active provider does not support uint16 kernels");

    // ---- environment overrides (:1416-1449) ----
    std::map<std::string, std::string> env = {{"RUNMAT_ACCEL_THRESHOLD_UNARY", "100"}, {"RUNMAT_ACCEL_THRESHOLD_MATMUL", "12345"},
                                              {"RUNMAT_ACCEL_THRESHOLD_REDUCTION", "x1"}, {"RUNMAT_ACCEL_SMALL_BATCH_MAX_DIM", "0"}};
    auto lookup = [&](const char* k) -> const char* {
        auto it = env.find(k);
        return it == env.end() ? nullptr : it->second.c_str();
    };
    Thresholds t;
    CHECK(apply_env_overrides(t, lookup) && t.unary_min_elems == 100 && t.binary_min_elems == 4096 && t.reduction_min_elems == 256 && t.matmul_min_flops == 12345 &&
          t.small_batch_max_dim == 0);
    env = {{"RUNMAT_ACCEL_THRESHOLD_ALL", "7"}, {"RUNMAT_ACCEL_THRESHOLD_UNARY", "100"}};
    Thresholds t2;
    CHECK(apply_env_overrides(t2, lookup) && t2.unary_min_elems == 7 && t2.binary_min_elems == 7 && t2.reduction_min_elems == 7 && t2.matmul_min_flops == 1000000);
    env = {{"RUNMAT_ACCEL_THRESHOLD_ALL", "-3"}};
    Thresholds t3;
    CHECK(!apply_env_overrides(t3, lookup) && t3.unary_min_elems == 4096);
    Planner guard_off;
    guard_off.thresholds = t;  // small_batch_max_dim == 0 switches the guard off
    CHECK(guard_off.evaluate_elementwise(1048576, false, 1).gpu);

    // ---- calibration samples (:330-476) ----
    const std::string top = R"({"auto_offload_calibration": {"runs": 3, "cpu_time_ms": {"elementwise": 2.0, "reduction": 0.0, "matmul": 4.0},
        "units": {"elementwise": 1000000, "reduction": 1000, "matmul_flops": 2.0e9},
        "provider": {"name": "dev \"A\"", "vendor": "AMD", "backend": "hip", "device_id": 2}, "extra": [1, {"a": null}, true]},
        "rmhip_break_even": {"note": "ignored"}})";
    CalibrationSample s = load_calibration_sample(top);
    CHECK(s.runs == 3 && s.cpu_ms_elementwise == 2.0 && s.units_matmul_flops == 2.0e9 && s.provider && s.provider->name == "dev \"A\"" && s.provider->device_id == 2 &&
          *s.provider->backend == "hip" && !s.provider_conflict);
    Thresholds c;
    CalibrationDelta delta;
    CHECK(apply_calibration_sample(c, s, &delta));
    CHECK(near(c.cpu_elem_per_elem, 2.0e-9) && c.cpu_reduction_per_elem == 1.2e-7 && near(c.cpu_matmul_per_flop, 2.0e-12));
    CHECK(delta.cpu_elem_per_elem && delta.cpu_elem_per_elem->first == 1.0e-7 && !delta.cpu_reduction_per_elem && delta.cpu_matmul_per_flop);
    CHECK(!apply_calibration_sample(c, s));  // the same sample again changes nothing: "did not produce coefficient updates"
    CHECK(provider_matches(*s.provider, "dev \"A\"", "AMD", std::string("hip"), 2) && !provider_matches(*s.provider, "dev \"A\"", "AMD", std::nullopt, 2) &&
          !provider_matches(*s.provider, "dev \"A\"", "AMD", std::string("hip"), 0));
    // a suite file: the nested section wins over the top-level one; members that are missing default to zero
    const std::string suite = R"({"suite": {"auto_offload_calibration": {"runs": 9, "units": {"reduction": 10}, "cpu_time_ms": {"reduction": 1}, "provider_conflict": true}},
        "auto_offload_calibration": {"runs": 1}})";
    CalibrationSample s2 = load_calibration_sample(suite);
    CHECK(s2.runs == 9 && s2.cpu_ms_reduction == 1.0 && s2.units_reduction == 10.0 && s2.units_elementwise == 0.0 && !s2.provider && s2.provider_conflict);
    Thresholds c2;
    CHECK(apply_calibration_sample(c2, s2) && near(c2.cpu_reduction_per_elem, 1.0e-4) && c2.cpu_elem_per_elem == 1.0e-7);
    bool threw = false;
    try {
        load_calibration_sample(R"({"something_else": {}})");
    } catch (const std::runtime_error& e) {
        threw = std::strstr(e.what(), "does not contain an auto_offload_calibration section") != nullptr;
    }
    CHECK(threw);
    threw = false;
    try {
        load_calibration_sample("{\"auto_offload_calibration\": {\"runs\": ");
    } catch (const std::runtime_error& e) {
        threw = std::strstr(e.what(), "failed to parse calibration file") != nullptr;
    }
    CHECK(threw);

    // ---- a GPU profile file (RUNMAT_ACCEL_PROFILE format), when given as `--profile file`: the fitted models and a few decisions ----
    if (argc > 2 && std::strcmp(argv[1], "--profile") == 0) {
        std::ifstream f(argv[2]);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::vector<ProfileReport> file_reports = load_profile_reports(ss.str());
        Planner fp;
        fp.profile = ProfileCostModel::from_reports(file_reports);
        if (argc > 3) {  // the CPU side from a calibration sample of the same box
            std::ifstream cf(argv[3]);
            std::stringstream cs;
            cs << cf.rdbuf();
            CHECK(apply_calibration_sample(fp.thresholds, load_calibration_sample(cs.str())));
        }
        CHECK(!file_reports.empty() && fp.profile->elem && fp.profile->reduction && fp.profile->matmul);
        std::printf("profile: reports %zu  elem %.6e %.6e  reduction %.6e %.6e  matmul %.6e %.6e\n", file_reports.size(), fp.profile->elem->slope,
                    fp.profile->elem->intercept, fp.profile->reduction->slope, fp.profile->reduction->intercept, fp.profile->matmul->slope, fp.profile->matmul->intercept);
        for (size_t n : {64u, 1024u, 2048u, 4096u, 65536u, 1048576u}) {
            const Decision de = fp.evaluate_elementwise(n, false), dr = fp.evaluate_reduction(n);
            std::printf("decide: elementwise %zu -> %s (%s)  reduction -> %s\n", n, de.gpu ? "gpu" : "cpu", reason_name(de.reason), dr.gpu ? "gpu" : "cpu");
        }
        for (size_t n : {8u, 16u, 32u, 48u, 128u}) std::printf("decide: matmul %zu^3 -> %s\n", n, fp.evaluate_matmul(n * n * n).gpu ? "gpu" : "cpu");
        return failures ? 1 : 0;
    }
    // ---- a file from the calibrator (tests/tools/offload_calibrate.cpp), when given ----
    if (argc > 1) {
        std::ifstream f(argv[1]);
        std::stringstream ss;
        ss << f.rdbuf();
        CalibrationSample fs = load_calibration_sample(ss.str());
        Thresholds ft;
        const bool changed = apply_calibration_sample(ft, fs);
        CHECK(fs.runs > 0 && changed && fs.provider && fs.provider->vendor == "AMD" && fs.provider->backend && *fs.provider->backend == "hip");
        std::printf("calibration: runs %zu  cpu_elem_per_elem %.6e  cpu_reduction_per_elem %.6e  cpu_matmul_per_flop %.6e  provider \"%s\"\n", fs.runs,
                    ft.cpu_elem_per_elem, ft.cpu_reduction_per_elem, ft.cpu_matmul_per_flop, fs.provider ? fs.provider->name.c_str() : "");
    }
    if (failures) {
        std::fprintf(stderr, "%d auto-offload KAT(s) failed\n", failures);
        return 1;
    }
    std::printf("auto-offload KATs ok\n");
    return 0;
}
