// wgsl_front.h -- front-end for the WGSL text the reference planner hands to
// `AccelProvider::fused_elementwise` / `fused_reduction`
// (crates/runmat-accelerate-api/src/lib.rs:2946-3008).
//
// The reference generates that text in crates/runmat-accelerate/src/fusion.rs:
//   elementwise body  :1710-1763 (single output) / :1632-1708 (multi output)
//       "    let tmp{K}: {T} = <expr>;"  per op, then "    output[k].data[g] = <expr>;"
//   reduction         :1765-2077
//       "    let val: {T} = <expr>;" over v, v1, ...; axis from the load addressing; OMITNAN const
//   expression vocabulary :2874-3026 (primitive_expr / builtin_expr), literals :3051-3057
// The reference's own test provider parses the same text (crates/runmat-vm/tests/fusion_gpu.rs:
// 886-1395), which is what makes it a stable contract.  This front-end is strict: anything
// outside the subset is an error (-> RMHIP_ERR_COMPILE -> caller falls back to CPU).
#pragma once

#include <memory>
#include <string>
#include <vector>

namespace rmhip {

struct Expr;
using ExprPtr = std::shared_ptr<Expr>;

struct Expr {
    enum Kind { Lit, Input, Tmp, RedVar, Unary, Binary, Call, Select };
    Kind kind = Lit;
    double value = 0.0;        // Lit
    std::string text;          // Lit: the number token as written ("1.0", "0.4342944819032518")
    int index = 0;             // Input / Tmp / RedVar
    std::string op;            // Unary ("-","+","!"), Binary ("+","-","*","/","<",...,"&&","||"), Call name
    std::vector<ExprPtr> args;
    bool paren = false;        // was written inside its own parentheses
    bool is_bool = false;      // comparison / logical / isNan-family result
};

struct EwStatement {
    int tmp = -1;      // let tmp{tmp}
    ExprPtr expr;
};

struct ElementwiseProgram {
    bool f32 = false;                // shader written for an F32 provider (`array<f32>`, `let tmpK: f32`)
    int n_inputs = 0;
    std::vector<EwStatement> lets;   // in order
    std::vector<ExprPtr> outputs;    // output k expression (usually a Tmp or Input leaf)
    std::string canonical;           // canonical rendering used as the cache key
};

struct ReductionProgram {
    bool f32 = false;
    int n_inputs = 0;
    int axis = 0;          // 0: slice s = contiguous [s*reduce_len, ...); 1: element (s, c) at s + c*num_slices
    bool omitnan = false;
    ExprPtr val;
    std::string canonical;
};

// Return true on success; on failure *err describes the offending construct.
bool parse_elementwise_wgsl(const std::string& shader, ElementwiseProgram* out, std::string* err);
bool parse_reduction_wgsl(const std::string& shader, ReductionProgram* out, std::string* err);

// Render an expression as HIP C++ (f64). Input leaves render as `x{i}`, tmps as `tmp{K}`,
// reduction vars as `v{i}`.
std::string emit_expr(const ExprPtr& e);
// Same, coerced to double (bools become 1.0 / 0.0).
std::string emit_expr_f64(const ExprPtr& e);

}  // namespace rmhip
