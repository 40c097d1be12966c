// wgsl_front.cpp -- strict parser for the reference planner's fused-kernel WGSL (see wgsl_front.h).
#include "wgsl_front.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <unordered_map>

namespace rmhip {
namespace {

struct Token {
    enum Kind { Ident, Number, Punct, End } kind = End;
    std::string text;
};

struct Lexer {
    const std::string& s;
    size_t pos = 0;
    std::string* err;
    explicit Lexer(const std::string& src, std::string* e) : s(src), err(e) {}

    Token next() {
        while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos;
        Token t;
        if (pos >= s.size()) return t;
        char c = s[pos];
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t b = pos;
            while (pos < s.size() && (std::isalnum((unsigned char)s[pos]) || s[pos] == '_')) ++pos;
            t.kind = Token::Ident;
            t.text = s.substr(b, pos - b);
            return t;
        }
        if (std::isdigit((unsigned char)c) || (c == '.' && pos + 1 < s.size() && std::isdigit((unsigned char)s[pos + 1]))) {
            size_t b = pos;
            if (c == '0' && pos + 1 < s.size() && (s[pos + 1] == 'x' || s[pos + 1] == 'X')) {
                pos += 2;
                while (pos < s.size() && std::isxdigit((unsigned char)s[pos])) ++pos;
            } else {
                while (pos < s.size() && std::isdigit((unsigned char)s[pos])) ++pos;
                if (pos < s.size() && s[pos] == '.') {
                    ++pos;
                    while (pos < s.size() && std::isdigit((unsigned char)s[pos])) ++pos;
                }
                if (pos < s.size() && (s[pos] == 'e' || s[pos] == 'E')) {
                    size_t save = pos;
                    ++pos;
                    if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) ++pos;
                    if (pos < s.size() && std::isdigit((unsigned char)s[pos])) {
                        while (pos < s.size() && std::isdigit((unsigned char)s[pos])) ++pos;
                    } else {
                        pos = save;
                    }
                }
            }
            t.kind = Token::Number;
            t.text = s.substr(b, pos - b);
            // type suffixes (u, i, f, h) are accepted and dropped
            if (pos < s.size() && (s[pos] == 'u' || s[pos] == 'i' || s[pos] == 'f' || s[pos] == 'h')) {
                size_t q = pos + 1;
                if (q >= s.size() || !(std::isalnum((unsigned char)s[q]) || s[q] == '_')) ++pos;
            }
            return t;
        }
        static const char* two[] = {"<=", ">=", "==", "!=", "&&", "||"};
        for (const char* op : two) {
            if (s.compare(pos, 2, op) == 0) {
                t.kind = Token::Punct;
                t.text = op;
                pos += 2;
                return t;
            }
        }
        t.kind = Token::Punct;
        t.text = std::string(1, c);
        ++pos;
        return t;
    }
};

struct Parser {
    Lexer lex;
    Token cur;
    std::string* err;
    bool ok = true;
    bool reduction_vars;  // allow v / vK leaves

    Parser(const std::string& src, std::string* e, bool red) : lex(src, e), err(e), reduction_vars(red) {
        cur = lex.next();
    }
    void advance() { cur = lex.next(); }
    bool fail(const std::string& m) {
        if (ok && err) *err = m;
        ok = false;
        return false;
    }
    bool is_punct(const char* p) const { return cur.kind == Token::Punct && cur.text == p; }
    bool expect(const char* p) {
        if (!is_punct(p)) return fail(std::string("expected '") + p + "' near '" + cur.text + "'");
        advance();
        return true;
    }

    static ExprPtr mk(Expr::Kind k) {
        auto e = std::make_shared<Expr>();
        e->kind = k;
        return e;
    }

    ExprPtr parse_expr() { return parse_or(); }

    ExprPtr binary(const std::string& op, ExprPtr l, ExprPtr r, bool is_bool) {
        auto e = mk(Expr::Binary);
        e->op = op;
        e->args = {l, r};
        e->is_bool = is_bool;
        return e;
    }
    ExprPtr parse_or() {
        ExprPtr l = parse_and();
        while (ok && is_punct("||")) {
            advance();
            ExprPtr r = parse_and();
            l = binary("||", l, r, true);
        }
        return l;
    }
    ExprPtr parse_and() {
        ExprPtr l = parse_cmp();
        while (ok && is_punct("&&")) {
            advance();
            ExprPtr r = parse_cmp();
            l = binary("&&", l, r, true);
        }
        return l;
    }
    ExprPtr parse_cmp() {
        ExprPtr l = parse_add();
        while (ok && cur.kind == Token::Punct &&
               (cur.text == "<" || cur.text == ">" || cur.text == "<=" || cur.text == ">=" ||
                cur.text == "==" || cur.text == "!=")) {
            std::string op = cur.text;
            advance();
            ExprPtr r = parse_add();
            l = binary(op, l, r, true);
        }
        return l;
    }
    ExprPtr parse_add() {
        ExprPtr l = parse_mul();
        while (ok && (is_punct("+") || is_punct("-"))) {
            std::string op = cur.text;
            advance();
            ExprPtr r = parse_mul();
            l = binary(op, l, r, false);
        }
        return l;
    }
    ExprPtr parse_mul() {
        ExprPtr l = parse_unary();
        while (ok && (is_punct("*") || is_punct("/") || is_punct("%"))) {
            std::string op = cur.text;
            advance();
            ExprPtr r = parse_unary();
            l = binary(op, l, r, false);
        }
        return l;
    }
    ExprPtr parse_unary() {
        if (is_punct("-") || is_punct("+") || is_punct("!")) {
            std::string op = cur.text;
            advance();
            ExprPtr a = parse_unary();
            if (!ok) return nullptr;
            if (op == "-" && a->kind == Expr::Lit && !a->paren) {  // fold negative literals
                auto e = mk(Expr::Lit);
                e->value = -a->value;
                e->text = "-" + a->text;
                return e;
            }
            auto e = mk(Expr::Unary);
            e->op = op;
            e->args = {a};
            e->is_bool = (op == "!");
            return e;
        }
        return parse_primary();
    }

    static bool all_digits(const std::string& s, size_t from) {
        if (from >= s.size()) return false;
        for (size_t i = from; i < s.size(); ++i)
            if (!std::isdigit((unsigned char)s[i])) return false;
        return true;
    }

    ExprPtr parse_primary() {
        if (!ok) return nullptr;
        if (cur.kind == Token::Number) {
            auto e = mk(Expr::Lit);
            e->text = cur.text;
            if (cur.text.size() > 2 && cur.text[0] == '0' && (cur.text[1] == 'x' || cur.text[1] == 'X'))
                e->value = (double)std::strtoull(cur.text.c_str(), nullptr, 16);
            else
                e->value = std::strtod(cur.text.c_str(), nullptr);
            advance();
            return e;
        }
        if (is_punct("(")) {
            advance();
            ExprPtr e = parse_expr();
            if (!ok) return nullptr;
            if (!expect(")")) return nullptr;
            // copy-on-paren so leaf sharing never leaks the flag
            auto p = std::make_shared<Expr>(*e);
            p->paren = true;
            return p;
        }
        if (cur.kind != Token::Ident) {
            fail("unexpected token '" + cur.text + "'");
            return nullptr;
        }
        std::string name = cur.text;
        advance();
        if (is_punct("(")) {  // call
            advance();
            std::vector<ExprPtr> args;
            if (!is_punct(")")) {
                while (ok) {
                    args.push_back(parse_expr());
                    if (!ok) return nullptr;
                    if (is_punct(",")) {
                        advance();
                        continue;
                    }
                    break;
                }
            }
            if (!expect(")")) return nullptr;
            if (name == "f64") {  // cast_literal / `f64({})` literal printing, fusion.rs:1839-1873,3051-3057
                if (args.size() != 1) {
                    fail("f64() takes one argument");
                    return nullptr;
                }
                auto a = std::make_shared<Expr>(*args[0]);
                a->paren = false;
                return a;
            }
            if (name == "select") {
                if (args.size() != 3) {
                    fail("select() takes three arguments");
                    return nullptr;
                }
                auto e = mk(Expr::Select);
                e->args = args;
                return e;
            }
            auto e = mk(Expr::Call);
            e->op = name;
            e->args = args;
            e->is_bool = (name == "isNan" || name == "isInf" || name == "isFinite" || name == "isNanF");
            return e;
        }
        if (name == "inf" || name == "NaN" || name == "nan") {  // Rust Display of non-finite f64 inside f64(..)
            auto e = mk(Expr::Lit);
            e->text = name;
            e->value = (name == "inf") ? std::numeric_limits<double>::infinity()
                                       : std::numeric_limits<double>::quiet_NaN();
            return e;
        }
        if (name == "true" || name == "false") {
            auto e = mk(Expr::Lit);
            e->text = name;
            e->value = name == "true" ? 1.0 : 0.0;
            e->is_bool = true;
            return e;
        }
        if (name.rfind("input", 0) == 0 && all_digits(name, 5)) {
            // input{i}.data[ <index expr> ]
            if (!expect(".")) return nullptr;
            if (cur.kind != Token::Ident || cur.text != "data") {
                fail("expected '.data' after " + name);
                return nullptr;
            }
            advance();
            if (!expect("[")) return nullptr;
            int depth = 1;  // the index expression is positional (i{k}); skip it
            while (cur.kind != Token::End && depth > 0) {
                if (is_punct("[")) ++depth;
                if (is_punct("]")) {
                    --depth;
                    if (depth == 0) break;
                }
                advance();
            }
            if (!expect("]")) return nullptr;
            auto e = mk(Expr::Input);
            e->index = std::atoi(name.c_str() + 5);
            return e;
        }
        if (name.rfind("tmp", 0) == 0 && all_digits(name, 3)) {
            auto e = mk(Expr::Tmp);
            e->index = std::atoi(name.c_str() + 3);
            return e;
        }
        if (reduction_vars && (name == "v" || (name[0] == 'v' && all_digits(name, 1)))) {
            auto e = mk(Expr::RedVar);
            e->index = name == "v" ? 0 : std::atoi(name.c_str() + 1);
            return e;
        }
        fail("unknown identifier '" + name + "'");
        return nullptr;
    }
};

bool lit_text_is(const ExprPtr& e, const char* t) { return e->kind == Expr::Lit && e->text == t; }

// CPU-parity rewrites of the three lossy forms the generator emits for log10 / log1p / expm1
// (fusion.rs:3005-3019): the CPU builtins call libm log10 / ln_1p / exp_m1 (SURVEY.md "Parity
// hazards").  The literal spellings ("0.4342944819032518", "1.0") and the missing inner
// parentheses are unique to those generator branches: user-written `log(x+1)` arrives as two
// tmps, and reduction constants print as `f64(1)`.
ExprPtr rewrite(const ExprPtr& e) {
    if (!e) return e;
    auto out = std::make_shared<Expr>(*e);
    for (auto& a : out->args) a = rewrite(a);
    if (out->kind == Expr::Binary && out->op == "*" && out->paren && out->args[0]->kind == Expr::Call &&
        out->args[0]->op == "log" && out->args[0]->args.size() == 1 &&
        lit_text_is(out->args[1], "0.4342944819032518")) {
        auto c = std::make_shared<Expr>();
        c->kind = Expr::Call;
        c->op = "log10";
        c->args = {out->args[0]->args[0]};
        return c;
    }
    if (out->kind == Expr::Call && out->op == "log" && out->args.size() == 1) {
        const ExprPtr& a = out->args[0];
        if (a->kind == Expr::Binary && a->op == "+" && !a->paren && lit_text_is(a->args[1], "1.0")) {
            auto c = std::make_shared<Expr>();
            c->kind = Expr::Call;
            c->op = "log1p";
            c->args = {a->args[0]};
            return c;
        }
    }
    if (out->kind == Expr::Binary && out->op == "-" && out->paren && out->args[0]->kind == Expr::Call &&
        out->args[0]->op == "exp" && out->args[0]->args.size() == 1 && lit_text_is(out->args[1], "1.0")) {
        auto c = std::make_shared<Expr>();
        c->kind = Expr::Call;
        c->op = "expm1";
        c->args = {out->args[0]->args[0]};
        return c;
    }
    return out;
}

struct FnInfo {
    const char* hip;
    int arity;
};
const std::unordered_map<std::string, FnInfo>& fn_table() {
    // WGSL name (fusion.rs:2932-3026; test vocabulary fusion_gpu.rs:1291-1333) -> device function.
    static const std::unordered_map<std::string, FnInfo> t = {
        {"sin", {"rm_sin", 1}},    {"cos", {"rm_cos", 1}},       {"tan", {"tan", 1}},
        {"asin", {"asin", 1}},     {"acos", {"acos", 1}},     {"atan", {"atan", 1}},
        {"sinh", {"sinh", 1}},     {"cosh", {"cosh", 1}},     {"tanh", {"tanh", 1}},
        {"asinh", {"asinh", 1}},   {"acosh", {"acosh", 1}},   {"atanh", {"atanh", 1}},
        {"exp", {"exp", 1}},       {"exp2", {"exp2", 1}},     {"log", {"log", 1}},
        {"log2", {"log2", 1}},     {"log10", {"log10", 1}},   {"log1p", {"log1p", 1}},
        {"expm1", {"expm1", 1}},   {"sqrt", {"sqrt", 1}},     {"abs", {"fabs", 1}},
        {"floor", {"floor", 1}},   {"ceil", {"ceil", 1}},     {"round", {"round", 1}},
        {"trunc", {"trunc", 1}},   {"sign", {"rm_sign", 1}},  {"atan2", {"atan2", 2}},
        {"hypot", {"hypot", 2}},   {"pow", {"rm_pow", 2}},       {"max", {"rm_max", 2}},
        {"min", {"rm_min", 2}},    {"isNan", {"rm_isnan", 1}}, {"isNanF", {"rm_isnan", 1}},
        {"isInf", {"rm_isinf", 1}}, {"isFinite", {"rm_isfinite", 1}}, {"f32", {"rm_f32", 1}},
    };
    return t;
}

bool validate(const ExprPtr& e, int n_inputs, int max_tmp, bool reduction, std::string* err) {
    if (!e) return false;
    switch (e->kind) {
        case Expr::Input:
            if (reduction || e->index < 0 || e->index >= n_inputs) {
                *err = "input index out of range";
                return false;
            }
            break;
        case Expr::RedVar:
            if (!reduction || e->index < 0 || e->index >= n_inputs) {
                *err = "reduction variable index out of range";
                return false;
            }
            break;
        case Expr::Tmp:
            if (reduction || e->index < 0 || e->index >= max_tmp) {
                *err = "tmp used before definition";
                return false;
            }
            break;
        case Expr::Call: {
            auto it = fn_table().find(e->op);
            if (it == fn_table().end()) {
                *err = "unsupported function '" + e->op + "'";
                return false;
            }
            if ((int)e->args.size() != it->second.arity) {
                *err = "wrong argument count for '" + e->op + "'";
                return false;
            }
            break;
        }
        default:
            break;
    }
    for (const auto& a : e->args)
        if (!validate(a, n_inputs, max_tmp, reduction, err)) return false;
    return true;
}

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

int count_inputs(const std::string& shader) {
    int n = 0;
    size_t pos = 0;
    const std::string key = "var<storage, read> input";
    while ((pos = shader.find(key, pos)) != std::string::npos) {
        ++n;
        pos += key.size();
    }
    return n;
}

// The planner emits the shader in the provider's precision (`scalar_ty`, fusion.rs:1525-1533): f64 or f32.
bool check_scalar_type(const std::string& shader, bool* is_f32, std::string* err) {
    *is_f32 = false;
    if (shader.find("data: array<f64>") != std::string::npos) return true;
    if (shader.find("data: array<f32>") != std::string::npos) {
        *is_f32 = true;
        return true;
    }
    *err = "shader has no `struct Tensor { data: array<f64> }` (or f32) declaration";
    return false;
}

}  // namespace

std::string emit_expr(const ExprPtr& e);

static std::string as_bool(const ExprPtr& e) {
    if (e->is_bool) return emit_expr(e);
    return "(" + emit_expr(e) + " != 0.0)";
}

std::string emit_expr_f64(const ExprPtr& e) {
    if (e->is_bool) return "(" + emit_expr(e) + " ? 1.0 : 0.0)";
    return emit_expr(e);
}

std::string emit_expr(const ExprPtr& e) {
    char buf[96];
    switch (e->kind) {
        case Expr::Lit:
            if (e->is_bool) return e->value != 0.0 ? "true" : "false";
            if (std::isnan(e->value)) return "__builtin_nan(\"\")";
            if (std::isinf(e->value)) return e->value > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
            std::snprintf(buf, sizeof buf, "%a", e->value);  // exact hex-float
            return std::string("(") + buf + ")";
        case Expr::Input:
            return "x" + std::to_string(e->index);
        case Expr::Tmp:
            return "tmp" + std::to_string(e->index);
        case Expr::RedVar:
            return "v" + std::to_string(e->index);
        case Expr::Unary:
            if (e->op == "!") return "(!" + as_bool(e->args[0]) + ")";
            if (e->op == "+") return "(" + emit_expr_f64(e->args[0]) + ")";
            return "(-" + emit_expr_f64(e->args[0]) + ")";
        case Expr::Binary:
            if (e->op == "&&" || e->op == "||")
                return "(" + as_bool(e->args[0]) + " " + e->op + " " + as_bool(e->args[1]) + ")";
            if (e->op == "%")
                return "fmod(" + emit_expr_f64(e->args[0]) + ", " + emit_expr_f64(e->args[1]) + ")";
            return "(" + emit_expr_f64(e->args[0]) + " " + e->op + " " + emit_expr_f64(e->args[1]) + ")";
        case Expr::Call: {
            const FnInfo& f = fn_table().at(e->op);
            std::string s = std::string(f.hip) + "(";
            for (size_t i = 0; i < e->args.size(); ++i) {
                if (i) s += ", ";
                s += emit_expr_f64(e->args[i]);
            }
            return s + ")";
        }
        case Expr::Select:  // select(f, t, cond)
            if (e->args[0]->is_bool && e->args[1]->is_bool)
                return "(" + as_bool(e->args[2]) + " ? " + emit_expr(e->args[1]) + " : " + emit_expr(e->args[0]) + ")";
            return "(" + as_bool(e->args[2]) + " ? " + emit_expr_f64(e->args[1]) + " : " +
                   emit_expr_f64(e->args[0]) + ")";
    }
    return "0.0";
}

bool parse_elementwise_wgsl(const std::string& shader, ElementwiseProgram* out, std::string* err) {
    std::string local;
    if (!err) err = &local;
    *out = ElementwiseProgram();
    if (!check_scalar_type(shader, &out->f32, err)) return false;
    const std::string scalar_ty = out->f32 ? "f32" : "f64";
    out->n_inputs = count_inputs(shader);
    if (out->n_inputs == 0) {
        *err = "fused_elementwise: no inputs";  // elementwise.rs:1574-1576
        return false;
    }
    std::istringstream in(shader);
    std::string line;
    std::vector<std::pair<int, ExprPtr>> outs;
    while (std::getline(in, line)) {
        std::string t = trim(line);
        if (t.rfind("let tmp", 0) == 0) {
            size_t colon = t.find(':');
            size_t eq = t.find('=');
            if (colon == std::string::npos || eq == std::string::npos || eq < colon || t.back() != ';') {
                *err = "malformed let statement: " + t;
                return false;
            }
            std::string idx = trim(t.substr(7, colon - 7));
            if (idx.empty() || idx.find_first_not_of("0123456789") != std::string::npos) {
                *err = "malformed tmp name: " + t;
                return false;
            }
            std::string ty = trim(t.substr(colon + 1, eq - colon - 1));
            if (ty != scalar_ty) {
                *err = "unsupported scalar type '" + ty + "'";
                return false;
            }
            std::string rhs = t.substr(eq + 1, t.size() - eq - 2);
            Parser p(rhs, err, false);
            ExprPtr e = p.parse_expr();
            if (!p.ok || !e) return false;
            if (p.cur.kind != Token::End) {
                *err = "trailing tokens in: " + t;
                return false;
            }
            int k = std::atoi(idx.c_str());
            if (k != (int)out->lets.size()) {
                *err = "tmp statements out of order";
                return false;
            }
            e = rewrite(e);
            if (!validate(e, out->n_inputs, k, false, err)) return false;
            out->lets.push_back({k, e});
        } else if (t.rfind("output", 0) == 0 && t.find(".data[g]") != std::string::npos &&
                   t.find("var<") == std::string::npos) {
            size_t dot = t.find(".data[g]");
            std::string name = t.substr(0, dot);
            int k = 0;
            if (name != "output") {
                std::string digits = name.substr(6);
                if (digits.empty() || digits.find_first_not_of("0123456789") != std::string::npos) {
                    *err = "malformed output name: " + t;
                    return false;
                }
                k = std::atoi(digits.c_str());
            }
            size_t eq = t.find('=', dot);
            if (eq == std::string::npos || t.back() != ';') {
                *err = "malformed output statement: " + t;
                return false;
            }
            std::string rhs = t.substr(eq + 1, t.size() - eq - 2);
            Parser p(rhs, err, false);
            ExprPtr e = p.parse_expr();
            if (!p.ok || !e) return false;
            if (p.cur.kind != Token::End) {
                *err = "trailing tokens in: " + t;
                return false;
            }
            e = rewrite(e);
            if (!validate(e, out->n_inputs, (int)out->lets.size(), false, err)) return false;
            outs.push_back({k, e});
        }
    }
    if (outs.empty()) {
        *err = "shader has no output store";
        return false;
    }
    out->outputs.resize(outs.size());
    for (auto& kv : outs) {
        if (kv.first < 0 || kv.first >= (int)outs.size() || out->outputs[kv.first]) {
            *err = "output bindings are not 0..N-1";
            return false;
        }
        out->outputs[kv.first] = kv.second;
    }
    std::string canon = "ew|" + std::to_string(out->n_inputs) + "|";
    for (auto& st : out->lets) canon += "t" + std::to_string(st.tmp) + "=" + emit_expr_f64(st.expr) + ";";
    for (size_t k = 0; k < out->outputs.size(); ++k)
        canon += "o" + std::to_string(k) + "=" + emit_expr_f64(out->outputs[k]) + ";";
    out->canonical = canon;
    return true;
}

bool parse_reduction_wgsl(const std::string& shader, ReductionProgram* out, std::string* err) {
    std::string local;
    if (!err) err = &local;
    *out = ReductionProgram();
    if (!check_scalar_type(shader, &out->f32, err)) return false;
    out->n_inputs = count_inputs(shader);
    if (out->n_inputs == 0) {
        *err = "fused_reduction: no inputs";
        return false;
    }
    // axis from the load addressing (fusion.rs:2004 vs :2049)
    bool col = shader.find("input0.data[ (col * params.nrows) + r ]") != std::string::npos;
    bool row = shader.find("input0.data[ row + (c * params.ncols) ]") != std::string::npos;
    if (col == row) {
        *err = "cannot determine reduction axis from shader addressing";
        return false;
    }
    out->axis = col ? 0 : 1;
    out->omitnan = shader.find("const OMITNAN: bool = true") != std::string::npos;
    size_t pos = shader.find("let val:");
    if (pos == std::string::npos) {
        *err = "shader has no `let val:` statement";
        return false;
    }
    size_t eol = shader.find('\n', pos);
    std::string t = trim(shader.substr(pos, eol == std::string::npos ? std::string::npos : eol - pos));
    size_t colon = t.find(':');
    size_t eq = t.find('=');
    if (eq == std::string::npos || t.back() != ';') {
        *err = "malformed val statement: " + t;
        return false;
    }
    std::string ty = trim(t.substr(colon + 1, eq - colon - 1));
    if (ty != (out->f32 ? "f32" : "f64")) {
        *err = "unsupported scalar type '" + ty + "'";
        return false;
    }
    std::string rhs = t.substr(eq + 1, t.size() - eq - 2);
    Parser p(rhs, err, true);
    ExprPtr e = p.parse_expr();
    if (!p.ok || !e) return false;
    if (p.cur.kind != Token::End) {
        *err = "trailing tokens in: " + t;
        return false;
    }
    e = rewrite(e);
    if (!validate(e, out->n_inputs, 0, true, err)) return false;
    out->val = e;
    out->canonical = "red|" + std::to_string(out->n_inputs) + "|a" + std::to_string(out->axis) +
                     (out->omitnan ? "|omit|" : "|incl|") + emit_expr_f64(e);
    return true;
}

}  // namespace rmhip
