// codegen.cpp -- lowers a parsed fused-kernel program to HIP source, compiles it with hipRTC for
// gfx950 and caches the loaded function per context (see codegen.h).
#include "codegen.h"

#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rmhip {

static const char* kSkelCommon =
#include "skel_common_str.inc"
    ;
static const char* kSkelReduce =
#include "skel_reduce_str.inc"
    ;
// lazy random_normal operands: the Box-Muller tables and the stream's device functions (skel_rng.h), text for hipRTC like the above
static const char* kRngTables =
#include "rng_tables_str.inc"
    ;
static const char* kSkelRng =
#include "skel_rng_str.inc"
    ;

uint64_t fnv1a(const std::string& s) {
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ULL;
    }
    return h;
}

static int env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    int x = std::atoi(v);
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

EwTuning EwTuning::from_env() {
    EwTuning t;
    t.unroll = env_int("RMHIP_EW_UNROLL", t.unroll, 0, 8);
    t.block = env_int("RMHIP_EW_BLOCK", t.block, 64, 1024);
    t.block = (t.block / 64) * 64;
    t.blocks_per_cu = env_int("RMHIP_EW_BLOCKS_PER_CU", t.blocks_per_cu, 1, 64);
    t.nt_load = env_int("RMHIP_EW_NT_LOAD", env_int("RMHIP_EW_NT", t.nt_load, 0, 1), 0, 1);
    t.nt_store = env_int("RMHIP_EW_NT_STORE", env_int("RMHIP_EW_NT", t.nt_store, 0, 1), 0, 1);
    t.chunked = env_int("RMHIP_EW_CHUNKED", t.chunked, 0, 1);
    t.bcast_block = (env_int("RMHIP_EW_BCAST_BLOCK", t.bcast_block, 64, 1024) / 64) * 64;
    t.bcast_elems = env_int("RMHIP_EW_BCAST_ELEMS", t.bcast_elems, 1, 8);
    return t;
}

int EwTuning::unroll_for(int n_streamed_inputs, bool heavy_math) const {
    if (unroll > 0) return unroll;
    (void)n_streamed_inputs;
    (void)heavy_math;
    return 1;  // see codegen.h
}

static bool expr_is_heavy(const ExprPtr& e) {
    if (!e) return false;
    if (e->kind == Expr::Call) {
        static const char* cheap[] = {"abs", "floor", "ceil", "round", "trunc", "sign", "max", "min", "isNan",
                                      "isNanF", "isInf", "isFinite", "f32", "sqrt"};
        bool is_cheap = false;
        for (const char* c : cheap) is_cheap |= (e->op == c);
        if (!is_cheap) return true;
    }
    if (e->kind == Expr::Binary && e->op == "/") return true;  // f64 division expands to ~10 VALU ops
    for (const auto& a : e->args)
        if (expr_is_heavy(a)) return true;
    return false;
}

bool program_is_heavy(const ElementwiseProgram& p) {
    for (const auto& st : p.lets)
        if (expr_is_heavy(st.expr)) return true;
    for (const auto& o : p.outputs)
        if (expr_is_heavy(o)) return true;
    return false;
}

FusedKernel::~FusedKernel() {
    if (module) (void)hipModuleUnload(module);
}

// ---- source generation -------------------------------------------------------------------------

static std::string body_function(const ElementwiseProgram& p) {
    std::ostringstream s;
    s << "__device__ __forceinline__ void rm_body(";
    for (int k = 0; k < p.n_inputs; ++k) s << (k ? ", " : "") << "const double x" << k;
    for (size_t k = 0; k < p.outputs.size(); ++k) s << ", double& o" << k;
    s << ") {\n";
    for (const auto& st : p.lets) s << "    const double tmp" << st.tmp << " = " << emit_expr_f64(st.expr) << ";\n";
    for (size_t k = 0; k < p.outputs.size(); ++k) s << "    o" << k << " = " << emit_expr_f64(p.outputs[k]) << ";\n";
    s << "}\n\n";
    return s.str();
}

// One streaming kernel. `vec` = 2 (16-byte accesses) or 1. Bit k of `mask` marks input k as a
// 1-element tensor (the executor uploads scalars that way, fusion_exec.rs:305-326): it is read
// once into an SGPR-resident value instead of being streamed.
// Bit k of `rng_mask` (vec == 2, f64 only) marks input k as a LAZY random_normal operand: the kernel receives the stream state the
// tensor was drawn at instead of a pointer and generates pair i - elements 2i, 2i + 1, one 16-byte vector - in registers with the
// functions of skel_rng.h, following the state by the constant jump of its grid stride exactly as k_rng_normal does.
static void emit_fast_kernel(std::ostringstream& s, const ElementwiseProgram& p, const EwTuning& t_in, int vec,
                             unsigned mask, const char* name, bool f32, unsigned rng_mask = 0) {
    EwTuning t = t_in;
    if (rng_mask) t.chunked = 0;  // the state jump is one constant per launch: grid-stride only
    const int nin = p.n_inputs, nout = (int)p.outputs.size();
    // storage type S: f32 tensors are read and written as f32 (16-byte vectors of four), the body computes in f64
    const char* S = f32 ? "float" : "double";
    const char* vt = vec == 4 ? "rm_v4f" : vec == 2 ? "rm_v2" : S;
    const std::string cast_in = f32 ? "(double)" : "";
    const std::string cast_out = f32 ? "(float)" : "";
    auto is_scalar = [&](int k) { return (mask >> k) & 1u; };
    auto is_rng = [&](int k) { return (rng_mask >> k) & 1u; };
    int n_stream = 0;
    for (int k = 0; k < nin; ++k) n_stream += is_scalar(k) ? 0 : 1;
    const int U = t.unroll_for(n_stream, program_is_heavy(p));
    s << "extern \"C\" __global__ void __launch_bounds__(" << t.block << ") " << name << "(";
    for (int k = 0; k < nin; ++k) {
        if (is_rng(k)) s << "const rm_u64 in" << k << ", ";
        else s << "const " << S << "* __restrict__ in" << k << ", ";
    }
    for (int k = 0; k < nout; ++k) s << S << "* __restrict__ out" << k << ", ";
    s << "const rm_u64 n" << (rng_mask ? ", const rm_u64 rng_jm, const rm_u64 rng_jp" : "") << ") {\n";
    if (rng_mask) {
        s << "    __shared__ __attribute__((aligned(16))) double rm_tab[kBmLdsDoubles];\n";
        s << "    const BmTables tb = bm_stage_tables(rm_tab, threadIdx.x, " << t.block << ");\n";
    }
    s << "    const rm_u64 nvec_all = n / " << vec << ";\n";
    if (t.chunked) {
        // contiguous chunk per block (multiple of the per-iteration footprint), block-stride inside
        s << "    const rm_u64 per_iter = " << (t.block * U) << "ull;\n";
        s << "    rm_u64 chunk = (nvec_all + gridDim.x - 1) / gridDim.x;\n";
        s << "    chunk = (chunk + per_iter - 1) / per_iter * per_iter;\n";
        s << "    const rm_u64 begin = (rm_u64)blockIdx.x * chunk;\n";
        s << "    const rm_u64 nvec = begin + chunk < nvec_all ? begin + chunk : nvec_all;\n";
        s << "    const rm_u64 stride = " << t.block << "ull;\n";
        s << "    rm_u64 i = begin + threadIdx.x;\n";
    } else {
        s << "    const rm_u64 nvec = nvec_all;\n";
        s << "    const rm_u64 stride = (rm_u64)gridDim.x * " << t.block << ";\n";
        s << "    rm_u64 i = (rm_u64)blockIdx.x * " << t.block << " + threadIdx.x;\n";
    }
    for (int k = 0; k < nin; ++k)
        if (is_scalar(k)) s << "    const double s" << k << " = " << cast_in << "in" << k << "[0];\n";
    for (int k = 0; k < nin; ++k)  // the state u1 of this thread's first pair is drawn from (2 i + 1 steps into the tensor's stream)
        if (is_rng(k))
            s << "    rm_u64 x" << k << " = lcg_skip2(in" << k << ", " << (2 * t.block) << "ull * blockIdx.x, 2ull * threadIdx.x + 1ull);\n";
    auto load = [&](int k, const std::string& idx, const std::string& dst) {
        if (is_scalar(k)) return;
        if (is_rng(k)) {  // loads are emitted in index order, so the state simply follows them
            s << "        rm_v2 " << dst << ";\n        {\n";
            s << "            const double rad = bm_radius(x" << k << ", tb);\n            double sn, cs;\n";
            s << "            bm_sincos(lcg_step(x" << k << "), tb, &sn, &cs);\n";
            s << "            " << dst << " = rm_v2{rad * cs, rad * sn};\n";
            s << "            x" << k << " = rng_jm * x" << k << " + rng_jp;\n        }\n";
            return;
        }
        s << "        const " << vt << " " << dst << " = " << (t.nt_load ? "__builtin_nontemporal_load" : "*")
          << "((const " << vt << "*)in" << k << " + (" << idx << "));\n";
    };
    auto operand = [&](int k, const std::string& sfx, const char* comp) {
        if (is_scalar(k)) return "s" + std::to_string(k);
        return cast_in + "a" + std::to_string(k) + sfx + comp;
    };
    auto compute_store = [&](const std::string& sfx, const std::string& idx) {
        for (int k = 0; k < nout; ++k) s << "        " << vt << " r" << k << sfx << ";\n";
        if (vec >= 2) {
            static const char* comps[] = {".x", ".y", ".z", ".w"};
            for (int lane = 0; lane < vec; ++lane) {
                const char* c = comps[lane];
                s << "        { ";
                for (int k = 0; k < nout; ++k) s << "double q" << k << "; ";
                s << "rm_body(";
                for (int k = 0; k < nin; ++k) s << (k ? ", " : "") << operand(k, sfx, c);
                for (int k = 0; k < nout; ++k) s << ", q" << k;
                s << "); ";
                for (int k = 0; k < nout; ++k) s << "r" << k << sfx << c << " = " << cast_out << "q" << k << "; ";
                s << "}\n";
            }
        } else if (f32) {
            s << "        { ";
            for (int k = 0; k < nout; ++k) s << "double q" << k << "; ";
            s << "rm_body(";
            for (int k = 0; k < nin; ++k) s << (k ? ", " : "") << operand(k, sfx, "");
            for (int k = 0; k < nout; ++k) s << ", q" << k;
            s << "); ";
            for (int k = 0; k < nout; ++k) s << "r" << k << sfx << " = (float)q" << k << "; ";
            s << "}\n";
        } else {
            s << "        rm_body(";
            for (int k = 0; k < nin; ++k) s << (k ? ", " : "") << operand(k, sfx, "");
            for (int k = 0; k < nout; ++k) s << ", r" << k << sfx;
            s << ");\n";
        }
        for (int k = 0; k < nout; ++k) {
            if (t.nt_store)
                s << "        __builtin_nontemporal_store(r" << k << sfx << ", (" << vt << "*)out" << k << " + (" << idx << "));\n";
            else
                s << "        *((" << vt << "*)out" << k << " + (" << idx << ")) = r" << k << sfx << ";\n";
        }
    };
    if (U > 1) {
        s << "    for (; i + " << (U - 1) << " * stride < nvec; i += " << U << " * stride) {\n";
        for (int u = 0; u < U; ++u)
            for (int k = 0; k < nin; ++k)
                load(k, "i + " + std::to_string(u) + " * stride", "a" + std::to_string(k) + "_" + std::to_string(u));
        for (int u = 0; u < U; ++u) compute_store("_" + std::to_string(u), "i + " + std::to_string(u) + " * stride");
        s << "    }\n";
    }
    s << "    for (; i < nvec; i += stride) {\n";
    for (int k = 0; k < nin; ++k) load(k, "i", "a" + std::to_string(k) + "_t");
    compute_store("_t", "i");
    s << "    }\n";
    if (vec == 2) {
        s << "    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {\n";
        s << "        ";
        for (int k = 0; k < nout; ++k) s << "double q" << k << "; ";
        s << "rm_body(";
        for (int k = 0; k < nin; ++k)
            s << (k ? ", " : "")
              << (is_scalar(k) ? "s" + std::to_string(k)
                  : is_rng(k)  ? "rm_rng_tail(in" + std::to_string(k) + ", n, tb)"
                               : cast_in + "in" + std::to_string(k) + "[n - 1]");
        for (int k = 0; k < nout; ++k) s << ", q" << k;
        s << ");\n";
        for (int k = 0; k < nout; ++k) s << "        out" << k << "[n - 1] = " << cast_out << "q" << k << ";\n";
        s << "    }\n";
    } else if (vec == 4) {  // up to three tail elements, one thread each
        s << "    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {\n";
        s << "        const rm_u64 t = nvec_all * 4 + threadIdx.x;\n        ";
        for (int k = 0; k < nout; ++k) s << "double q" << k << "; ";
        s << "rm_body(";
        for (int k = 0; k < nin; ++k)
            s << (k ? ", " : "") << (is_scalar(k) ? "s" + std::to_string(k) : cast_in + "in" + std::to_string(k) + "[t]");
        for (int k = 0; k < nout; ++k) s << ", q" << k;
        s << ");\n";
        for (int k = 0; k < nout; ++k) s << "        out" << k << "[t] = " << cast_out << "q" << k << ";\n";
        s << "    }\n";
    }
    s << "}\n\n";
}

static void emit_bcast_kernel(std::ostringstream& s, const ElementwiseProgram& p, const EwTuning& t, bool f32) {
    const int nin = p.n_inputs, nout = (int)p.outputs.size(), E = t.bcast_elems;
    const char* S = f32 ? "float" : "double";
    const char* cast_in = f32 ? "(double)" : "";
    const char* cast_out = f32 ? "(float)" : "";
    // params: v[0]=d0, v[1]=nchunks, v[2]=rank, v[3..10]=shape, v[11+8k .. ] = stride of input k
    s << "struct RmBcast { rm_u64 v[" << (11 + 8 * nin) << "]; };\n";
    s << "extern \"C\" __global__ void __launch_bounds__(" << t.bcast_block << ") rm_ew_bcast(";
    for (int k = 0; k < nin; ++k) s << "const " << S << "* __restrict__ in" << k << ", ";
    for (int k = 0; k < nout; ++k) s << S << "* __restrict__ out" << k << ", ";
    s << "const RmBcast p) {\n";
    s << "    const rm_u64 d0 = p.v[0], nchunks = p.v[1];\n";
    s << "    const int rank = (int)p.v[2];\n";
    s << "    const rm_u64 blk = blockIdx.x + (rm_u64)gridDim.x * blockIdx.y;\n";
    s << "    const rm_u64 chunk = blk % nchunks;\n";
    s << "    const rm_u64 outer = blk / nchunks;\n";
    for (int k = 0; k < nin; ++k) s << "    rm_u64 off" << k << " = 0;\n";
    s << "    rm_u64 rem = outer;\n";
    s << "    for (int d = 1; d < rank; ++d) {\n";
    s << "        const rm_u64 ext = p.v[3 + d];\n";
    s << "        const rm_u64 c = rem % ext;\n";
    s << "        rem /= ext;\n";
    for (int k = 0; k < nin; ++k) s << "        off" << k << " += c * p.v[" << (11 + 8 * k) << " + d];\n";
    s << "    }\n";
    s << "    if (rem != 0) return;  // padding blocks of the 2-D grid\n";
    s << "    const rm_u64 obase = outer * d0;\n";
    s << "    const rm_u64 i0 = chunk * " << (t.bcast_block * E) << "ull + threadIdx.x;\n";
    for (int e = 0; e < E; ++e) {
        s << "    {\n        const rm_u64 i = i0 + " << (e * t.bcast_block) << "ull;\n        if (i < d0) {\n";
        for (int k = 0; k < nout; ++k) s << "            double q" << k << ";\n";
        s << "            rm_body(";
        for (int k = 0; k < nin; ++k) s << (k ? ", " : "") << cast_in << "in" << k << "[off" << k << " + i * p.v[" << (11 + 8 * k) << "]]";
        for (int k = 0; k < nout; ++k) s << ", q" << k;
        s << ");\n";
        for (int k = 0; k < nout; ++k) s << "            out" << k << "[obase + i] = " << cast_out << "q" << k << ";\n";
        s << "        }\n    }\n";
    }
    s << "}\n\n";
    // short dim 0 with many outer indices (a 32 x N matrix and a row or column operand): threads over the FLAT output, each decoding
    // its coordinates with 32-bit arithmetic (ew_kernels.hip: k_bcast2_flat) - the kernel above would run 32 lanes per block
    s << "extern \"C\" __global__ void __launch_bounds__(256) rm_ew_bcast_flat(";
    for (int k = 0; k < nin; ++k) s << "const " << S << "* __restrict__ in" << k << ", ";
    for (int k = 0; k < nout; ++k) s << S << "* __restrict__ out" << k << ", ";
    s << "const RmBcast p, const unsigned n) {\n";
    s << "    const unsigned d0 = (unsigned)p.v[0], stride = gridDim.x * 256u;\n";
    s << "    const int rank = (int)p.v[2];\n";
    s << "    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += stride) {\n";
    s << "        unsigned rem = idx / d0;\n        const unsigned i = idx - rem * d0;\n";
    for (int k = 0; k < nin; ++k) s << "        rm_u64 off" << k << " = (rm_u64)i * p.v[" << (11 + 8 * k) << "];\n";
    s << "        for (int d = 1; d < rank; ++d) {\n            const unsigned ext = (unsigned)p.v[3 + d];\n"
         "            const unsigned q = rem / ext, c = rem - q * ext;\n            rem = q;\n";
    for (int k = 0; k < nin; ++k) s << "            off" << k << " += (rm_u64)c * p.v[" << (11 + 8 * k) << " + d];\n";
    s << "        }\n";
    for (int k = 0; k < nout; ++k) s << "        double q" << k << ";\n";
    s << "        rm_body(";
    for (int k = 0; k < nin; ++k) s << (k ? ", " : "") << cast_in << "in" << k << "[off" << k << "]";
    for (int k = 0; k < nout; ++k) s << ", q" << k;
    s << ");\n";
    for (int k = 0; k < nout; ++k) s << "        out" << k << "[idx] = " << cast_out << "q" << k << ";\n";
    s << "    }\n}\n\n";
}

std::string generate_elementwise_source(const ElementwiseProgram& p, const EwTuning& t, unsigned mask, bool f32, unsigned rng_mask) {
    std::ostringstream s;
    if (rng_mask) {  // the streaming kernel alone (16-byte vectors, f64): every other shape of request sees materialised operands
        s << "// generated by librmhip from a fused elementwise plan (" << p.lets.size() << " ops, " << p.n_inputs << " inputs, lazy random_normal mask "
          << rng_mask << ")\n";
        s << kSkelCommon << "\n" << kRngTables << "\n" << kSkelRng << "\n";
        s << "typedef double rm_v2 __attribute__((ext_vector_type(2)));\n";
        // element n - 1 of an odd-length tensor: z0 of pair (n - 1) / 2, whose u1 comes from the state n steps in
        s << "__device__ __noinline__ double rm_rng_tail(rm_u64 state, rm_u64 n, const BmTables& tb) {\n"
             "    rm_u64 m, q;\n    lcg_jump(n, &m, &q);\n    const rm_u64 x1 = m * state + q;\n"
             "    const double rad = bm_radius(x1, tb);\n    double sn, cs;\n    bm_sincos(lcg_step(x1), tb, &sn, &cs);\n    return rad * cs;\n}\n";
        s << "\n" << body_function(p);
        emit_fast_kernel(s, p, t, 2, mask, "rm_ew_fast", false, rng_mask);
        return s.str();
    }
    s << "// generated by librmhip from a fused elementwise plan (" << p.lets.size() << " ops, " << p.n_inputs
      << " inputs, " << p.outputs.size() << " outputs)\n";
    if (f32) s << "#define RM_RESULT_F32 1\n";  // results are stored as f32: skel_common.h's short sin / cos
    s << kSkelCommon << "\n";
    s << "typedef double rm_v2 __attribute__((ext_vector_type(2)));\n";
    if (f32) s << "typedef float rm_v4f __attribute__((ext_vector_type(4)));\n";
    s << "\n" << body_function(p);
    emit_fast_kernel(s, p, t, f32 ? 4 : 2, mask, "rm_ew_fast", f32);
    emit_fast_kernel(s, p, t, 1, mask, "rm_ew_fast1", f32);
    emit_bcast_kernel(s, p, t, f32);
    return s.str();
}

std::string generate_reduction_source(const ReductionProgram& p, bool f32) {
    std::ostringstream s;
    const int nin = p.n_inputs;
    const std::string S = f32 ? "float" : "double";
    const std::string cast_in = f32 ? "(double)" : "";
    s << "// generated by librmhip from a fused reduction plan (" << nin << " inputs, axis " << p.axis << ")\n";
    if (f32) s << "#define RM_RESULT_F32 1\n";
    s << kSkelCommon << "\n" << kSkelReduce << "\n";
    s << "struct RmVal {\n";
    for (int k = 0; k < nin; ++k) s << "    const " << S << "* __restrict__ in" << k << ";\n    rm_u64 m" << k << ";\n";
    s << "    __device__ __forceinline__ double operator()(rm_u64 idx) const {\n";
    for (int k = 0; k < nin; ++k) s << "        const double v" << k << " = " << cast_in << "in" << k << "[idx * m" << k << "];\n";
    s << "        return " << emit_expr_f64(p.val) << ";\n    }\n};\n\n";
    // the same value for two adjacent elements of a contiguous slice (kernel A over 16-byte vectors, skel_reduce.h:
    // rm_reduce_contig_v2): full-size inputs are read as pairs, 1-element inputs broadcast
    s << "typedef " << S << " rm_sv2 __attribute__((ext_vector_type(2)));\n";
    s << "struct RmVal2 {\n";
    for (int k = 0; k < nin; ++k) s << "    const " << S << "* __restrict__ in" << k << ";\n    rm_u64 m" << k << ";\n";
    s << "    __device__ __forceinline__ rm_rv2 operator()(rm_u64 i2) const {\n";
    for (int k = 0; k < nin; ++k)
        s << "        rm_sv2 p" << k << ";\n        if (m" << k << ") p" << k << " = __builtin_nontemporal_load((const rm_sv2*)in" << k
          << " + i2);\n        else p" << k << " = rm_sv2{in" << k << "[0], in" << k << "[0]};\n";
    s << "        rm_rv2 r;\n";
    for (const char* lane : {"x", "y"}) {
        s << "        {\n";
        for (int k = 0; k < nin; ++k) s << "            const double v" << k << " = " << cast_in << "p" << k << "." << lane << ";\n";
        s << "            r." << lane << " = " << emit_expr_f64(p.val) << ";\n        }\n";
    }
    s << "        return r;\n    }\n};\n\n";
    auto args = [&]() {
        std::string a;
        for (int k = 0; k < nin; ++k)
            a += "const " + S + "* __restrict__ in" + std::to_string(k) + ", const rm_u64 m" + std::to_string(k) + ", ";
        return a;
    };
    auto init = [&]() {
        std::string a = "    RmVal f;\n";
        for (int k = 0; k < nin; ++k)
            a += "    f.in" + std::to_string(k) + " = in" + std::to_string(k) + "; f.m" + std::to_string(k) + " = m" +
                 std::to_string(k) + ";\n";
        return a;
    };
    s << "extern \"C\" __global__ void __launch_bounds__(RM_ABLOCK) rm_red_contig(" << args()
      << "const rm_u64 red, const rm_u64 nslices, const rm_u64 nsplit, double* part_v, double* part_nan) {\n"
      << init() << "    rm_reduce_contig<RM_RSUM>(f, red, nslices, nsplit, part_v, part_nan);\n}\n\n";
    s << "extern \"C\" __global__ void __launch_bounds__(RM_ABLOCK) rm_red_contig2(" << args()
      << "const rm_u64 red, const rm_u64 nslices, const rm_u64 nsplit, double* part_v, double* part_nan) {\n"
      << "    RmVal2 f2;\n";
    for (int k = 0; k < nin; ++k) s << "    f2.in" << k << " = in" << k << "; f2.m" << k << " = m" << k << ";\n";
    s << "    rm_reduce_contig_v2<RM_RSUM>(f2, red, nslices, nsplit, part_v, part_nan);\n}\n\n";
    s << "extern \"C\" __global__ void __launch_bounds__(RM_RBLOCK) rm_red_strided(" << args()
      << "const rm_u64 pre, const rm_u64 red, const rm_u64 nsplit, const int tx, double* part_v, double* part_nan) {\n"
      << init() << "    rm_reduce_strided<RM_RSUM>(f, pre, red, nsplit, tx, part_v, part_nan);\n}\n\n";
    s << "extern \"C\" __global__ void __launch_bounds__(RM_RBLOCK) rm_red_strided2(" << args()
      << "const rm_u64 pre, const rm_u64 red, const rm_u64 nsplit, const unsigned win, double* part_v, double* part_nan) {\n"
      << "    RmVal2 f2;\n";
    for (int k = 0; k < nin; ++k) s << "    f2.in" << k << " = in" << k << "; f2.m" << k << " = m" << k << ";\n";
    s << "    rm_reduce_strided_v2<RM_RSUM, 4>(f2, pre, red, nsplit, win, part_v, part_nan);\n}\n\n";
    s << "extern \"C\" __global__ void __launch_bounds__(RM_RBLOCK) rm_red_final(const double* part_v, const double* "
         "part_nan, const rm_u64 nslices, const rm_u64 nsplit, const rm_u64 red, const int mean, const int omitnan, "
         "const double scale, double* out) {\n"
         "    rm_reduce_finalize<RM_RSUM>(part_v, part_nan, nslices, nsplit, red, mean, omitnan, scale, out);\n}\n\n";
    s << "extern \"C\" __global__ void __launch_bounds__(RM_RBLOCK) rm_red_final_flat(const double* part_v, const double* "
         "part_nan, const rm_u64 nslices, const rm_u64 nsplit, const rm_u64 red, const int mean, const int omitnan, "
         "const double scale, double* out) {\n"
         "    rm_reduce_finalize_flat<RM_RSUM>(part_v, part_nan, nslices, nsplit, red, mean, omitnan, scale, out);\n}\n";
    return s.str();
}

// ---- hipRTC ------------------------------------------------------------------------------------

static std::string cache_dir() {
    const char* d = std::getenv("RMHIP_CACHE_DIR");
    if (d && *d) return d;
    return "";
}

int compile_to_code_object(const std::string& source, std::vector<char>* code) {
    const std::string dir = cache_dir();
    // file name = two independent 64-bit hashes + the source length: a collision would have to match all three
    char namebuf[96];
    std::snprintf(namebuf, sizeof namebuf, "%016llx-%016llx-%zx.hsaco", (unsigned long long)fnv1a(source + "|gfx950|v2"),
                  (unsigned long long)fnv1a("rmhip:" + source), source.size());
    if (!dir.empty()) {  // persisted code objects keyed by source hash + arch
        std::ifstream f(dir + "/" + namebuf, std::ios::binary);
        if (f) {
            code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            if (!code->empty()) return RMHIP_OK;
        }
    }
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, source.c_str(), "rmhip_fused.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return fail(RMHIP_ERR_COMPILE, "hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    // -ffp-contract=off: the CPU path rounds after every op (no FMA contraction in Rust), so
    // `sin(A).*B + C` must be a multiply then an add.
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    r = hiprtcCompileProgram(prog, 4, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        if (log.size() > 1500) log.resize(1500);
        return fail(RMHIP_ERR_COMPILE, "hipRTC compile failed: %s\n%s", hiprtcGetErrorString(r), log.c_str());
    }
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    code->resize(sz);
    hiprtcGetCode(prog, code->data());
    hiprtcDestroyProgram(&prog);
    if (!dir.empty()) {
        // one process per GPU compiles the same shaders at start-up: write a private temporary and rename() it into
        // place, so a reader sees either no file or the complete one
        ::mkdir(dir.c_str(), 0755);
        char tmpbuf[160];
        std::snprintf(tmpbuf, sizeof tmpbuf, "%s.tmp.%ld", namebuf, (long)::getpid());
        const std::string tmp = dir + "/" + tmpbuf;
        bool ok = false;
        {
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            if (f) {
                f.write(code->data(), (std::streamsize)code->size());
                f.flush();
                ok = f.good();
            }
        }
        if (!ok || std::rename(tmp.c_str(), (dir + "/" + namebuf).c_str()) != 0) std::remove(tmp.c_str());
    }
    return RMHIP_OK;
}

static int load_function(hipModule_t m, const char* name, hipFunction_t* fn) {
    hipError_t e = hipModuleGetFunction(fn, m, name);
    if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "hipModuleGetFunction(%s): %s", name, hipGetErrorString(e));
    return RMHIP_OK;
}

int get_elementwise_kernel(Context* c, const ElementwiseProgram& p, unsigned mask, bool f32,
                           std::shared_ptr<FusedKernel>* out, unsigned rng_mask) {
    EwTuning t = EwTuning::from_env();
    // f32 storage: 4-byte accesses in the broadcast kernel, so twice the elements per thread keep the same bytes in flight
    // (interleaved A/B at 8192^2, `A - row`: 4051 -> 4244 GB/s)
    if (f32 && !std::getenv("RMHIP_EW_BCAST_ELEMS")) t.bcast_elems = 8;
    char tun[96];
    std::snprintf(tun, sizeof tun, "|u%d|b%d|bb%dx%d|nt%d%d|c%d|m%x", t.unroll, t.block, t.bcast_block, t.bcast_elems, t.nt_load, t.nt_store, t.chunked, mask);
    const std::string key_text = p.canonical + tun + (f32 ? "|f32" : "") + (rng_mask ? "|rng" + std::to_string(rng_mask) : "");
    const uint64_t key = fnv1a(key_text);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->kernel_cache.find(key);
        if (it != c->kernel_cache.end() && it->second->key_text == key_text) {  // a 64-bit hash alone could run the wrong kernel
            c->tel.cache_hits++;
            *out = it->second;
            return RMHIP_OK;
        }
    }
    c->tel.cache_misses++;
    std::vector<char> code;
    RMHIP_TRY(compile_to_code_object(generate_elementwise_source(p, t, mask, f32, rng_mask), &code));
    auto k = std::make_shared<FusedKernel>();
    k->key_text = key_text;
    k->tuning = t;
    k->n_inputs = p.n_inputs;
    k->n_outputs = (int)p.outputs.size();
    RMHIP_HIP_CHECK(hipModuleLoadData(&k->module, code.data()));
    RMHIP_TRY(load_function(k->module, "rm_ew_fast", &k->fn_fast));
    if (!rng_mask) {  // (a kernel with lazy random_normal operands is the streaming form alone)
        RMHIP_TRY(load_function(k->module, "rm_ew_fast1", &k->fn_fast1));
        RMHIP_TRY(load_function(k->module, "rm_ew_bcast", &k->fn_bcast));
        RMHIP_TRY(load_function(k->module, "rm_ew_bcast_flat", &k->fn_bcast_flat));
    }
    std::lock_guard<std::mutex> lk(c->mu);
    c->kernel_cache[key] = k;
    *out = k;
    return RMHIP_OK;
}

int get_reduction_kernel(Context* c, const ReductionProgram& p, bool f32, std::shared_ptr<FusedKernel>* out) {
    const std::string key_text = p.canonical + (f32 ? "|f32" : "") + "|red";
    const uint64_t key = fnv1a(key_text);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->kernel_cache.find(key);
        if (it != c->kernel_cache.end() && it->second->key_text == key_text) {
            c->tel.cache_hits++;
            *out = it->second;
            return RMHIP_OK;
        }
    }
    c->tel.cache_misses++;
    std::vector<char> code;
    RMHIP_TRY(compile_to_code_object(generate_reduction_source(p, f32), &code));
    auto k = std::make_shared<FusedKernel>();
    k->key_text = key_text;
    k->n_inputs = p.n_inputs;
    k->n_outputs = 1;
    RMHIP_HIP_CHECK(hipModuleLoadData(&k->module, code.data()));
    RMHIP_TRY(load_function(k->module, "rm_red_contig", &k->fn_contig));
    RMHIP_TRY(load_function(k->module, "rm_red_contig2", &k->fn_contig2));
    RMHIP_TRY(load_function(k->module, "rm_red_strided", &k->fn_strided));
    RMHIP_TRY(load_function(k->module, "rm_red_strided2", &k->fn_strided2));
    RMHIP_TRY(load_function(k->module, "rm_red_final", &k->fn_final));
    RMHIP_TRY(load_function(k->module, "rm_red_final_flat", &k->fn_final_flat));
    std::lock_guard<std::mutex> lk(c->mu);
    c->kernel_cache[key] = k;
    *out = k;
    return RMHIP_OK;
}

}  // namespace rmhip
