// rmhip_ops.cpp -- the operator half of the C ABI (include/rmhip.h): fused elementwise / fused
// reduction dispatch, per-op kernels, reductions, matmul, lu, mldivide, rng.
// Host-side logic mirrors the provider duties of the reference's backends:
//   broadcast shape/stride preparation  backend/wgpu/provider/ops/elementwise.rs:1655-1697
//   reduction geometry handed in by     crates/runmat-vm/src/accel/fusion.rs:540-915 (reduce_len, num_slices)
//   output shapes of the plain reducers crates/runmat-accelerate/src/simple_provider.rs:6728-6806
#include <mutex>
#include <unordered_map>
#include <memory>
#include <string>
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstring>

#include "codegen.h"
#include <limits>

#include "common.h"
#include "host_shape.h"
#include "reduce_plan.h"
#include "wgsl_front.h"

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c);                                             \
    NarrowScope _ns(c)

namespace {

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Copy `src` (rows x cols, dense) into the padded workspace and factor it; when the persistent panel kernels
// report that their workgroups were not co-resident the copy is refreshed and factored conservatively.
// pad_n > rows (square systems on the solve path): factor [A 0; 0 I] of order pad_n instead - see lu_pad_rows
static int lu_copy_and_factor(Context* c, const double* src, size_t rows, size_t cols, double* work, size_t ldw, int* perm,
                              int* info, bool solve_path = false, size_t pad_n = 0) {
    // solve_path: the caller only needs SOME stable factorisation (mldivide / linsolve / mrdivide: the pivots never leave the provider),
    // so the first attempt restricts pivoting to each panel's top block and checks the multipliers (lu.hip, k_rp_below); when that
    // check fails the copy is refreshed and factored with the reference's grid-wide rule.
    int mode = solve_path ? 1 : 0;
    for (int attempt = 0; attempt < 4; ++attempt) {  // [solve path ->] one-XCD panels -> spread panels -> one launch per column
        // (Tried: only the first 1024 columns here and the rest on the factorisation's update stream, under the first panel - the 0.8 ms
        // of a 2 GiB copy off the critical path on paper; n = 16384 98.9 vs 98.6-99.0 ms, n = 8192 34.6 vs 34.6: nothing.)
        if (rows && cols) {
            hipError_t e = hipMemcpy2DAsync(work, ldw * sizeof(double), src, rows * sizeof(double), rows * sizeof(double), cols,
                                            hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "lu copy: %s", hipGetErrorString(e));
        }
        const bool padded = pad_n > rows && rows == cols;
        if (padded) RMHIP_TRY(lu_pad_identity_device(c, work, ldw, rows, pad_n));
        const int rc = lu_factor_device(c, work, padded ? pad_n : rows, padded ? pad_n : cols, ldw, perm, info, nullptr, mode);
        if (rc == RMHIP_LU_GROWTH) {
            mode = 0;
            c->lu_growth_fallbacks++;
            // telemetry.solve_fallbacks names it when a multiplier actually exceeded the bound; a pivot at the singular cut-off inside a
            // top block also lands here (the grid-wide rule has to confirm it), and is then reported as what it turns out to be
            if (!(c->lu_last_growth <= c->lu_tau) || std::getenv("RMHIP_LU_TEST_GROWTH")) c->record_solve_fallback("lu:pivot_growth");
            continue;
        }
        if (rc == RMHIP_OK && mode == 1 && c->lu_last_fast) c->lu_fast_count++;  // RMHIP_LU_FAST=0 / conservative panels: the grid-wide rule ran
        if (rc != RMHIP_LU_RETRY) return rc;
    }
    return fail(RMHIP_ERR_HIP, "lu: factorisation failed on every panel path");
}

size_t lu_padded_ld(size_t rows) { return rows >= 256 ? ((rows + 1) & ~(size_t)1) + 32 : ((rows + 1) & ~(size_t)1); }

// Order the solves factor at.  The blocked driver's trailing updates are whole 128 x 128 x 16 tiles only when n is a multiple of
// 128; otherwise EVERY update of the factorisation runs a guarded kernel (n = 10000: 37.0 ms against 31.7 at 10240, n = 13001: 55.6
// against 47.2 at 13056).  The solves therefore factor [A 0; 0 I] at the next multiple of 128 - 2-4 % more flops, all of them on
// the fast kernels - and drop the padded unknowns (zero).  `lu` itself returns factors of the order it was given.
// RMHIP_LU_PAD=0 disables.
static size_t lu_pad_rows(size_t n) {
    const char* v = std::getenv("RMHIP_LU_PAD");  // read per call: the tests compare both forms
    static const size_t min_n = std::getenv("RMHIP_LU_PAD_MIN") ? (size_t)std::atol(std::getenv("RMHIP_LU_PAD_MIN")) : 2048;  // dev knob
    if ((v && *v == '0') || n < min_n || n % 128 == 0) return n;
    const size_t np = (n + 127) / 128 * 128;
    return np <= 65535 ? np : n;
}
// X (n x nrhs, ld n) = A^-1 B from factors of order np >= n (np > n: the padded system)
static int lu_solve_padded(Context* c, const double* LU, size_t n, size_t np, size_t ldw, const int* perm, const double* B, size_t nrhs, double* X) {
    if (nrhs == 0) return RMHIP_OK;  // (a zero-height hipMemcpy2DAsync is hipErrorInvalidValue)
    if (np == n) return lu_solve_device(c, LU, n, ldw, perm, B, nrhs, n, X, n);
    std::shared_ptr<Allocation> bp, xp;
    RMHIP_TRY(c->alloc_device(np * nrhs, &bp));
    RMHIP_TRY(c->alloc_device(np * nrhs, &xp));
    RMHIP_HIP_CHECK(hipMemsetAsync(bp->ptr, 0, sizeof(double) * np * nrhs, c->stream));
    RMHIP_HIP_CHECK(hipMemcpy2DAsync(bp->ptr, np * sizeof(double), B, n * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDeviceToDevice, c->stream));
    RMHIP_TRY(lu_solve_device(c, LU, np, ldw, perm, bp->ptr, nrhs, np, xp->ptr, np));
    RMHIP_HIP_CHECK(hipMemcpy2DAsync(X, n * sizeof(double), xp->ptr, np * sizeof(double), n * sizeof(double), nrhs, hipMemcpyDeviceToDevice, c->stream));
    return RMHIP_OK;
}

// Precision-32 contexts: fetch an operand for a kernel that has an f32-storage variant.  `*native` stays true while
// every operand so far is plain f32 storage; otherwise the caller falls back to widened f64 copies (Context::get).
int get_operand(Context* c, rmhip_buf id, Buffer* out, bool* native) {
    if (*native) {
        RMHIP_TRY(c->get_raw(id, out));
        if (out->dtype == DT_F32 && out->lazy()) {  // materialise the view once, as f32
            RMHIP_TRY(c->settle_view(id));
            RMHIP_TRY(c->get_raw(id, out));
        }
        if (out->dtype == DT_F32 && !out->lazy()) return RMHIP_OK;
        *native = false;
    }
    return c->get(id, out);
}

// Operand of a broadcasting elementwise launch (rmhip_binary, rmhip_fused_elementwise): as get_operand, but a repmat view
// stays a view - the launch indexes its base with stride 0 (host_shape.h refined_strides).
int get_bcast_operand(Context* c, rmhip_buf id, Buffer* out, bool* native, bool keep_rng = false) {
    RMHIP_TRY(c->get_raw(id, out, keep_rng));
    if (out->rng_lazy) return RMHIP_OK;  // (f64 contexts only; consumed in registers by the streaming kernel)
    if (out->tview) {
        RMHIP_TRY(c->settle_view(id));
        RMHIP_TRY(c->get_raw(id, out));
    }
    if (*native) {
        if (out->dtype == DT_F32) return RMHIP_OK;
        *native = false;
    }
    if (out->dtype == DT_F64) return RMHIP_OK;
    return c->get(id, out);  // f32 storage read by the f64 variant: tiled (if a view) and widened
}

// refined_strides over Buffers; an operand whose tiling conflicts with another view's is materialised and the preparation repeated
int bcast_prepare(Context* c, const rmhip_buf* ids, std::vector<Buffer>* in, bool f32, const size_t* out_shape, size_t rank,
                  std::vector<uint64_t>* rshape, std::vector<std::vector<uint64_t>>* strides, const char* what) {
    for (size_t attempt = 0; attempt <= in->size(); ++attempt) {
        std::vector<OperandDims> ops(in->size());
        for (size_t k = 0; k < in->size(); ++k) {
            ops[k].shape = (*in)[k].shape;
            ops[k].base = (*in)[k].rep_base;
        }
        size_t bad = 0;
        const int rc = refined_strides(ops, out_shape, rank, rshape, strides, &bad);
        if (rc == 0) return RMHIP_OK;
        if (rc == 1) return fail(RMHIP_ERR_SHAPE, "%s: input %zu does not broadcast to the output shape", what, bad);
        RMHIP_TRY(c->settle_view(ids[bad]));
        if (f32) RMHIP_TRY(c->get_raw(ids[bad], &(*in)[bad]));
        else RMHIP_TRY(c->get(ids[bad], &(*in)[bad]));
    }
    return fail(RMHIP_ERR_UNSUPPORTED, "%s: could not reconcile the operands' tilings", what);
}

// Precision 32: may this product run on the f32 matrix cores (sgemm.hip)?  RMHIP_F32_MATMUL=f64 keeps the widen ->
// dgemm -> round-once path (the CPU's `single` result exactly), which also serves k == 0.  Read per call: tests flip
// the variable.
bool f32_gemm_eligible(const Context* c, size_t m, size_t n, size_t k) {
    (void)m;
    (void)n;
    if (c->precision != 32 || k == 0) return false;
    const char* mode = std::getenv("RMHIP_F32_MATMUL");
    return !(mode && std::strcmp(mode, "f64") == 0);
}

std::vector<size_t> normalize_matrix_shape(const std::vector<size_t>& s) {
    if (s.empty()) return {1, 1};
    if (s.size() == 1) return {s[0], 1};
    return s;
}

}  // namespace

// ---- parsed-request cache ---------------------------------------------------------------------------
// The planner re-sends the same shader text for every execution of a fusion group; lexing and parsing it
// (2.5-4 KB) cost 7-17 us per call (measured: 13 us for a 3-op request, 23 us for the 14-op chain, against
// 6 us for a per-op call), i.e. more than the kernel at 1024^2.  Parsed programs are immutable and shared.
namespace {
template <typename Prog>
struct ParseCache {
    std::mutex mu;
    std::unordered_map<std::string, std::shared_ptr<const Prog>> map;
    template <typename ParseFn>
    std::shared_ptr<const Prog> get(const char* shader, ParseFn parse, std::string* err) {
        std::string key(shader);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = map.find(key);
            if (it != map.end()) return it->second;
        }
        auto prog = std::make_shared<Prog>();
        if (!parse(key, prog.get(), err)) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (map.size() > 4096) map.clear();  // unbounded growth guard; entries are cheap to rebuild
        map.emplace(std::move(key), prog);
        return prog;
    }
};
ParseCache<ElementwiseProgram> g_ew_parse_cache;
ParseCache<ReductionProgram> g_red_parse_cache;
}  // namespace

extern "C" {

int rmhip_wgsl_translate(const char* shader, int kind, char* out, size_t cap, size_t* needed) {
    if (!shader) return fail(RMHIP_ERR_INVALID, "null shader");
    std::string err, src;
    if (kind == 0 || (kind & 0x100)) {
        ElementwiseProgram p;
        if (!parse_elementwise_wgsl(shader, &p, &err)) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
        if ((kind & 0x100) && p.f32) return fail(RMHIP_ERR_UNSUPPORTED, "lazy random_normal operands are f64 only");
        src = generate_elementwise_source(p, EwTuning::from_env(), 0u, p.f32, (kind & 0x100) ? (unsigned)(kind & 0xff) : 0u);
    } else {
        ReductionProgram p;
        if (!parse_reduction_wgsl(shader, &p, &err)) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
        src = generate_reduction_source(p, p.f32);
    }
    if (needed) *needed = src.size() + 1;
    if (out && cap) {
        const size_t n = std::min(cap - 1, src.size());
        std::memcpy(out, src.data(), n);
        out[n] = '\0';
    }
    return RMHIP_OK;
}

int rmhip_wgsl_compile_check(const char* shader, int kind) {
    if (!shader) return fail(RMHIP_ERR_INVALID, "null shader");
    std::string err, src;
    if (kind == 0 || (kind & 0x100)) {
        ElementwiseProgram p;
        if (!parse_elementwise_wgsl(shader, &p, &err)) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
        if ((kind & 0x100) && p.f32) return fail(RMHIP_ERR_UNSUPPORTED, "lazy random_normal operands are f64 only");
        src = generate_elementwise_source(p, EwTuning::from_env(), 0u, p.f32, (kind & 0x100) ? (unsigned)(kind & 0xff) : 0u);
    } else {
        ReductionProgram p;
        if (!parse_reduction_wgsl(shader, &p, &err)) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
        src = generate_reduction_source(p, p.f32);
    }
    std::vector<char> code;
    return compile_to_code_object(src, &code);
}

int rmhip_fused_elementwise(rmhip_ctx* ctx, const char* shader, const rmhip_buf* inputs, size_t n_in,
                            const size_t* out_shape, size_t rank, size_t len, size_t n_out, rmhip_buf* out_ids) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.fused_elementwise_count, &c->tel.fused_elementwise_ns);
    if (!shader || !inputs || !out_ids || (rank && !out_shape)) return fail(RMHIP_ERR_INVALID, "fused_elementwise: null argument");
    if (n_in == 0) return fail(RMHIP_ERR_INVALID, "fused_elementwise: no inputs");  // elementwise.rs:1574
    if (n_in > 24) return fail(RMHIP_ERR_UNSUPPORTED, "fused_elementwise: more than 24 inputs");
    if (rank > 8 + 8) return fail(RMHIP_ERR_UNSUPPORTED, "fused_elementwise: rank too large");
    if (shape_numel(out_shape, rank) != len) return fail(RMHIP_ERR_SHAPE, "fused_elementwise: len %zu != prod(output_shape)", len);
    if (len == 0) return fail(RMHIP_ERR_UNSUPPORTED, "fusion: zero-length execution not supported");  // fusion_exec.rs:273
    std::string err;
    const std::shared_ptr<const ElementwiseProgram> prog_ptr = g_ew_parse_cache.get(shader, parse_elementwise_wgsl, &err);
    if (!prog_ptr) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
    const ElementwiseProgram& prog = *prog_ptr;
    if ((size_t)prog.n_inputs != n_in) return fail(RMHIP_ERR_INVALID, "fused_elementwise: shader binds %d inputs, got %zu", prog.n_inputs, n_in);
    if (prog.outputs.size() != n_out) return fail(RMHIP_ERR_INVALID, "fused_elementwise: shader writes %zu outputs, caller expects %zu", prog.outputs.size(), n_out);

    if (prog.f32 != (c->precision == 32))
        return fail(RMHIP_ERR_COMPILE, "%s shader handed to an %s provider (precision() is %s)", prog.f32 ? "f32" : "f64",
                    c->precision == 32 ? "F32" : "F64", c->precision == 32 ? "F32" : "F64");
    std::vector<Buffer> in(n_in);
    std::vector<uint64_t> oshape(out_shape, out_shape + rank);
    std::vector<std::vector<uint64_t>> strides(n_in);
    // Lazy random_normal operands (Buffer::rng_lazy) stay lazy only for the streaming kernel over 16-byte vectors: every operand is
    // either a plain full-size tensor of the output's shape or a 1-element tensor.  Any other request materialises them first.
    unsigned rng_mask = 0;
    if (c->precision != 32 && c->n_rng_lazy != 0) {  // (no lazy record alive in this context: nothing to look for)
        bool any = false, eligible = len >= 2;
        auto extents = [](const std::vector<size_t>& sh) {
            std::vector<size_t> e;
            for (size_t d : sh)
                if (d != 1) e.push_back(d);
            return e;
        };
        const std::vector<size_t> want = extents(std::vector<size_t>(out_shape, out_shape + rank));
        for (size_t k = 0; k < n_in; ++k) {
            RMHIP_TRY(c->get_raw(inputs[k], &in[k], /*keep_rng=*/true));
            any |= in[k].rng_lazy;
        }
        if (any) {
            for (size_t k = 0; k < n_in && eligible; ++k) {
                const Buffer& b = in[k];
                const bool full = b.numel == len && extents(b.shape) == want;
                if (b.rng_lazy) eligible = full;
                else eligible = !b.lazy() && b.dtype == DT_F64 && (b.numel == 1 || (full && aligned16(b.data())));
            }
            for (size_t k = 0; k < n_in; ++k) {
                if (!in[k].rng_lazy) continue;
                if (eligible) rng_mask |= 1u << k;
                else RMHIP_TRY(c->settle_rng(inputs[k]));
            }
        }
        for (auto& b : in) b = Buffer();
    }
    // f32 storage is read and written in place by the f32 variant of the generated kernel; a mixed operand list
    // (externally wrapped f64 memory, transpose views) runs the f64 variant on widened copies
    bool f32 = c->precision == 32;
    size_t tried = 0;  // when the native attempt gives up at operand tried - 1, that one already holds its f64 copy
    // repmat views are read in place (stride 0 over their base)
    for (; tried < n_in && f32; ++tried) RMHIP_TRY(get_bcast_operand(c, inputs[tried], &in[tried], &f32));
    for (size_t k = 0; k < n_in; ++k)
        if (!f32 && (k + 1 != tried || in[k].dtype == DT_F32)) {
            bool no = false;
            RMHIP_TRY(get_bcast_operand(c, inputs[k], &in[k], &no, (rng_mask >> k) & 1u));
        }
    RMHIP_TRY(bcast_prepare(c, inputs, &in, f32, out_shape, rank, &oshape, &strides, "fused_elementwise"));
    RMHIP_TRACEF("fused_elementwise: operands ready (f32 storage path %d)", (int)f32);
    collapse(&oshape, &strides);
    const size_t crank = oshape.size();
    if (crank > 8) return fail(RMHIP_ERR_UNSUPPORTED, "fused_elementwise: broadcast rank %zu > 8 after collapsing", crank);

    bool fast = crank == 1;
    unsigned mask = 0;
    if (fast)
        for (size_t k = 0; k < n_in; ++k)
            if (strides[k][0] == 0) mask |= 1u << k;
    if (!fast) mask = 0;
    if (rng_mask && !fast) {  // (not expected after the eligibility test above: materialise and run the request again)
        for (size_t k = 0; k < n_in; ++k)
            if ((rng_mask >> k) & 1u) RMHIP_TRY(c->settle_rng(inputs[k]));
        return rmhip_fused_elementwise(ctx, shader, inputs, n_in, out_shape, rank, len, n_out, out_ids);
    }

    std::shared_ptr<FusedKernel> kern;
    RMHIP_TRY(get_elementwise_kernel(c, prog, mask, f32, &kern, rng_mask));
    RMHIP_TRACEF("fused_elementwise: kernel ready (fast %d mask %x)", (int)fast, mask);

    std::vector<Buffer> outs(n_out);
    std::vector<rmhip_buf> ids(n_out, 0);
    for (size_t k = 0; k < n_out; ++k) {
        int rc = f32 ? c->new_buffer_f32(out_shape, rank, &ids[k], &outs[k]) : c->new_buffer(out_shape, rank, &ids[k], &outs[k]);
        if (rc != RMHIP_OK) {
            for (size_t j = 0; j < k; ++j) rmhip_free(ctx, ids[j]);
            return rc;
        }
    }

    std::vector<const double*> in_ptr(n_in);
    std::vector<double*> out_ptr(n_out);
    for (size_t k = 0; k < n_in; ++k) in_ptr[k] = in[k].data();
    for (size_t k = 0; k < n_out; ++k) out_ptr[k] = outs[k].data();
    std::vector<void*> args;
    std::vector<unsigned long long> rng_states(n_in, 0);
    for (size_t k = 0; k < n_in; ++k) {
        rng_states[k] = in[k].rng_state;
        if ((rng_mask >> k) & 1u) args.push_back(&rng_states[k]);  // the stream state the tensor was drawn at, by value
        else args.push_back(&in_ptr[k]);
    }
    for (size_t k = 0; k < n_out; ++k) args.push_back(&out_ptr[k]);
    unsigned long long rng_jm = 1, rng_jp = 0;

    const EwTuning& t = kern->tuning;
    hipError_t e;
    if (fast) {
        bool vec_ok = true;
        for (size_t k = 0; k < n_in; ++k)
            if (!((mask >> k) & 1u) && !((rng_mask >> k) & 1u) && !aligned16(in_ptr[k])) vec_ok = false;
        for (size_t k = 0; k < n_out; ++k)
            if (!aligned16(out_ptr[k])) vec_ok = false;
        unsigned long long n = len;
        args.push_back(&n);
        const size_t work = vec_ok ? len / (f32 ? 4 : 2) : len;
        int n_stream = 0;
        for (size_t k = 0; k < n_in; ++k) n_stream += ((mask >> k) & 1u) ? 0 : 1;
        const size_t per_block = (size_t)t.block * t.unroll_for(n_stream, program_is_heavy(prog));
        size_t want = (work + per_block - 1) / per_block;
        // a kernel that generates normals is VALU-bound and pays a skip-ahead + 10 KiB of table staging per block: one resident set of
        // blocks (2048 threads per CU) walks the tensor instead of blocks_per_cu waves of them
        const size_t cap = rng_mask ? (size_t)c->num_cus * std::max(1, 2048 / t.block) : (size_t)c->num_cus * t.blocks_per_cu;
        if (want < 1) want = 1;
        const unsigned grid = (unsigned)std::min(want, cap);
        if (rng_mask) {
            if (!vec_ok) {  // (outputs are fresh allocations and the inputs were tested above)
                for (size_t k = 0; k < n_out; ++k) rmhip_free(ctx, ids[k]);
                return fail(RMHIP_ERR_HIP, "fused_elementwise: unaligned buffer beside a lazy random_normal operand");
            }
            // one 16-byte vector is one Box-Muller pair (two draws): the thread's state jumps 2 * grid * block steps per iteration
            lcg_jump_host(2ull * grid * (unsigned long long)t.block, &rng_jm, &rng_jp);
            args.push_back(&rng_jm);
            args.push_back(&rng_jp);
            for (size_t k = 0; k < n_in; ++k) c->lazy_randn_fused += (rng_mask >> k) & 1u;
        }
        e = hipModuleLaunchKernel(vec_ok ? kern->fn_fast : kern->fn_fast1, grid, 1, 1, t.block, 1, 1, 0, c->stream,
                                  args.data(), nullptr);
    } else {
        std::vector<unsigned long long> p(11 + 8 * n_in, 0);
        const unsigned long long d0 = oshape[0];
        const unsigned long long per_block = (unsigned long long)t.bcast_block * t.bcast_elems;
        const unsigned long long nchunks = (d0 + per_block - 1) / per_block;
        unsigned long long outer = 1;
        p[0] = d0;
        p[1] = nchunks;
        p[2] = crank;
        for (size_t d = 0; d < 8; ++d) p[3 + d] = d < crank ? oshape[d] : 1;
        for (size_t d = 1; d < crank; ++d) outer *= oshape[d];
        for (size_t k = 0; k < n_in; ++k)
            for (size_t d = 0; d < crank; ++d) p[11 + 8 * k + d] = strides[k][d];
        args.push_back(p.data());
        if (d0 < 128 && outer >= 64 && len < 0x80000000ULL) {  // short dim 0, many outer indices: flat threads (32-bit index + stride cannot wrap below 2^31)
            unsigned n32 = (unsigned)len;
            args.push_back(&n32);
            const unsigned long long want = (len + 255) / 256, cap = (unsigned long long)c->num_cus * 16;
            e = hipModuleLaunchKernel(kern->fn_bcast_flat, (unsigned)std::min(want, cap), 1, 1, 256, 1, 1, 0, c->stream, args.data(), nullptr);
        } else {
            const unsigned long long blocks = nchunks * outer;
            const unsigned long long gx = std::min<unsigned long long>(blocks, 1048576ULL);
            const unsigned long long gy = (blocks + gx - 1) / gx;
            if (gy > 65535ULL) {
                for (size_t k = 0; k < n_out; ++k) rmhip_free(ctx, ids[k]);
                return fail(RMHIP_ERR_UNSUPPORTED, "fused_elementwise: broadcast grid too large");
            }
            e = hipModuleLaunchKernel(kern->fn_bcast, (unsigned)gx, (unsigned)gy, 1, t.bcast_block, 1, 1, 0, c->stream, args.data(),
                                      nullptr);
        }
    }
    if (e != hipSuccess) {
        for (size_t k = 0; k < n_out; ++k) rmhip_free(ctx, ids[k]);
        return fail(RMHIP_ERR_HIP, "fused_elementwise launch: %s", hipGetErrorString(e));
    }
    c->tel.kernel_launches++;
    if (n_out == 1) c->record_launch("fused_elementwise", {{"len", len}, {"inputs", n_in}, {"rank", rank}}, {{"wg", (uint64_t)(fast ? t.block : t.bcast_block)}});
    else c->record_launch("fused_elementwise_multi", {{"len", len}, {"inputs", n_in}, {"rank", rank}, {"num_outputs", n_out}},
                          {{"wg", (uint64_t)(fast ? t.block : t.bcast_block)}});
    for (size_t k = 0; k < n_out; ++k) out_ids[k] = ids[k];
    RMHIP_TRACEF("fused_elementwise: launched");
    return RMHIP_OK;
}

int rmhip_fused_reduction(rmhip_ctx* ctx, const char* shader, const rmhip_buf* inputs, size_t n_in,
                          const size_t* out_shape, size_t rank, size_t reduce_len, size_t num_slices,
                          uint32_t workgroup_size, int flavor, double custom_scale, rmhip_buf* out) {
    (void)workgroup_size;
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.fused_reduction_count, &c->tel.fused_reduction_ns);
    if (!shader || !inputs || !out) return fail(RMHIP_ERR_INVALID, "fused_reduction: null argument");
    if (n_in == 0 || n_in > 24) return fail(RMHIP_ERR_UNSUPPORTED, "fused_reduction: unsupported input count %zu", n_in);
    if (reduce_len * num_slices == 0) return fail(RMHIP_ERR_UNSUPPORTED, "fusion: zero-length execution not supported");  // fusion_exec.rs:489
    if (shape_numel(out_shape, rank) != num_slices)
        return fail(RMHIP_ERR_SHAPE, "fused_reduction: prod(output_shape) != num_slices %zu", num_slices);
    if (flavor < RMHIP_FLAVOR_SUM || flavor > RMHIP_FLAVOR_CUSTOM_SCALE) return fail(RMHIP_ERR_INVALID, "fused_reduction: bad flavor %d", flavor);
    std::string err;
    const std::shared_ptr<const ReductionProgram> prog_ptr = g_red_parse_cache.get(shader, parse_reduction_wgsl, &err);
    if (!prog_ptr) return fail(RMHIP_ERR_COMPILE, "WGSL front-end: %s", err.c_str());
    const ReductionProgram& prog = *prog_ptr;
    if ((size_t)prog.n_inputs != n_in) return fail(RMHIP_ERR_INVALID, "fused_reduction: shader binds %d inputs, got %zu", prog.n_inputs, n_in);

    if (prog.f32 != (c->precision == 32))
        return fail(RMHIP_ERR_COMPILE, "%s shader handed to an %s provider (precision() is %s)", prog.f32 ? "f32" : "f64",
                    c->precision == 32 ? "F32" : "F64", c->precision == 32 ? "F32" : "F64");
    const size_t total = reduce_len * num_slices;
    std::vector<Buffer> in(n_in);
    std::vector<unsigned long long> mult(n_in);
    bool f32 = c->precision == 32;  // f32 operands are read in place, partials and the result are f64 (narrowed on return)
    size_t tried = 0;
    for (; tried < n_in && f32; ++tried) RMHIP_TRY(get_operand(c, inputs[tried], &in[tried], &f32));
    for (size_t k = 0; k < n_in; ++k) {
        if (!f32 && k + 1 != tried) RMHIP_TRY(c->get(inputs[k], &in[k]));
        if (in[k].numel == total) mult[k] = 1;
        else if (in[k].numel == 1) mult[k] = 0;  // scalar operand uploaded as a 1-element tensor (fusion_exec.rs:522-543)
        else return fail(RMHIP_ERR_SHAPE, "fused_reduction: input %zu has %zu elements, expected %zu", k, in[k].numel, total);
    }
    std::shared_ptr<FusedKernel> kern;
    RMHIP_TRY(get_reduction_kernel(c, prog, f32, &kern));

    // axis 0: slice s is contiguous (pre=1, red, post=slices); axis 1: element (s, r) at s + r*slices.
    const size_t pre = prog.axis == 0 ? 1 : num_slices;
    const size_t post = prog.axis == 0 ? num_slices : 1;
    const ReducePlan plan = plan_reduction(pre, reduce_len, post, c->num_cus, f32 ? 4u : 8u);
    if (!plan.valid) return fail(RMHIP_ERR_UNSUPPORTED, "fused_reduction: geometry exceeds launch limits");
    std::vector<const double*> in_ptr(n_in);
    std::vector<void*> args;
    for (size_t k = 0; k < n_in; ++k) {
        in_ptr[k] = in[k].data();
        args.push_back(&in_ptr[k]);
        args.push_back(&mult[k]);
    }
    // the 16-byte forms need every full-size input aligned to its pair
    bool pairs_ok = true;
    for (size_t k = 0; k < n_in && pairs_ok; ++k)
        if (mult[k] && (((uintptr_t)in_ptr[k]) & (f32 ? 7u : 15u)) != 0) pairs_ok = false;
    // kernel B over 16-byte vectors (two adjacent slices per thread) with the XCD-pinned window geometry of sum(x,2) (reduce_plan.h):
    // even `pre` >= 512
    const bool wide_b = !plan.contiguous && pairs_ok && (pre & 1) == 0 && pre >= 512 && post <= 65535;
    StridedWidePlan wplan{};
    if (wide_b) wplan = plan_strided_wide(pre, reduce_len, post, c->num_cus, c->num_xcc, f32 ? 4u : 8u);
    const unsigned long long nsplit_used = wide_b ? wplan.nsplit : plan.nsplit;
    const size_t nparts = (size_t)(plan.nslices * nsplit_used);
    RMHIP_TRY(c->ensure_scratch(2 * nparts * sizeof(double)));
    double* pv = c->scratch;
    double* pn = c->scratch + nparts;

    Buffer ob;
    rmhip_buf oid = 0;
    RMHIP_TRY(c->new_buffer(out_shape, rank, &oid, &ob));

    unsigned long long u_pre = pre, u_red = reduce_len, u_nsplit = nsplit_used, u_nslices = plan.nslices;
    int tx = plan.tx;
    hipError_t e;
    if (plan.contiguous) {
        args.push_back(&u_red);
        args.push_back(&u_nslices);
        args.push_back(&u_nsplit);
        args.push_back(&pv);
        args.push_back(&pn);
        // 16-byte form (two adjacent elements per call of the value functor): even slices of at least 2048 elements, every full-size
        // input aligned to its pair - as k_reduce_contig_v2 for plain tensors (+12-18 % there; the Monte-Carlo payoff sum 160 -> us)
        const bool wide = pairs_ok && (reduce_len & 1) == 0 && reduce_len >= 2048;
        e = hipModuleLaunchKernel(wide ? kern->fn_contig2 : kern->fn_contig, plan.gx, plan.gy, plan.gz, (unsigned)plan.tx, 1, 1, 0, c->stream,
                                  args.data(), nullptr);
    } else if (wide_b) {
        unsigned win = wplan.win;
        args.push_back(&u_pre);
        args.push_back(&u_red);
        args.push_back(&u_nsplit);
        args.push_back(&win);
        args.push_back(&pv);
        args.push_back(&pn);
        e = hipModuleLaunchKernel(kern->fn_strided2, wplan.bx, (unsigned)wplan.nsplit, (unsigned)post, wplan.threads, 1, 1, 0, c->stream, args.data(),
                                  nullptr);
    } else {
        args.push_back(&u_pre);
        args.push_back(&u_red);
        args.push_back(&u_nsplit);
        args.push_back(&tx);
        args.push_back(&pv);
        args.push_back(&pn);
        e = hipModuleLaunchKernel(kern->fn_strided, plan.gx, plan.gy, plan.gz, 256, 1, 1, 0, c->stream, args.data(), nullptr);
    }
    if (e == hipSuccess) {
        const double* cpv = pv;
        const double* cpn = pn;
        int mean = flavor == RMHIP_FLAVOR_MEAN ? 1 : 0;
        int omit = prog.omitnan ? 1 : 0;
        double scale = flavor == RMHIP_FLAVOR_CUSTOM_SCALE ? custom_scale : 1.0;
        double* optr = ob.data();
        void* fargs[] = {&cpv, &cpn, &u_nslices, &u_nsplit, &u_red, &mean, &omit, &scale, &optr};
        if (nsplit_used <= 8 && plan.nslices >= 1024) {
            e = hipModuleLaunchKernel(kern->fn_final_flat, (unsigned)ceil_div_u64(plan.nslices, 256), 1, 1, 256, 1, 1, 0, c->stream, fargs, nullptr);
        } else {
            const unsigned fb = (unsigned)ceil_div_u64(plan.nslices, 4);
            e = hipModuleLaunchKernel(kern->fn_final, fb, 1, 1, 256, 1, 1, 0, c->stream, fargs, nullptr);
        }
    }
    if (e != hipSuccess) {
        rmhip_free(ctx, oid);
        return fail(RMHIP_ERR_HIP, "fused_reduction launch: %s", hipGetErrorString(e));
    }
    c->tel.kernel_launches += 2;
    c->record_launch("fused_reduction", {{"reduce_len", reduce_len}, {"slices", num_slices}, {"rank", rank}},
                     {{"wg", (uint64_t)(plan.contiguous ? plan.tx : 256)}, {"flavor", (uint64_t)flavor}});
    *out = oid;
    return RMHIP_OK;
}

int rmhip_unary(rmhip_ctx* ctx, int op, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op < 0 || op >= RMHIP_UNARY_OP_COUNT) return fail(RMHIP_ERR_UNSUPPORTED, "unary op %d not supported by provider", op);
    Buffer ab, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    int rc;
    if (f32) {
        RMHIP_TRY(c->new_buffer_f32(ab.shape.data(), ab.shape.size(), out, &ob));
        rc = launch_unary_f32(c, op, ab.data_f32(), ob.data_f32(), ab.numel);
    } else {
        RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), out, &ob));
        rc = launch_unary(c, op, ab.data(), ob.data(), ab.numel);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_scalar(rmhip_ctx* ctx, int op, rmhip_buf a, double s, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op < 0 || op >= RMHIP_SCALAR_OP_COUNT) return fail(RMHIP_ERR_UNSUPPORTED, "scalar op %d not supported by provider", op);
    Buffer ab, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    int rc;
    if (f32) {
        RMHIP_TRY(c->new_buffer_f32(ab.shape.data(), ab.shape.size(), out, &ob));
        rc = launch_scalar_f32(c, op, ab.data_f32(), s, ob.data_f32(), ab.numel);
    } else {
        RMHIP_TRY(c->new_buffer(ab.shape.data(), ab.shape.size(), out, &ob));
        rc = launch_scalar(c, op, ab.data(), s, ob.data(), ab.numel);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_binary(rmhip_ctx* ctx, int op, rmhip_buf a, rmhip_buf b, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op < 0 || op >= RMHIP_BINARY_OP_COUNT) return fail(RMHIP_ERR_UNSUPPORTED, "binary op %d not supported by provider", op);
    Buffer ab, bb, ob;
    bool f32 = c->precision == 32;
    // repmat views stay views: the reference's callers expand with `repmat`, call `elem_*`, free (times.rs:501-543)
    RMHIP_TRY(get_bcast_operand(c, a, &ab, &f32));
    RMHIP_TRY(get_bcast_operand(c, b, &bb, &f32));
    if (!f32 && ab.dtype == DT_F32) RMHIP_TRY(c->get(a, &ab));  // b turned out not to be plain f32 storage
    // broadcast_shapes (broadcast.rs:8-47): front-pad, extents equal or 1
    const size_t rank = std::max(ab.shape.size(), bb.shape.size());
    if (rank > 16) return fail(RMHIP_ERR_UNSUPPORTED, "binary: rank too large");
    std::vector<size_t> oshape(rank);
    for (size_t d = 0; d < rank; ++d) {
        const size_t ea = d < rank - ab.shape.size() ? 1 : ab.shape[d - (rank - ab.shape.size())];
        const size_t eb = d < rank - bb.shape.size() ? 1 : bb.shape[d - (rank - bb.shape.size())];
        if (ea == eb) oshape[d] = ea;
        else if (ea == 1) oshape[d] = eb;
        else if (eb == 1) oshape[d] = ea;
        else
            return fail(RMHIP_ERR_SHAPE, "size mismatch between inputs (dimension %zu has lengths %zu and %zu)", d + 1, ea, eb);
    }
    if (f32) RMHIP_TRY(c->new_buffer_f32(oshape.data(), rank, out, &ob));
    else RMHIP_TRY(c->new_buffer(oshape.data(), rank, out, &ob));
    int rc;
    if (ab.numel == ob.numel && bb.numel == ob.numel && ab.rep_base.empty() && bb.rep_base.empty()) {
        rc = f32 ? launch_binary_same_f32(c, op, ab.data_f32(), bb.data_f32(), ob.data_f32(), ob.numel)
                 : launch_binary_same(c, op, ab.data(), bb.data(), ob.data(), ob.numel);
    } else {
        std::vector<std::vector<uint64_t>> strides;
        std::vector<uint64_t> os;
        const rmhip_buf ids[2] = {a, b};
        std::vector<Buffer> in = {ab, bb};
        rc = bcast_prepare(c, ids, &in, f32, oshape.data(), rank, &os, &strides, "binary");
        if (rc) {
            rmhip_free(ctx, *out);
            return rc;
        }
        ab = in[0];
        bb = in[1];
        collapse(&os, &strides);
        if (os.size() > 8) {
            rmhip_free(ctx, *out);
            return fail(RMHIP_ERR_UNSUPPORTED, "binary: broadcast rank %zu > 8 after collapsing", os.size());
        }
        BroadcastDesc d{};
        d.rank = (int)os.size();
        for (size_t i = 0; i < os.size(); ++i) {
            d.out_shape[i] = os[i];
            d.stride_a[i] = strides[0][i];
            d.stride_b[i] = strides[1][i];
        }
        rc = f32 ? launch_binary_bcast_f32(c, op, ab.data_f32(), bb.data_f32(), ob.data_f32(), ob.numel, d)
                 : launch_binary_bcast(c, op, ab.data(), bb.data(), ob.data(), ob.numel, d);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_reduce(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int nan_mode, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op < 0 || op >= RMHIP_REDUCE_OP_COUNT) return fail(RMHIP_ERR_UNSUPPORTED, "reduce op %d not supported by provider", op);
    Buffer ab, ob;
    bool f32 = c->precision == 32;  // f32 storage is read in place; the (small) result is f64 and narrowed on return
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    if (dim < 0) {
        const size_t oshape[2] = {1, 1};  // simple_provider.rs:6743
        RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
        int rc = f32 ? launch_reduce_mid_f32(c, op, nan_mode, ab.data_f32(), 1, ab.numel, 1, ob.data())
                     : launch_reduce_all(c, op, nan_mode, ab.data(), ab.numel, ob.data());
        if (rc) rmhip_free(ctx, *out);
        return rc;
    }
    std::vector<size_t> shape = normalize_matrix_shape(ab.shape);
    if ((size_t)dim >= shape.size()) return fail(RMHIP_ERR_UNSUPPORTED, "reduce: dim %d out of range for rank %zu", dim, shape.size());
    size_t pre = 1, post = 1;
    for (int d = 0; d < dim; ++d) pre *= shape[d];
    for (size_t d = dim + 1; d < shape.size(); ++d) post *= shape[d];
    const size_t red = shape[dim];
    std::vector<size_t> oshape = shape;
    oshape[dim] = 1;
    RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), out, &ob));
    int rc = f32 ? launch_reduce_mid_f32(c, op, nan_mode, ab.data_f32(), pre, red, post, ob.data())
                 : launch_reduce_mid(c, op, nan_mode, ab.data(), pre, red, post, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

namespace {
// [pre, red, post] view of `shape` around `dim` (dim < 0: everything is reduced) and the output shape (extent 1 at `dim`; [1,1] for all)
struct DimView {
    size_t pre = 1, red = 1, post = 1;
    std::vector<size_t> oshape;
};
int dim_view(const Buffer& b, int dim, const char* what, DimView* v) {
    const std::vector<size_t> shape = normalize_matrix_shape(b.shape);
    if (dim < 0) {
        v->red = b.numel;
        v->oshape = {1, 1};
        return RMHIP_OK;
    }
    if ((size_t)dim >= shape.size()) return fail(RMHIP_ERR_UNSUPPORTED, "%s: dim %d out of range for rank %zu", what, dim, shape.size());
    for (int d = 0; d < dim; ++d) v->pre *= shape[d];
    for (size_t d = dim + 1; d < shape.size(); ++d) v->post *= shape[d];
    v->red = shape[dim];
    v->oshape = shape;
    v->oshape[dim] = 1;
    return RMHIP_OK;
}
}  // namespace

int rmhip_reduce_minmax_dim(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int nan_mode, rmhip_buf* values, rmhip_buf* indices) {
    CTX_OR_FAIL(ctx);
    if (!values || !indices) return fail(RMHIP_ERR_INVALID, "reduce_minmax_dim: null output");
    if (op != RMHIP_RMIN && op != RMHIP_RMAX) return fail(RMHIP_ERR_INVALID, "reduce_minmax_dim: op must be RMHIP_RMIN or RMHIP_RMAX");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "reduce_minmax_dim: dim must be >= 0");
    Buffer ab, vb, ib;
    bool f32 = c->precision == 32;  // f32 storage is read in place and widened in registers (exact, order preserving); outputs are narrowed on return
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    DimView v;
    RMHIP_TRY(dim_view(ab, dim, "reduce_minmax_dim", &v));
    if (ab.numel == 0) return fail(RMHIP_ERR_UNSUPPORTED, "reduce_minmax_dim: empty tensor");
    RMHIP_TRY(c->new_buffer(v.oshape.data(), v.oshape.size(), values, &vb));
    int rc = c->new_buffer(v.oshape.data(), v.oshape.size(), indices, &ib);
    if (!rc)
        rc = f32 ? launch_argreduce_f32(c, op, nan_mode, ab.data_f32(), v.pre, v.red, v.post, vb.data(), ib.data())
                 : launch_argreduce(c, op, nan_mode, ab.data(), v.pre, v.red, v.post, vb.data(), ib.data());
    if (rc) {
        rmhip_free(ctx, *values);
        if (*indices) rmhip_free(ctx, *indices);
        return rc;
    }
    c->record_launch(op == RMHIP_RMIN ? "reduce_min_dim" : "reduce_max_dim", {{"reduce_len", v.red}, {"slices", v.pre * v.post}}, {{"wg", 256}});
    return RMHIP_OK;
}

int rmhip_reduce_std(rmhip_ctx* ctx, rmhip_buf a, int dim, int normalization, int nan_mode, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (normalization != 0 && normalization != 1) return fail(RMHIP_ERR_INVALID, "reduce_std: normalization must be 0 (sample) or 1 (population)");
    Buffer ab, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    DimView v;
    RMHIP_TRY(dim_view(ab, dim, "reduce_std", &v));
    if (ab.numel == 0) return fail(RMHIP_ERR_UNSUPPORTED, "reduce_std: empty tensor");
    RMHIP_TRY(c->new_buffer(v.oshape.data(), v.oshape.size(), out, &ob));
    const int rc = f32 ? launch_reduce_std_f32(c, normalization, nan_mode, ab.data_f32(), v.pre, v.red, v.post, ob.data())
                       : launch_reduce_std(c, normalization, nan_mode, ab.data(), v.pre, v.red, v.post, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_reduce_truth(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int omit_nan, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op < 0 || op >= RMHIP_TRUTH_OP_COUNT) return fail(RMHIP_ERR_INVALID, "reduce_truth: bad op %d", op);
    Buffer ab, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    DimView v;
    RMHIP_TRY(dim_view(ab, dim, "reduce_truth", &v));
    if (ab.numel == 0) return fail(RMHIP_ERR_UNSUPPORTED, "reduce_truth: empty tensor");
    RMHIP_TRY(c->new_buffer(v.oshape.data(), v.oshape.size(), out, &ob));
    const int rc = f32 ? launch_reduce_truth_f32(c, op, omit_nan, ab.data_f32(), v.pre, v.red, v.post, ob.data())
                       : launch_reduce_truth(c, op, omit_nan, ab.data(), v.pre, v.red, v.post, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_cumulative(rmhip_ctx* ctx, int op, rmhip_buf a, int dim, int reverse, int nan_mode, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (op != 0 && op != 1) return fail(RMHIP_ERR_INVALID, "cumulative: op must be 0 (sum) or 1 (prod)");
    if (dim < 0) return fail(RMHIP_ERR_INVALID, "cumulative: dim must be >= 0");
    Buffer ab, ob;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = normalize_matrix_shape(ab.shape);
    DimView v;
    RMHIP_TRY(dim_view(ab, dim, "cumulative", &v));
    RMHIP_TRY(c->new_buffer(shape.data(), shape.size(), out, &ob));
    const int rc = launch_cumulative(c, op, reverse, nan_mode, ab.data(), v.pre, v.red, v.post, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_reduce_nd(rmhip_ctx* ctx, int op, rmhip_buf a, const size_t* dims_zero_based, size_t ndims, int nan_mode,
                    rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (!dims_zero_based && ndims)) return fail(RMHIP_ERR_INVALID, "reduce_nd: null argument");
    Buffer ab;
    RMHIP_TRY(c->get_raw(a, &ab));  // shape only
    const size_t rank = normalize_matrix_shape(ab.shape).size();
    // nd.rs:62-72: dims beyond the rank are ignored, duplicates dropped, ascending order
    std::vector<size_t> dims;
    for (size_t i = 0; i < ndims; ++i)
        if (dims_zero_based[i] < rank) dims.push_back(dims_zero_based[i]);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    if (dims.empty()) return fail(RMHIP_ERR_INVALID, "reduce_nd: no valid dims to reduce");
    // the CPU reduces one dimension after the other in ascending order (mean of means, mean.rs:1107-1116)
    rmhip_buf cur = a;
    for (size_t i = 0; i < dims.size(); ++i) {
        rmhip_buf next = 0;
        const int rc = rmhip_reduce(ctx, op, cur, (int)dims[i], nan_mode, &next);
        if (cur != a) rmhip_free(ctx, cur);
        if (rc) return rc;
        cur = next;
    }
    *out = cur;
    return RMHIP_OK;
}

int rmhip_reduce_moments_nd(rmhip_ctx* ctx, rmhip_buf a, const size_t* dims_zero_based, size_t ndims, rmhip_buf* mean_out,
                            rmhip_buf* ex2_out) {
    CTX_OR_FAIL(ctx);
    if (!mean_out || !ex2_out) return fail(RMHIP_ERR_INVALID, "reduce_moments_nd: null output");
    size_t numel = 0;
    RMHIP_TRY(rmhip_numel(ctx, a, &numel));
    if (numel == 0) return fail(RMHIP_ERR_UNSUPPORTED, "reduce_moments_nd: empty tensor");  // nd.rs:318
    // The CPU's mean(x, dims) and mean(x .^ 2, dims) are means of means, one dimension after the other in ascending order
    // (mean.rs:1107-1116).  The first step is the only one that reads the whole tensor: ONE pass gives both of its results (sum and
    // sum of squares side by side, reduce2.hip SqAcc - x .* x is never materialised); the later steps run on the small intermediates.
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> shape = normalize_matrix_shape(ab.shape);
    std::vector<size_t> dims;  // nd.rs:62-72: dims beyond the rank are ignored, duplicates dropped, ascending order
    for (size_t i = 0; i < ndims; ++i)
        if (dims_zero_based && dims_zero_based[i] < shape.size()) dims.push_back(dims_zero_based[i]);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    if (dims.empty()) return fail(RMHIP_ERR_INVALID, "reduce_nd: no valid dims to reduce");
    const size_t d0 = dims[0];
    size_t pre = 1, post = 1;
    for (size_t i = 0; i < d0; ++i) pre *= shape[i];
    for (size_t i = d0 + 1; i < shape.size(); ++i) post *= shape[i];
    std::vector<size_t> oshape = shape;
    oshape[d0] = 1;
    rmhip_buf mean = 0, ex2 = 0;
    Buffer mb, eb;
    RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), &mean, &mb));
    int rc = c->new_buffer(oshape.data(), oshape.size(), &ex2, &eb);
    if (!rc) rc = launch_reduce_moments(c, ab.data(), pre, shape[d0], post, mb.data(), eb.data());
    if (!rc && dims.size() > 1) {
        rmhip_buf m2 = 0, e2 = 0;
        rc = rmhip_reduce_nd(ctx, RMHIP_RMEAN, mean, dims.data() + 1, dims.size() - 1, 0, &m2);
        if (!rc) rc = rmhip_reduce_nd(ctx, RMHIP_RMEAN, ex2, dims.data() + 1, dims.size() - 1, 0, &e2);
        if (rc && m2) rmhip_free(ctx, m2);
        if (!rc) {
            rmhip_free(ctx, mean);
            rmhip_free(ctx, ex2);
            mean = m2;
            ex2 = e2;
        }
    }
    if (rc) {
        if (mean) rmhip_free(ctx, mean);
        if (ex2) rmhip_free(ctx, ex2);
        return rc;
    }
    *mean_out = mean;
    *ex2_out = ex2;
    return RMHIP_OK;
}

int rmhip_dot(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, int dim, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    RMHIP_TRY(get_operand(c, b, &bb, &f32));
    if (!f32 && ab.dtype == DT_F32) RMHIP_TRY(c->get(a, &ab));
    const std::vector<size_t> sa = normalize_matrix_shape(ab.shape), sb = normalize_matrix_shape(bb.shape);
    if (sa != sb) return fail(RMHIP_ERR_SHAPE, "dot: A and B must be the same size");
    int d = dim;
    if (d < 0) {  // first non-singleton dimension
        d = 0;
        for (size_t i = 0; i < sa.size(); ++i)
            if (sa[i] != 1) {
                d = (int)i;
                break;
            }
    }
    if ((size_t)d >= sa.size()) return fail(RMHIP_ERR_UNSUPPORTED, "dot: dim %d out of range for rank %zu", d, sa.size());
    size_t pre = 1, post = 1;
    for (int i = 0; i < d; ++i) pre *= sa[i];
    for (size_t i = d + 1; i < sa.size(); ++i) post *= sa[i];
    std::vector<size_t> oshape = sa;
    oshape[d] = 1;
    RMHIP_TRY(c->new_buffer(oshape.data(), oshape.size(), out, &ob));
    int rc = f32 ? launch_reduce_dot_f32(c, ab.data_f32(), bb.data_f32(), pre, sa[d], post, ob.data())
                 : launch_reduce_dot(c, ab.data(), bb.data(), pre, sa[d], post, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_matmul(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.matmul_count, &c->tel.matmul_ns);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb, ob;
    // Precision 32: both operands in f32 storage run on the f32 matrix cores (sgemm.hip; f32 accumulation like the
    // reference's F32 backend).  RMHIP_F32_MATMUL=f64 keeps the widen -> dgemm -> round-once path (the CPU's `single`
    // result exactly); it is also what mixed operands, k == 0 and few-tile / long-k shapes (split-K) use.
    if (c->precision == 32) {
        Buffer ra, rb;
        RMHIP_TRY(c->get_raw(a, &ra));
        RMHIP_TRY(c->get_raw(b, &rb));
        if (!ra.rep_base.empty()) {
            RMHIP_TRY(c->settle_view(a));
            RMHIP_TRY(c->get_raw(a, &ra));
        }
        if (!rb.rep_base.empty()) {
            RMHIP_TRY(c->settle_view(b));
            RMHIP_TRY(c->get_raw(b, &rb));
        }
        if (ra.dtype == DT_F32 && rb.dtype == DT_F32 && ra.shape.size() == 2 && rb.shape.size() == 2) {
            if (ra.tview && rb.tview) {
                RMHIP_TRY(c->settle_view(b));
                RMHIP_TRY(c->get_raw(b, &rb));
            }
            const size_t m = ra.shape[0], k = ra.shape[1], kb = rb.shape[0], n = rb.shape[1];
            if (k != kb) return fail(RMHIP_ERR_SHAPE, "matmul: inner dims must agree (%zux%zu * %zux%zu)", m, k, kb, n);
            if (f32_gemm_eligible(c, m, n, k)) {
                const size_t oshape[2] = {m, n};
                RMHIP_TRY(c->new_buffer_f32(oshape, 2, out, &ob));
                int rc = launch_sgemm_trans(c, ra.tview, rb.tview, m, n, k, ra.data_f32(), ra.tview ? k : m, rb.data_f32(),
                                            rb.tview ? n : k, ob.data_f32(), m);
                if (rc) rmhip_free(ctx, *out);
                else c->record_launch("matmul", {{"m", m}, {"n", n}, {"k", k}}, {{"mfma_f32", 1}, {"ta", (uint64_t)ra.tview}, {"tb", (uint64_t)rb.tview}});
                return rc;
            }
        }
    }
    // transpose views are consumed in place (A'*B, A*B'); with both operands transposed B is materialised
    RMHIP_TRY(c->get_view(a, &ab));
    RMHIP_TRY(c->get_view(b, &bb));
    if (ab.tview && bb.tview) RMHIP_TRY(c->get(b, &bb));
    if (ab.shape.size() != 2 || bb.shape.size() != 2) return fail(RMHIP_ERR_UNSUPPORTED, "matmul: only 2D supported");  // simple_provider.rs:7705
    const size_t m = ab.shape[0], k = ab.shape[1], kb = bb.shape[0], n = bb.shape[1];
    if (k != kb) return fail(RMHIP_ERR_SHAPE, "matmul: inner dims must agree (%zux%zu * %zux%zu)", m, k, kb, n);
    const size_t oshape[2] = {m, n};
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    int rc = RMHIP_OK;
    if (k == 0) rc = launch_fill(c, ob.data(), ob.numel, 0.0);
    else if (ab.tview || bb.tview)  // storage: A' is k x m (ld k), B' is n x k (ld n)
        rc = launch_dgemm_trans(c, ab.tview, bb.tview, m, n, k, 1.0, ab.data(), ab.tview ? k : m, bb.data(), bb.tview ? n : k,
                                0.0, ob.data(), m);
    else rc = launch_dgemm(c, m, n, k, 1.0, ab.data(), m, bb.data(), k, 0.0, ob.data(), m);
    if (rc) rmhip_free(ctx, *out);
    else c->record_launch("matmul", {{"m", m}, {"n", n}, {"k", k}}, {{"mfma_f64", 1}, {"ta", (uint64_t)ab.tview}, {"tb", (uint64_t)bb.tview}});
    return rc;
}

int rmhip_matmul_power_step(rmhip_ctx* ctx, rmhip_buf lhs, rmhip_buf rhs, double epsilon, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    // simple_provider.rs:7859-7884 composed from the provider's own ops: P = lhs*rhs; acc_c = sum_r P(r,c)^2 (+ eps);
    // P(:,c) /= sqrt(acc_c).  The column sums run in the dot kernels (k_dot_*), the division in the broadcast kernel.
    rmhip_buf p = 0, sq = 0, sq_eps = 0, norms = 0;
    int rc = rmhip_matmul(ctx, lhs, rhs, &p);
    if (!rc) rc = rmhip_dot(ctx, p, p, 0, &sq);
    if (!rc) rc = rmhip_scalar(ctx, RMHIP_SADD, sq, epsilon, &sq_eps);
    if (!rc) rc = rmhip_unary(ctx, RMHIP_SQRT, sq_eps, &norms);
    if (!rc) rc = rmhip_binary(ctx, RMHIP_DIV, p, norms, out);
    for (rmhip_buf t : {p, sq, sq_eps, norms})
        if (t) rmhip_free(ctx, t);
    return rc;
}

int rmhip_image_normalize(rmhip_ctx* ctx, rmhip_buf input, const rmhip_image_normalize_t* d, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || !d) return fail(RMHIP_ERR_INVALID, "image_normalize: null argument");
    if (!std::isfinite(d->epsilon)) return fail(RMHIP_ERR_INVALID, "image_normalize: epsilon must be finite");
    if (d->epsilon < 0.0) return fail(RMHIP_ERR_INVALID, "image_normalize: epsilon must be non-negative");
    Buffer ib, ob;
    bool f32 = c->precision == 32 && d->batch <= 256;  // more planes than that: the f64 kernels on a widened copy (special.hip IN_MAX_BATCH)
    RMHIP_TRY(get_operand(c, input, &ib, &f32));
    if (ib.shape.size() != 3) return fail(RMHIP_ERR_SHAPE, "image_normalize: expected 3-D tensor, got rank %zu", ib.shape.size());
    if (ib.shape[0] != d->batch || ib.shape[1] != d->height || ib.shape[2] != d->width)
        return fail(RMHIP_ERR_SHAPE, "image_normalize: descriptor dims (%zu, %zu, %zu) do not match tensor shape (%zu, %zu, %zu)",
                    d->batch, d->height, d->width, ib.shape[0], ib.shape[1], ib.shape[2]);
    int rc;
    if (f32) {
        RMHIP_TRY(c->new_buffer_f32(ib.shape.data(), 3, out, &ob));
        rc = image_normalize_device_f32(c, ib.data_f32(), ob.data_f32(), d->batch, d->height, d->width, d->epsilon, d->has_gain,
                                        d->gain, d->has_bias, d->bias, d->clamp_zero, d->has_gamma, d->gamma);
    } else {
        RMHIP_TRY(c->new_buffer(ib.shape.data(), 3, out, &ob));
        rc = image_normalize_device(c, ib.data(), ob.data(), d->batch, d->height, d->width, d->epsilon, d->has_gain, d->gain,
                                    d->has_bias, d->bias, d->clamp_zero, d->has_gamma, d->gamma);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

// rank / cond / pinv: the CPU decomposes with nalgebra's SVD (rank.rs:280-295, cond.rs:326-330, 448-467, pinv.rs:276-285); here the one-sided
// Jacobi decomposition of svdsolve.hip supplies the singular values (relative accuracy) and the pseudo-inverse.
static int matrix_dims_2d(const char* who, const Buffer& b, size_t* rows, size_t* cols) {
    for (size_t d = 2; d < b.shape.size(); ++d)
        if (b.shape[d] != 1) return fail(RMHIP_ERR_INVALID, "%s: inputs must be 2-D matrices or vectors", who);
    *rows = b.shape.empty() ? 1 : b.shape[0];
    *cols = b.shape.size() < 2 ? 1 : b.shape[1];
    return RMHIP_OK;
}

static int scalar_result(Context* c, double v, rmhip_buf* out) {
    const size_t one[2] = {1, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(one, 2, out, &ob));
    return launch_fill(c, ob.data(), 1, v);
}

int rmhip_rank(rmhip_ctx* ctx, rmhip_buf matrix, int has_tolerance, double tolerance, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb;
    RMHIP_TRY(c->get(matrix, &mb));
    size_t rows, cols;
    RMHIP_TRY(matrix_dims_2d("rank", mb, &rows, &cols));
    if (rows == 0 || cols == 0) return scalar_result(c, 0.0, out);  // rank.rs:283-285
    std::vector<double> sv;
    RMHIP_TRY(svd_values_host(c, "rank", mb.data(), rows, cols, &sv));
    const double cutoff = has_tolerance ? tolerance : svd_default_tolerance(sv, rows, cols);
    size_t r = 0;
    for (double v : sv) r += (std::isinf(v) || v > cutoff) ? 1 : 0;
    return scalar_result(c, (double)r, out);
}

int rmhip_cond(rmhip_ctx* ctx, rmhip_buf matrix, int norm, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (norm != 0) return fail(RMHIP_ERR_UNSUPPORTED, "cond: only the 2-norm is served (the 1 / inf / fro forms invert on the CPU path)");
    Buffer mb;
    RMHIP_TRY(c->get(matrix, &mb));
    size_t rows, cols;
    RMHIP_TRY(matrix_dims_2d("cond", mb, &rows, &cols));
    if (rows == 0 || cols == 0) return scalar_result(c, 0.0, out);  // cond.rs:278-280
    std::vector<double> sv;
    RMHIP_TRY(svd_values_host(c, "cond", mb.data(), rows, cols, &sv));
    double mn = INFINITY, mx = 0.0;  // singular_value_cond, cond.rs:448-467
    for (double v : sv) {
        const double a = std::fabs(v);
        if (!std::isfinite(a)) return scalar_result(c, INFINITY, out);
        mn = a < mn ? a : mn, mx = a > mx ? a : mx;
    }
    return scalar_result(c, mn == 0.0 ? INFINITY : mx / mn, out);
}

int rmhip_rcond(rmhip_ctx* ctx, rmhip_buf matrix, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb;
    RMHIP_TRY(c->get(matrix, &mb));
    size_t rows, cols;
    RMHIP_TRY(matrix_dims_2d("rcond", mb, &rows, &cols));
    if (rows != cols) return fail(RMHIP_ERR_INVALID, "rcond: input must be a square matrix.");
    if (rows == 0) return scalar_result(c, INFINITY, out);  // rcond.rs:311-313
    std::vector<double> sv;
    RMHIP_TRY(svd_values_host(c, "rcond", mb.data(), rows, cols, &sv));
    double mn = INFINITY, mx = 0.0;  // singular_value_rcond, common/linalg.rs:241-259
    for (double v : sv) {
        const double a = std::fabs(v);
        if (!std::isfinite(a)) return scalar_result(c, 0.0, out);
        mn = a < mn ? a : mn, mx = a > mx ? a : mx;
    }
    return scalar_result(c, mx == 0.0 ? 0.0 : mn / mx, out);
}

int rmhip_pinv(rmhip_ctx* ctx, rmhip_buf matrix, int has_tolerance, double tolerance, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (has_tolerance && !(tolerance >= 0.0)) return fail(RMHIP_ERR_INVALID, "pinv: tolerance must be >= 0");
    Buffer mb, ob;
    RMHIP_TRY(c->get(matrix, &mb));
    size_t rows, cols;
    RMHIP_TRY(matrix_dims_2d("pinv", mb, &rows, &cols));
    const size_t oshape[2] = {cols, rows};
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    if (ob.numel == 0) return RMHIP_OK;  // pinv.rs:245-248
    const int rc = svd_pinv_device(c, mb.data(), rows, cols, has_tolerance ? tolerance : -1.0, ob.data());
    if (rc != RMHIP_OK) {
        rmhip_free(ctx, *out);
        *out = 0;
    }
    return rc;
}

int rmhip_covariance(rmhip_ctx* ctx, rmhip_buf matrix, int biased, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb;
    RMHIP_TRY(c->get_raw(matrix, &mb));  // shape only: the steps below fetch the data themselves
    if (mb.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "covariance: only 2D supported");
    const std::vector<size_t> ms = normalize_matrix_shape(mb.shape);
    const size_t rows = ms[0], cols = ms[1];
    const size_t oshape[2] = {cols, cols};
    const double denom = biased ? (double)rows : (double)rows - 1.0;
    if (cols == 0 || denom <= 0.0) {  // cov.rs:920-934: empty, or the all-NaN matrix
        Buffer ob;
        RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
        int rc0 = launch_fill(c, ob.data(), ob.numel, std::numeric_limits<double>::quiet_NaN());
        if (rc0) rmhip_free(ctx, *out);
        return rc0;
    }
    rmhip_buf means = 0, centred = 0, gram = 0;
    int rc = rmhip_reduce(ctx, RMHIP_RMEAN, matrix, 0, 0, &means);       // [1, cols]
    if (!rc && gram_skinny_applies(rows, cols) && !std::getenv("RMHIP_NO_GRAM_SKINNY")) {
        // many samples of a few variables: centred on the way, no centred copy, no 256-wide MFMA tiles of 8-32 columns (special.hip)
        // (the division by the denominator and the diagonal rule ride on its second kernel).  f32 storage is read in place; the
        // products and sums are f64 either way and the result rounds once on the way out.
        Buffer xb, mub, ob;
        bool f32 = c->precision == 32;
        rc = get_operand(c, matrix, &xb, &f32);
        if (!rc) rc = c->get(means, &mub);
        if (!rc) rc = c->new_buffer(oshape, 2, out, &ob);
        if (!rc) {
            rc = f32 ? gram_skinny_device_f32(c, xb.data_f32(), rows, cols, mub.data(), denom, true, ob.data())
                     : gram_skinny_device(c, xb.data(), rows, cols, mub.data(), denom, true, ob.data());
            if (rc) rmhip_free(ctx, *out);
        }
        rmhip_free(ctx, means);
        return rc;
    } else {
        if (!rc) rc = rmhip_binary(ctx, RMHIP_SUB, matrix, means, &centred);  // broadcast over rows
        if (!rc) rc = rmhip_syrk(ctx, centred, &gram);                        // Xc' * Xc
    }
    if (!rc) {
        // gram / denom and the CPU's diagonal rules on an f64 result; at precision 32 `gb` is a widened copy and the result
        // is rounded to f32 storage on return (writing through Context::get of an f32 buffer would only touch a temporary)
        Buffer gb, ob;
        rc = c->get(gram, &gb);
        if (!rc) rc = c->new_buffer(oshape, 2, out, &ob);
        if (!rc) {
            rc = launch_scalar(c, RMHIP_SDIV, gb.data(), denom, ob.data(), ob.numel);
            if (!rc) rc = cov_sanitize_diag_device(c, ob.data(), cols);
            if (rc) rmhip_free(ctx, *out);
        }
    }
    for (rmhip_buf t : {means, centred, gram})
        if (t) rmhip_free(ctx, t);
    return rc;
}

int rmhip_diag_extract(rmhip_ctx* ctx, rmhip_buf matrix, long long offset, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer mb;
    RMHIP_TRY(c->get(matrix, &mb));
    for (size_t d = 2; d < mb.shape.size(); ++d)
        if (mb.shape[d] != 1) return fail(RMHIP_ERR_SHAPE, "diag: input must be 2-D");
    const size_t rows = mb.shape.empty() ? 1 : mb.shape[0], cols = mb.shape.size() < 2 ? 1 : mb.shape[1];
    if (rows == 1 || cols == 1 || mb.shape.size() <= 1) return fail(RMHIP_ERR_SHAPE, "diag: matrix input required");
    size_t len = 0;  // simple_provider.rs:2357-2376
    if (offset >= 0) {
        const size_t shift = (size_t)offset;
        len = shift >= cols ? 0 : std::min(rows, cols - shift);
    } else {
        const size_t shift = (size_t)(-offset);
        len = shift >= rows ? 0 : std::min(rows - shift, cols);
    }
    const size_t oshape[2] = {len, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    int rc = diag_extract_device(c, mb.data(), rows, offset, len, ob.data());
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_syrk(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.matmul_count, &c->tel.matmul_ns);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, ob;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, a, &ab, &f32));
    if (ab.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "syrk: only 2D supported");
    const std::vector<size_t> as = normalize_matrix_shape(ab.shape);
    const size_t rows = as[0], cols = as[1];
    const size_t oshape[2] = {cols, cols};
    if (f32 && f32_gemm_eligible(c, cols, cols, rows)) {  // A' * A on the f32 matrix cores
        RMHIP_TRY(c->new_buffer_f32(oshape, 2, out, &ob));
        int rc = launch_sgemm_trans(c, true, false, cols, cols, rows, ab.data_f32(), rows, ab.data_f32(), rows, ob.data_f32(), cols);
        if (rc) rmhip_free(ctx, *out);
        return rc;
    }
    if (f32) RMHIP_TRY(c->get(a, &ab));  // f64 kernel on a widened copy
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    int rc = RMHIP_OK;
    if (rows == 0) rc = launch_fill(c, ob.data(), ob.numel, 0.0);
    else if (!f32 && c->precision == 64 && gram_skinny_applies(rows, cols) && !std::getenv("RMHIP_NO_GRAM_SKINNY"))
        rc = gram_skinny_device(c, ab.data(), rows, cols, nullptr, 1.0, false, ob.data());
    else rc = launch_dgemm_trans(c, true, false, cols, cols, rows, 1.0, ab.data(), rows, ab.data(), rows, 0.0, ob.data(), cols);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_matmul_epilogue(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, const rmhip_matmul_epilogue_t* ep, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.matmul_count, &c->tel.matmul_ns);
    if (!out || !ep) return fail(RMHIP_ERR_INVALID, "matmul_epilogue: null argument");
    Buffer ab, bb, ob, rs, cs, dg;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    if (ab.shape.size() != 2 || bb.shape.size() != 2) return fail(RMHIP_ERR_UNSUPPORTED, "matmul: only 2D supported");
    const size_t m = ab.shape[0], k = ab.shape[1], kb = bb.shape[0], n = bb.shape[1];
    if (k != kb) return fail(RMHIP_ERR_SHAPE, "matmul: inner dims must agree (%zux%zu * %zux%zu)", m, k, kb, n);
    GemmEpilogue e{EP_ACTIVE, ep->alpha, ep->beta, nullptr, nullptr, ep->clamp_min, ep->clamp_max, ep->pow_exponent, nullptr};
    if (ep->row_scale) {
        RMHIP_TRY(c->get(ep->row_scale, &rs));
        if (rs.numel < m) return fail(RMHIP_ERR_SHAPE, "matmul_epilogue: row scale length %zu < %zu rows", rs.numel, m);
        e.row_scale = rs.data();
        e.flags |= EP_ROW | (ep->row_op ? EP_ROW_DIV : 0);
    }
    if (ep->col_scale) {
        RMHIP_TRY(c->get(ep->col_scale, &cs));
        if (cs.numel < n) return fail(RMHIP_ERR_SHAPE, "matmul_epilogue: col scale length %zu < %zu cols", cs.numel, n);
        e.col_scale = cs.data();
        e.flags |= EP_COL | (ep->col_op ? EP_COL_DIV : 0);
    }
    Buffer dg_raw;  // diag_output is written IN PLACE: f32 storage gets the widened copy narrowed back after the launch
    if (ep->diag_output) {
        RMHIP_TRY(c->get_raw(ep->diag_output, &dg_raw));
        if (dg_raw.lazy()) return fail(RMHIP_ERR_UNSUPPORTED, "matmul_epilogue: diag_output must not be a transpose / repmat view");
        RMHIP_TRY(c->detach_views_of(ep->diag_output));  // written in place
        RMHIP_TRY(c->get(ep->diag_output, &dg));
        const size_t expected = m < n ? m : n;
        if (dg.numel < expected)  // simple_provider.rs:7790-7799
            return fail(RMHIP_ERR_SHAPE, "matmul_epilogue: diag_output length %zu insufficient for diag size %zu", dg.numel, expected);
        e.diag = dg.data();
        e.flags |= EP_DIAG;
    }
    if (ep->has_clamp_min) e.flags |= EP_CLAMP_MIN;
    if (ep->has_clamp_max) e.flags |= EP_CLAMP_MAX;
    if (ep->has_pow) e.flags |= EP_POW;
    const size_t oshape[2] = {m, n};
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    int rc = launch_dgemm_epilogue(c, m, n, k, ab.data(), m, bb.data(), k ? k : 1, ob.data(), m, e);
    if (!rc && ep->diag_output && dg_raw.dtype == DT_F32) rc = launch_narrow(c, dg.data(), dg_raw.data_f32(), dg_raw.numel);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_lu(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf out5[5]) {
    CTX_OR_FAIL(ctx);
    if (!out5) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab;
    RMHIP_TRY(c->get(a, &ab));
    if (ab.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "lu: only 2D supported");
    const std::vector<size_t> shape = normalize_matrix_shape(ab.shape);
    const size_t rows = shape[0], cols = shape[1];
    Buffer comb, L, U, P, piv;
    rmhip_buf ids[5] = {0, 0, 0, 0, 0};
    const size_t s_comb[2] = {rows, cols}, s_l[2] = {rows, rows}, s_piv[2] = {rows, 1};
    int rc = c->new_buffer(s_comb, 2, &ids[0], &comb);
    if (!rc) rc = c->new_buffer(s_l, 2, &ids[1], &L);
    if (!rc) rc = c->new_buffer(s_comb, 2, &ids[2], &U);
    if (!rc) rc = c->new_buffer(s_l, 2, &ids[3], &P);
    if (!rc) rc = c->new_buffer(s_piv, 2, &ids[4], &piv);
    int* perm = nullptr;
    std::shared_ptr<Allocation> perm_mem;  // pooled (a hipMalloc / hipFree pair costs two device synchronisations per call)
    if (!rc) rc = c->alloc_device((rows + 2) / 2 + 1, &perm_mem);
    if (!rc) perm = (int*)perm_mem->ptr;
    const size_t ldw = lu_padded_ld(rows);
    std::shared_ptr<Allocation> work;
    if (!rc) rc = c->alloc_device(ldw * (cols ? cols : 1), &work);
    int info = 0;
    if (!rc) rc = lu_copy_and_factor(c, ab.data(), rows, cols, work->ptr, ldw, perm, &info);
    if (!rc && ab.numel) {
        hipError_t e = hipMemcpy2DAsync(comb.data(), rows * sizeof(double), work->ptr, ldw * sizeof(double), rows * sizeof(double),
                                        cols, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "lu copy back: %s", hipGetErrorString(e));
    }
    if (!rc) rc = lu_extract_device(c, comb.data(), rows, cols, perm, L.data(), U.data(), P.data(), piv.data());
    if (rc) {
        for (auto id : ids)
            if (id) rmhip_free(ctx, id);
        return rc;
    }
    for (int i = 0; i < 5; ++i) out5[i] = ids[i];
    return RMHIP_OK;
}

static int mldivide_impl(rmhip_ctx* ctx, Context* c, rmhip_buf a, rmhip_buf b, rmhip_buf* out);

// solve_fallbacks (telemetry.rs:95-99): a soft failure of a solve is what sends the caller to its CPU path
static int count_fallback(Context* c, int rc, const char* unsupported, const char* singular) {
    if (rc == RMHIP_ERR_UNSUPPORTED) c->record_solve_fallback(unsupported);
    else if (rc == RMHIP_ERR_SINGULAR) c->record_solve_fallback(singular);
    return rc;
}

int rmhip_mldivide(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.mldivide_count, &c->tel.mldivide_ns);
    return count_fallback(c, mldivide_impl(ctx, c, a, b, out), "mldivide:unsupported", "mldivide:singular");
}

int rmhip_inv(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab;
    RMHIP_TRY(c->get_raw(a, &ab));
    // matrix_dimensions + inv_real_tensor_impl (inv.rs:209-230, 258-280)
    const std::vector<size_t>& s = ab.shape;
    size_t rows = 1, cols = 1;
    if (s.size() == 1) {
        if (s[0] != 1) return fail(RMHIP_ERR_INVALID, "inv: input must be a square matrix.");
    } else if (s.size() >= 2) {
        for (size_t d = 2; d < s.size(); ++d)
            if (s[d] != 1) return fail(RMHIP_ERR_INVALID, "inv: inputs must be 2-D matrices.");
        rows = s[0];
        cols = s[1];
    }
    if (rows != cols) return fail(RMHIP_ERR_INVALID, "inv: input must be a square matrix.");
    if (rows == 0) {
        Buffer ob;
        return c->new_buffer(s.data(), s.size(), out, &ob);
    }
    // X = A \ I on the LU path (the CPU's nalgebra `try_inverse` is the same factorisation followed by the same substitutions); a pivot
    // below the solver's cut-off is RMHIP_ERR_SINGULAR and the caller's CPU path words the "singular to working precision" error
    rmhip_buf eye = 0, x = 0, same = 0;
    const size_t sq[2] = {rows, rows};
    const std::vector<size_t> given = s;  // (rmhip_reshape renames the shape of the SAME buffer: the operand gets its own back below)
    int rc = rmhip_eye(ctx, sq, 2, &eye);
    const bool reshaped = !rc && given.size() != 2;
    if (reshaped) rc = rmhip_reshape(ctx, a, sq, 2, &same);
    if (!rc) {
        c->solve_strict = true;  // mldivide answers a singular square system with the minimum-norm solution; inv must not
        rc = count_fallback(c, mldivide_impl(ctx, c, a, eye, &x), "inv:unsupported", "inv:singular");
        c->solve_strict = false;
    }
    if (reshaped) rmhip_reshape(ctx, a, given.data(), given.size(), &same);
    if (eye) rmhip_free(ctx, eye);
    if (rc) return rc;
    if (given.size() > 2) {  // inv.rs:402-412: a trailing singleton dimension is kept
        rc = rmhip_reshape(ctx, x, given.data(), given.size(), &same);
        if (rc) {
            rmhip_free(ctx, x);
            return rc;
        }
    }
    *out = x;
    return RMHIP_OK;
}

int rmhip_mrdivide(rmhip_ctx* ctx, rmhip_buf b, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.mrdivide_count, &c->tel.mrdivide_ns);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb;
    RMHIP_TRY(c->get_raw(a, &ab));
    RMHIP_TRY(c->get_raw(b, &bb));
    if (ab.shape.size() > 2 || bb.shape.size() > 2)
        return count_fallback(c, fail(RMHIP_ERR_UNSUPPORTED, "mrdivide: only 2D supported"), "mrdivide:unsupported", "mrdivide:singular");
    const std::vector<size_t> as = normalize_matrix_shape(ab.shape), bs = normalize_matrix_shape(bb.shape);
    if (ab.numel == 1) {  // scalar divisor: lhs * (1/rhs), mrdivide.rs:321-325
        RMHIP_TRY(c->get(a, &ab));
        RMHIP_TRY(c->get(b, &bb));
        double rhs = 0.0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&rhs, ab.data(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        Buffer ob;
        RMHIP_TRY(c->new_buffer(bs.data(), 2, out, &ob));
        int rc = launch_scalar(c, RMHIP_SMUL, bb.data(), 1.0 / rhs, ob.data(), bb.numel);
        if (rc) rmhip_free(ctx, *out);
        return rc;
    }
    if (bs[1] != as[1]) return fail(RMHIP_ERR_SHAPE, "mrdivide: column mismatch (%zu vs %zu)", bs[1], as[1]);  // mrdivide.rs:327
    // X = B / A  <=>  A' X' = B'  (mrdivide.rs:379-388): transpose views feed the LU path (a view is materialised on first use)
    rmhip_buf at = 0, bt = 0, xt = 0, x = 0;
    int rc = rmhip_transpose(ctx, a, &at);
    if (!rc) rc = rmhip_transpose(ctx, b, &bt);
    if (!rc) rc = mldivide_impl(ctx, c, at, bt, &xt);
    if (!rc) rc = rmhip_transpose(ctx, xt, &x);
    if (!rc) rc = c->settle_view(x);  // the result is a plain buffer, not a view of the temporary
    if (at) rmhip_free(ctx, at);
    if (bt) rmhip_free(ctx, bt);
    if (xt) rmhip_free(ctx, xt);
    if (rc) {
        if (x) rmhip_free(ctx, x);
        return count_fallback(c, rc, "mrdivide:unsupported", "mrdivide:singular");
    }
    *out = x;
    return RMHIP_OK;
}


// Rectangular A\b for FULL-RANK, reasonably conditioned A.  The reference answers every shape with the SVD's minimum-norm
// least-squares solution (mldivide.rs:380-404); for full column rank (rows > cols) that is the unique least-squares
// solution, for full row rank (rows < cols) the minimum-norm solution A' (A A')^-1 b.  Both come from the kernels already
// here: the Gram matrix on the MFMA path (A'A or AA'), its LU with partial pivoting, and ONE step of refinement on the
// residual (corrected semi-normal equations): x0 = G^-1 A'b, r = b - A x0, x = x0 + G^-1 A'r - error ~ eps * cond(A) once
// cond(A)^2 * eps < 1.  Anything else stays with the caller's CPU SVD path: a pivot of G below the LU's cut-off, or a
// pivot ratio min|u_ii| / max|u_ii| below 1e-11 (cond(A) beyond ~3e5, or rank deficient) -> RMHIP_ERR_UNSUPPORTED.
static int lstsq_full_rank(rmhip_ctx* ctx, Context* c, const double* A, size_t m, size_t n, const double* B, size_t nrhs,
                           rmhip_buf* out) {
    if (m == 0 || n == 0 || nrhs == 0) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: empty system");
    const bool tall = m > n;
    const size_t g = tall ? n : m;  // order of the Gram matrix
    std::shared_ptr<Allocation> gram, work, perm_mem, t1, t2, t3;
    // Regression shapes - many observations of a few variables: the Gram matrix of [A | b] on the VALU kernel (special.hip) holds A'A
    // and A'b from ONE pass over both; the MFMA route ran 256-wide split-k tiles for 8-32 useful columns (2^20 x 8 \ b: 2.0 ms).
    const bool skinny = tall && gram_skinny_applies(m, n + nrhs) && !std::getenv("RMHIP_NO_GRAM_SKINNY");
    const size_t gl = skinny ? n + nrhs : g;  // leading dimension of the Gram buffer
    RMHIP_TRY(c->alloc_device(gl * gl, &gram));
    // G = A'A (tall) or A A' (wide); the transposed operand is read in place
    if (skinny) RMHIP_TRY(gram_skinny_device(c, A, m, n, nullptr, 1.0, false, gram->ptr, B, nrhs));
    else if (tall) RMHIP_TRY(launch_dgemm_trans(c, true, false, n, n, m, 1.0, A, m, A, m, 0.0, gram->ptr, n));
    else RMHIP_TRY(launch_dgemm_trans(c, false, true, m, m, n, 1.0, A, m, A, m, 0.0, gram->ptr, m));
    const size_t ldw = lu_padded_ld(g);
    RMHIP_TRY(c->alloc_device(ldw * g, &work));
    RMHIP_TRY(c->alloc_device((g + 2) / 2 + 1, &perm_mem));
    int* perm = (int*)perm_mem->ptr;
    int info = 0;
    if (skinny) {  // the leading n x n block of the (n + nrhs)^2 Gram matrix
        std::shared_ptr<Allocation> sq;
        RMHIP_TRY(c->alloc_device(g * g, &sq));
        RMHIP_HIP_CHECK(hipMemcpy2DAsync(sq->ptr, g * sizeof(double), gram->ptr, gl * sizeof(double), g * sizeof(double), g, hipMemcpyDeviceToDevice, c->stream));
        RMHIP_TRY(lu_copy_and_factor(c, sq->ptr, g, g, work->ptr, ldw, perm, &info, true));
    } else {
        RMHIP_TRY(lu_copy_and_factor(c, gram->ptr, g, g, work->ptr, ldw, perm, &info, true));
    }
    if (info > 0)
        return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: rank-deficient rectangular system (%d pivot(s) of the Gram matrix <= 1e-12): CPU SVD path", info);
    {
        // (a strided device-to-host copy of the diagonal - one 8-byte row per pivot - took 15 of the 20 ms of a 1000 x 10000 solve)
        double lo = INFINITY, hi = 0.0;
        size_t zeros = 0;
        RMHIP_TRY(diag_stats_device(c, work->ptr, ldw, g, &lo, &hi, &zeros));
        if (!(lo > 1e-11 * hi))
            return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: ill-conditioned rectangular system (Gram pivot ratio %.2e): CPU SVD path", hi > 0 ? lo / hi : 0.0);
    }
    Buffer ob;
    rmhip_buf oid = 0;
    const size_t oshape[2] = {n, nrhs};
    RMHIP_TRY(c->new_buffer(oshape, 2, &oid, &ob));
    double* X = ob.data();
    int rc = RMHIP_OK;
    auto run = [&]() -> int {
        RMHIP_TRY(c->alloc_device(g * nrhs, &t1));  // right-hand side of the Gram system
        RMHIP_TRY(c->alloc_device(g * nrhs, &t2));  // its solution
        RMHIP_TRY(c->alloc_device(m * nrhs, &t3));  // residual b - A x
        const size_t blk = 256;
        if (tall) {
            if (skinny)  // A'b = the last nrhs columns of the Gram matrix of [A | b]
                RMHIP_HIP_CHECK(hipMemcpy2DAsync(t1->ptr, n * sizeof(double), gram->ptr + n * gl, gl * sizeof(double), n * sizeof(double), nrhs,
                                                 hipMemcpyDeviceToDevice, c->stream));
            else RMHIP_TRY(launch_dgemm_trans(c, true, false, n, nrhs, m, 1.0, A, m, B, m, 0.0, t1->ptr, n));         // A'b
            RMHIP_TRY(lu_solve_device(c, work->ptr, n, ldw, perm, t1->ptr, nrhs, n, X, n));                           // x0
            RMHIP_HIP_CHECK(hipMemcpyAsync(t3->ptr, B, m * nrhs * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            RMHIP_TRY(launch_dgemm(c, m, nrhs, n, -1.0, A, m, X, n, 1.0, t3->ptr, m));                                // r = b - A x0
            if (skinny) {  // A'r the same way
                RMHIP_TRY(gram_skinny_device(c, A, m, n, nullptr, 1.0, false, gram->ptr, t3->ptr, nrhs));
                RMHIP_HIP_CHECK(hipMemcpy2DAsync(t1->ptr, n * sizeof(double), gram->ptr + n * gl, gl * sizeof(double), n * sizeof(double), nrhs,
                                                 hipMemcpyDeviceToDevice, c->stream));
            } else RMHIP_TRY(launch_dgemm_trans(c, true, false, n, nrhs, m, 1.0, A, m, t3->ptr, m, 0.0, t1->ptr, n));  // A'r
            RMHIP_TRY(lu_solve_device(c, work->ptr, n, ldw, perm, t1->ptr, nrhs, n, t2->ptr, n));                     // dx
            RMHIP_TRY(launch_binary_same(c, RMHIP_ADD, X, t2->ptr, X, n * nrhs));
        } else {
            RMHIP_TRY(lu_solve_device(c, work->ptr, m, ldw, perm, B, nrhs, m, t2->ptr, m));                           // y0
            RMHIP_TRY(launch_dgemm_trans(c, true, false, n, nrhs, m, 1.0, A, m, t2->ptr, m, 0.0, X, n));              // x0 = A'y0
            RMHIP_HIP_CHECK(hipMemcpyAsync(t3->ptr, B, m * nrhs * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            RMHIP_TRY(launch_dgemm(c, m, nrhs, n, -1.0, A, m, X, n, 1.0, t3->ptr, m));                                // r = b - A x0
            RMHIP_TRY(lu_solve_device(c, work->ptr, m, ldw, perm, t3->ptr, nrhs, m, t2->ptr, m));                     // dy
            RMHIP_TRY(launch_dgemm_trans(c, true, false, n, nrhs, m, 1.0, A, m, t2->ptr, m, 1.0, X, n));              // x += A'dy
        }
        (void)blk;
        return RMHIP_OK;
    };
    rc = run();
    if (rc) {
        rmhip_free(ctx, oid);
        return rc;
    }
    *out = oid;
    return RMHIP_OK;
}

// What the LU / Gram paths refuse - singular, rank-deficient or ill-conditioned systems - answered the way the reference answers every
// system: minimum-norm least squares from an SVD with its tolerance rule (svdsolve.hip), as long as min(rows, cols) <= svd_max_cols().
// `refused` is the status of the path that gave up (returned unchanged when the system is too large for the SVD path).
static int svd_fallback(rmhip_ctx* ctx, Context* c, int refused, const double* A, size_t m, size_t n, const double* B, size_t nrhs, rmhip_buf* out) {
    if (c->solve_strict || (m < n ? m : n) > (size_t)svd_max_cols() || std::getenv("RMHIP_NO_SVD_PATH")) return refused;
    Buffer ob;
    rmhip_buf oid = 0;
    const size_t oshape[2] = {n, nrhs};
    RMHIP_TRY(c->new_buffer(oshape, 2, &oid, &ob));
    int rank = 0;
    const int rc = svd_solve_device(c, A, m, n, B, nrhs, ob.data(), &rank);
    if (rc) {
        rmhip_free(ctx, oid);
        return rc;
    }
    c->svd_solves++;
    *out = oid;
    return RMHIP_OK;
}

static int mldivide_impl(rmhip_ctx* ctx, Context* c, rmhip_buf a, rmhip_buf b, rmhip_buf* out) {
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, bb;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    if (ab.shape.size() > 2 || bb.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: only 2D supported");
    const std::vector<size_t> as = normalize_matrix_shape(ab.shape), bs = normalize_matrix_shape(bb.shape);
    if (ab.numel == 1) {  // scalar lhs: rhs * (1/lhs), mldivide.rs:321-325
        double lhs = 0.0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&lhs, ab.data(), sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        Buffer ob;
        RMHIP_TRY(c->new_buffer(bs.data(), 2, out, &ob));
        int rc = launch_scalar(c, RMHIP_SMUL, bb.data(), 1.0 / lhs, ob.data(), bb.numel);
        if (rc) rmhip_free(ctx, *out);
        return rc;
    }
    if (as[0] != bs[0]) return fail(RMHIP_ERR_SHAPE, "mldivide: row mismatch (%zu vs %zu)", as[0], bs[0]);
    if (as[0] != as[1]) {
        const int lrc = lstsq_full_rank(ctx, c, ab.data(), as[0], as[1], bb.data(), bs[1], out);
        return lrc == RMHIP_ERR_UNSUPPORTED && as[0] && as[1] && bs[1] ? svd_fallback(ctx, c, lrc, ab.data(), as[0], as[1], bb.data(), bs[1], out) : lrc;
    }
    const size_t n = as[0], nrhs = bs[1];
    if (n == 0) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide: empty system");
    {
        // Small systems: elimination, substitution and the pivot statistics in ONE launch of one workgroup (small_solve.hip) - the blocked
        // path below is a dozen launches and two read-backs whatever the order.  Same decisions: a pivot <= 1e-12 is SINGULAR (-> the SVD
        // path), a pivot ratio below 1e3 n eps lets the SVD decide.  RMHIP_LU_FAST=0 (the grid-wide rule everywhere) and
        // RMHIP_NO_SMALL_SOLVE=1 keep the blocked path.
        const char* fe = std::getenv("RMHIP_LU_FAST");
        if (small_solve_applies(n, nrhs) && !(fe && fe[0] == '0') && !std::getenv("RMHIP_NO_SMALL_SOLVE")) {
            Buffer sb;
            rmhip_buf sid = 0;
            const size_t sshape[2] = {n, nrhs};
            RMHIP_TRY(c->new_buffer(sshape, 2, &sid, &sb));
            double mn = 0.0, mx = 0.0;
            size_t bad = 0;
            int src = small_solve_device(c, ab.data(), bb.data(), n, nrhs, sb.data(), &mn, &mx, &bad);
            if (src) {
                rmhip_free(ctx, sid);
                return src;
            }
            if (bad > 0) {
                rmhip_free(ctx, sid);
                src = fail(RMHIP_ERR_SINGULAR, "mldivide: %zu pivot(s) <= 1e-12; matrix is numerically singular, use the CPU SVD path", bad);
                return svd_fallback(ctx, c, src, ab.data(), n, n, bb.data(), nrhs, out);
            }
            if (!std::getenv("RMHIP_NO_SVD_PATH") && !(mn > 1.0e3 * (double)n * 2.220446049250313e-16 * mx)) {
                rmhip_free(ctx, sid);
                return svd_fallback(ctx, c, fail(RMHIP_ERR_SINGULAR, "mldivide: pivot ratio %.2e: numerically singular", mx > 0 ? mn / mx : 0.0), ab.data(), n, n,
                                    bb.data(), nrhs, out);
            }
            c->lu_fast_count++;
            c->lu_last_growth = 0.0;  // partial pivoting over the whole column: every multiplier is <= 1
            *out = sid;
            return RMHIP_OK;
        }
    }
    // Factorisation workspace with a PADDED leading dimension: with lda a large power of two every
    // element of a row maps to the same HBM channel / L2 slice, and the panel kernels (one lane per
    // row, walking across columns) serialise on it; +32 doubles rotates the channel per column.
    const size_t np = lu_pad_rows(n);  // order of the factorisation (the next multiple of 128 for a large ragged n)
    const size_t ldw = lu_padded_ld(np);
    std::shared_ptr<Allocation> work;
    RMHIP_TRY(c->alloc_device(ldw * np, &work));
    std::shared_ptr<Allocation> perm_mem;  // pooled, released in stream order
    RMHIP_TRY(c->alloc_device((np + 2) / 2 + 1, &perm_mem));
    int* perm = (int*)perm_mem->ptr;
    int info = 0;
    int rc = lu_copy_and_factor(c, ab.data(), n, n, work->ptr, ldw, perm, &info, true, np);
    if (!rc && info > 0) {
        rc = fail(RMHIP_ERR_SINGULAR, "mldivide: %d pivot(s) <= 1e-12; matrix is numerically singular, use the CPU SVD path", info);
        if (nrhs) return svd_fallback(ctx, c, rc, ab.data(), n, n, bb.data(), nrhs, out);  // up to svd_max_cols(): the SVD answer on the device
    }
    if (!rc && n <= (size_t)kSvdProxyMaxCols && n <= (size_t)svd_max_cols() && n > 1 && nrhs && !std::getenv("RMHIP_NO_SVD_PATH")) {
        // A nearly singular matrix need not produce a pivot below the cut-off, yet the reference would DROP its small singular values
        // (s_i <= eps * n * max(s_max, 1), mldivide.rs:396-404) where an LU divides by them.  Pivot ratio as the (cheap, rough) proxy of
        // the condition number: below 1e3 * n * eps the SVD decides.  Only where the SVD path is cheap (n <= kSvdProxyMaxCols, < 0.5 s); larger
        // systems keep the LU answer unless a pivot fell below the cut-off.
        double mn = 0.0, mx = 0.0;
        size_t zeros = 0;
        if (diag_stats_device(c, work->ptr, ldw, n, &mn, &mx, &zeros) == RMHIP_OK && !(mn > 1.0e3 * (double)n * 2.220446049250313e-16 * mx))
            return svd_fallback(ctx, c, fail(RMHIP_ERR_SINGULAR, "mldivide: pivot ratio %.2e: numerically singular", mx > 0 ? mn / mx : 0.0), ab.data(), n, n,
                                bb.data(), nrhs, out);
    }
    Buffer ob;
    rmhip_buf oid = 0;
    const size_t oshape[2] = {n, nrhs};
    if (!rc) rc = c->new_buffer(oshape, 2, &oid, &ob);
    if (!rc) rc = lu_solve_padded(c, work->ptr, n, np, ldw, perm, bb.data(), nrhs, ob.data());
    if (rc) {
        if (oid) rmhip_free(ctx, oid);
        return rc;
    }
    *out = oid;
    return RMHIP_OK;
}

int rmhip_transpose(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab;
    RMHIP_TRY(c->get_raw(a, &ab));  // the alias keeps the operand's storage type
    if (ab.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "transpose: only 2D supported");
    if (!ab.rep_base.empty()) {  // a view of a repmat view: tile first
        RMHIP_TRY(c->settle_view(a));
        RMHIP_TRY(c->get_raw(a, &ab));
    }
    const std::vector<size_t> as = normalize_matrix_shape(ab.shape);
    // No data moves: the result aliases the operand's storage as a transpose view (a view of a view is the plain
    // base again; a vector's transpose has the same memory layout).  RMHIP_EAGER_TRANSPOSE=1 materialises at once.
    Buffer r;
    r.alloc = ab.alloc;
    r.shape = {as[1], as[0]};
    r.numel = ab.numel;
    r.tview = (as[0] == 1 || as[1] == 1) ? false : !ab.tview;
    r.dtype = ab.dtype;
    RMHIP_TRY(c->register_buffer(std::move(r), out));
    if (const char* e = std::getenv("RMHIP_EAGER_TRANSPOSE"))
        if (e[0] == '1') RMHIP_TRY(c->settle_view(*out));
    return RMHIP_OK;
}

static int linsolve_impl(rmhip_ctx* ctx, Context* c, rmhip_buf a, rmhip_buf b, const rmhip_linsolve_options_t* opts, rmhip_buf* out,
                         double* reciprocal_condition);

int rmhip_linsolve(rmhip_ctx* ctx, rmhip_buf a, rmhip_buf b, const rmhip_linsolve_options_t* opts, rmhip_buf* out,
                   double* reciprocal_condition) {
    CTX_OR_FAIL(ctx);
    ScopedTimer timer(&c->tel.linsolve_count, &c->tel.linsolve_ns);
    return count_fallback(c, linsolve_impl(ctx, c, a, b, opts, out, reciprocal_condition), "linsolve:unsupported", "linsolve:singular");
}

static int linsolve_impl(rmhip_ctx* ctx, Context* c, rmhip_buf a, rmhip_buf b, const rmhip_linsolve_options_t* opts, rmhip_buf* out,
                         double* reciprocal_condition) {
    if (!out || !opts) return fail(RMHIP_ERR_INVALID, "linsolve: null argument");
    Buffer ab, bb;
    RMHIP_TRY(c->get(a, &ab));
    RMHIP_TRY(c->get(b, &bb));
    if (ab.shape.size() > 2 || bb.shape.size() > 2) return fail(RMHIP_ERR_UNSUPPORTED, "linsolve: only 2D supported");
    std::vector<size_t> as = normalize_matrix_shape(ab.shape);
    if (ab.numel == 1 || bb.numel == 1)  // linsolve.rs:408-412: scalar operands stay on the host path
        return fail(RMHIP_ERR_UNSUPPORTED, "linsolve: scalar operands use the CPU path");
    bool lower = opts->lower != 0, upper = opts->upper != 0;
    const double* A = ab.data();
    std::shared_ptr<Allocation> at;
    if (opts->transposed) {  // linsolve.rs:698-705: materialise A' and swap the triangle hints
        RMHIP_TRY(c->alloc_device(ab.numel ? ab.numel : 1, &at));
        RMHIP_TRY(transpose_device(c, ab.data(), as[0], as[0], as[1], at->ptr, as[1]));
        std::swap(as[0], as[1]);
        A = at->ptr;
        if (lower || upper) std::swap(lower, upper);
    }
    // normalize_rhs_tensor (linsolve.rs:972-984): a rank-1 rhs of the right length is a column
    std::vector<size_t> bs = normalize_matrix_shape(bb.shape);
    if (bs[0] != as[0]) {
        if (bb.shape.size() == 1 && bb.shape[0] == as[0]) bs = {as[0], 1};
        else return fail(RMHIP_ERR_SHAPE, "linsolve: Matrix dimensions must agree.");
    }
    const size_t n = as[0], nrhs = bs[1];
    if ((lower || upper) && as[0] != as[1]) return fail(RMHIP_ERR_SHAPE, "linsolve: triangular solves need a square matrix");
    if (!(lower || upper)) {
        if (opts->need_rcond || opts->has_rcond)
            return fail(RMHIP_ERR_UNSUPPORTED, "linsolve: rcond of a general matrix needs its singular values (CPU path)");
        if (as[0] != as[1]) {  // full-rank rectangular system: least squares / minimum norm as rmhip_mldivide (linsolve.rs:933-970 is the SVD solve)
            int lrc = lstsq_full_rank(ctx, c, A, as[0], as[1], bb.data(), bs[1], out);
            if (lrc == RMHIP_ERR_UNSUPPORTED && as[0] && as[1] && bs[1]) lrc = svd_fallback(ctx, c, lrc, A, as[0], as[1], bb.data(), bs[1], out);
            if (at) (void)hipStreamSynchronize(c->stream);  // the transposed copy is released on return
            if (!lrc && reciprocal_condition) *reciprocal_condition = std::numeric_limits<double>::quiet_NaN();
            return lrc;
        }
    }
    if (n == 0) return fail(RMHIP_ERR_UNSUPPORTED, "linsolve: empty system");
    double rcond = std::numeric_limits<double>::quiet_NaN();
    Buffer ob;
    rmhip_buf oid = 0;
    const size_t oshape[2] = {n, nrhs};
    int rc = RMHIP_OK;
    if (lower || upper) {
        double mn = 0.0, mx = 0.0;
        size_t zeros = 0;
        RMHIP_TRY(diag_stats_device(c, A, n, n, &mn, &mx, &zeros));
        if (zeros) return fail(RMHIP_ERR_SINGULAR, "linsolve: matrix is singular to working precision.");
        rcond = mx == 0.0 ? 0.0 : mn / mx;
        if (opts->has_rcond && rcond < opts->rcond)
            return fail(RMHIP_ERR_SINGULAR, "linsolve: matrix is singular to working precision.");
        RMHIP_TRY(c->new_buffer(oshape, 2, &oid, &ob));
        hipError_t e = hipMemcpyAsync(ob.data(), bb.data(), sizeof(double) * n * nrhs, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "linsolve: %s", hipGetErrorString(e));
        if (!rc)
            rc = lower ? trsm_lower_nonunit_device(c, A, n, n, ob.data(), n, nrhs) : trsm_upper_device(c, A, n, n, ob.data(), n, nrhs);
    } else {
        const size_t np = lu_pad_rows(n);
        const size_t ldw = lu_padded_ld(np);
        std::shared_ptr<Allocation> work;
        std::shared_ptr<Allocation> perm_mem;
        int* perm = nullptr;
        int info = 0;
        const char* fe = std::getenv("RMHIP_LU_FAST");
        const bool small = small_solve_applies(n, nrhs) && !(fe && fe[0] == '0') && !std::getenv("RMHIP_NO_SMALL_SOLVE");
        if (small) {  // one launch (small_solve.hip), the same kernel and therefore the same bits as mldivide's
            rc = c->new_buffer(oshape, 2, &oid, &ob);
            double mn = 0.0, mx = 0.0;
            size_t bad = 0;
            if (!rc) rc = small_solve_device(c, A, bb.data(), n, nrhs, ob.data(), &mn, &mx, &bad);
            if (!rc && bad > 0) {
                rmhip_free(ctx, oid);
                oid = 0;
                info = (int)bad;
            } else if (!rc) {
                c->lu_fast_count++;
                c->lu_last_growth = 0.0;
            }
        } else {
            RMHIP_TRY(c->alloc_device(ldw * np, &work));
            RMHIP_TRY(c->alloc_device((np + 2) / 2 + 1, &perm_mem));
            perm = (int*)perm_mem->ptr;
            rc = lu_copy_and_factor(c, A, n, n, work->ptr, ldw, perm, &info, true, np);
        }
        if (!rc && info > 0) {
            rc = fail(RMHIP_ERR_SINGULAR, "linsolve: %d pivot(s) <= 1e-12; use the CPU SVD path", info);
            if (nrhs) {
                rc = svd_fallback(ctx, c, rc, A, n, n, bb.data(), nrhs, &oid);
                if (at) (void)hipStreamSynchronize(c->stream);
                if (rc) return rc;
                *out = oid;
                if (reciprocal_condition) *reciprocal_condition = std::numeric_limits<double>::quiet_NaN();
                return RMHIP_OK;
            }
        }
        if (!rc && !small) rc = c->new_buffer(oshape, 2, &oid, &ob);
        if (!rc && !small) rc = lu_solve_padded(c, work->ptr, n, np, ldw, perm, bb.data(), nrhs, ob.data());
    }
    if (at) (void)hipStreamSynchronize(c->stream);  // the transposed copy is released on return
    if (rc) {
        if (oid) rmhip_free(ctx, oid);
        return rc;
    }
    *out = oid;
    if (reciprocal_condition) *reciprocal_condition = rcond;
    return RMHIP_OK;
}

// ---- block-level building blocks (views) ----------------------------------------------------------
namespace {
struct ViewPtr {
    Buffer buf;
    double* ptr = nullptr;
    size_t ld = 0, rows = 0, cols = 0;
};
// write: the block is updated in place - lazy views of the same storage held under other handles are materialised first
int resolve_view(Context* c, const rmhip_view_t* v, ViewPtr* out, bool write = false) {
    if (!v) return fail(RMHIP_ERR_INVALID, "null view");
    RMHIP_TRY(c->get_raw(v->buf, &out->buf));
    if (write) RMHIP_TRY(c->detach_views_of(v->buf));
    if (out->buf.dtype != DT_F64)  // in-place block updates cannot go through a widened temporary
        return fail(RMHIP_ERR_UNSUPPORTED, "block views address f64 storage; this buffer is f32 (precision-32 provider)");
    RMHIP_TRY(c->get(v->buf, &out->buf));
    const std::vector<size_t> s = normalize_matrix_shape(out->buf.shape);
    if (s.size() != 2) return fail(RMHIP_ERR_UNSUPPORTED, "view: only 2D buffers");
    if (v->row_off + v->rows > s[0] || v->col_off + v->cols > s[1])
        return fail(RMHIP_ERR_SHAPE, "view [%zu+%zu, %zu+%zu] exceeds buffer %zux%zu", v->row_off, v->rows, v->col_off, v->cols, s[0], s[1]);
    out->ld = s[0];
    out->rows = v->rows;
    out->cols = v->cols;
    out->ptr = out->buf.data() + v->row_off + v->col_off * s[0];
    return RMHIP_OK;
}
}  // namespace

int rmhip_blk_copy(rmhip_ctx* ctx, const rmhip_view_t* src, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    ViewPtr v;
    RMHIP_TRY(resolve_view(c, src, &v));
    Buffer ob;
    const size_t oshape[2] = {v.rows, v.cols};
    RMHIP_TRY(c->new_buffer(oshape, 2, out, &ob));
    if (v.rows && v.cols)
        RMHIP_HIP_CHECK(hipMemcpy2DAsync(ob.data(), v.rows * sizeof(double), v.ptr, v.ld * sizeof(double), v.rows * sizeof(double),
                                         v.cols, hipMemcpyDeviceToDevice, c->stream));
    return RMHIP_OK;
}

int rmhip_blk_assign(rmhip_ctx* ctx, const rmhip_view_t* dst, rmhip_buf src) {
    CTX_OR_FAIL(ctx);
    ViewPtr v;
    RMHIP_TRY(resolve_view(c, dst, &v, true));
    Buffer sb;
    RMHIP_TRY(c->get(src, &sb));
    if (sb.numel != v.rows * v.cols) return fail(RMHIP_ERR_SHAPE, "blk_assign: source has %zu elements, view %zux%zu", sb.numel, v.rows, v.cols);
    if (v.rows && v.cols)
        RMHIP_HIP_CHECK(hipMemcpy2DAsync(v.ptr, v.ld * sizeof(double), sb.data(), v.rows * sizeof(double), v.rows * sizeof(double),
                                         v.cols, hipMemcpyDeviceToDevice, c->stream));
    return RMHIP_OK;
}

int rmhip_blk_gemm(rmhip_ctx* ctx, double alpha, const rmhip_view_t* a, const rmhip_view_t* b, double beta,
                   const rmhip_view_t* cv) {
    CTX_OR_FAIL(ctx);
    ViewPtr va, vb, vc;
    RMHIP_TRY(resolve_view(c, a, &va));
    RMHIP_TRY(resolve_view(c, b, &vb));
    RMHIP_TRY(resolve_view(c, cv, &vc, true));
    if (va.cols != vb.rows || vc.rows != va.rows || vc.cols != vb.cols)
        return fail(RMHIP_ERR_SHAPE, "blk_gemm: %zux%zu * %zux%zu -> %zux%zu", va.rows, va.cols, vb.rows, vb.cols, vc.rows, vc.cols);
    if (va.cols == 0) return RMHIP_OK;
    return launch_dgemm(c, va.rows, vb.cols, va.cols, alpha, va.ptr, va.ld, vb.ptr, vb.ld, beta, vc.ptr, vc.ld);
}

int rmhip_blk_trsm(rmhip_ctx* ctx, int upper, const rmhip_view_t* t, const rmhip_view_t* b) {
    CTX_OR_FAIL(ctx);
    ViewPtr vt, vb;
    RMHIP_TRY(resolve_view(c, t, &vt));
    RMHIP_TRY(resolve_view(c, b, &vb, true));
    if (upper == 2) {
        // B <- B U^-1 (the multipliers of a row block against a factored diagonal tile: L21 = A21 U11^-1).  X U = B is U' X' = B': both
        // operands are transposed into temporaries (k_transpose, 64 x 64 LDS tiles), U' is lower with a stored diagonal - the
        // kernels of `linsolve`'s LT hint - and the solution is transposed back in place.  O(rows w) extra traffic around O(rows w^2) work.
        if (vt.rows != vt.cols || vb.cols != vt.rows)
            return fail(RMHIP_ERR_SHAPE, "blk_trsm (right): triangle %zux%zu vs block %zux%zu", vt.rows, vt.cols, vb.rows, vb.cols);
        const size_t w = vt.rows, m = vb.rows;
        if (w == 0 || m == 0) return RMHIP_OK;
        std::shared_ptr<Allocation> tt, bt;
        RMHIP_TRY(c->alloc_device(w * w, &tt));
        RMHIP_TRY(c->alloc_device(w * m, &bt));
        RMHIP_TRY(transpose_device(c, vt.ptr, vt.ld, w, w, tt->ptr, w));
        RMHIP_TRY(transpose_device(c, vb.ptr, vb.ld, m, w, bt->ptr, w));
        RMHIP_TRY(trsm_lower_nonunit_device(c, tt->ptr, w, w, bt->ptr, w, m));
        RMHIP_TRY(transpose_device(c, bt->ptr, w, w, m, vb.ptr, vb.ld));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // the temporaries go back to the pool on return
        return RMHIP_OK;
    }
    if (vt.rows != vt.cols || vb.rows != vt.rows) return fail(RMHIP_ERR_SHAPE, "blk_trsm: triangle %zux%zu vs rhs %zux%zu", vt.rows, vt.cols, vb.rows, vb.cols);
    return upper ? trsm_upper_device(c, vt.ptr, vt.ld, vt.rows, vb.ptr, vb.ld, vb.cols)
                 : trsm_lower_unit_device(c, vt.ptr, vt.ld, vt.rows, vb.ptr, vb.ld, vb.cols);
}

int rmhip_blk_lu(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf* ipiv_out, int* info) {
    CTX_OR_FAIL(ctx);
    if (!ipiv_out) return fail(RMHIP_ERR_INVALID, "null ipiv_out");
    ViewPtr va;
    RMHIP_TRY(resolve_view(c, a, &va, true));
    // the interchange vector is written on the device by the factorisation itself (round 6: it used to travel device -> host -> device
    // with a stream drain at each end, and rmhip_blk_swap_rows fetched it back again)
    const size_t kmin = va.rows < va.cols ? va.rows : va.cols;
    const size_t oshape[2] = {kmin, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(oshape, 2, ipiv_out, &ob));
    int inf = 0;
    int frc = RMHIP_LU_GROWTH;
    if (c->blk_lu_solve_path && va.rows > 0 && va.cols > 0) {
        // the solve path's kernels (k_rp_top / k_rp_below_mfma / matrix-core solves): the block is saved first - a multiplier beyond
        // the bound clobbers it - and restored for the grid-wide rule
        std::shared_ptr<Allocation> keep;
        frc = c->alloc_device(va.rows * va.cols, &keep);
        if (frc == RMHIP_OK) {
            hipError_t e = hipMemcpy2DAsync(keep->ptr, va.rows * sizeof(double), va.ptr, va.ld * sizeof(double), va.rows * sizeof(double), va.cols,
                                            hipMemcpyDeviceToDevice, c->stream);
            frc = e == hipSuccess ? lu_factor_device(c, va.ptr, va.rows, va.cols, va.ld, nullptr, &inf, nullptr, 1, ob.data())
                                  : fail(RMHIP_ERR_HIP, "blk_lu: saving the block: %s", hipGetErrorString(e));
            if (frc == RMHIP_LU_GROWTH || frc == RMHIP_LU_RETRY) {
                e = hipMemcpy2DAsync(va.ptr, va.ld * sizeof(double), keep->ptr, va.rows * sizeof(double), va.rows * sizeof(double), va.cols,
                                     hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                frc = e == hipSuccess ? RMHIP_LU_GROWTH : fail(RMHIP_ERR_HIP, "blk_lu: restoring the block: %s", hipGetErrorString(e));
            }
        }
    }
    if (frc == RMHIP_LU_GROWTH) frc = lu_factor_device(c, va.ptr, va.rows, va.cols, va.ld, nullptr, &inf, nullptr, 0, ob.data());
    if (frc == RMHIP_LU_RETRY)  // in place: the block is clobbered and there is no copy to restart from
        frc = fail(RMHIP_ERR_HIP, "blk_lu: panel workgroups were not co-resident (device shared?); the block is invalid");
    if (frc != RMHIP_OK) {
        rmhip_free(ctx, *ipiv_out);  // do not leak the interchange vector on the error paths
        *ipiv_out = 0;
        return frc;
    }
    if (info) *info = inf;
    return RMHIP_OK;
}

int rmhip_blk_lu_deferred(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf guard, rmhip_buf* ipiv_out) {
    CTX_OR_FAIL(ctx);
    if (!ipiv_out) return fail(RMHIP_ERR_INVALID, "null ipiv_out");
    ViewPtr va;
    RMHIP_TRY(resolve_view(c, a, &va, true));
    Buffer gb;
    RMHIP_TRY(c->get(guard, &gb));
    if (gb.numel != 1 || gb.dtype != DT_F64) return fail(RMHIP_ERR_INVALID, "blk_lu_deferred: the guard is a 1 x 1 f64 tensor");
    const size_t kmin = va.rows < va.cols ? va.rows : va.cols;
    if (kmin == 0) return fail(RMHIP_ERR_INVALID, "blk_lu_deferred: empty view");
    const size_t oshape[2] = {kmin, 1};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(oshape, 2, ipiv_out, &ob));
    int inf = 0;
    // solve-path panel kernels, no saved copy, no host read: the status lands in *guard (lu.hip, deferred form)
    int frc = lu_factor_device(c, va.ptr, va.rows, va.cols, va.ld, nullptr, &inf, nullptr, 1, ob.data(), gb.data());
    if (frc == RMHIP_LU_GROWTH || frc == RMHIP_LU_RETRY)  // (only when the deferred form was not taken: the ordinary checks ran and refused)
        frc = fail(RMHIP_ERR_GROWTH, "blk_lu_deferred: the panel was refused (multiplier bound or panel placement); the block is invalid");
    if (frc != RMHIP_OK) {
        rmhip_free(ctx, *ipiv_out);
        *ipiv_out = 0;
    }
    return frc;
}

int rmhip_blk_swap_rows(rmhip_ctx* ctx, const rmhip_view_t* a, rmhip_buf ipiv) {
    CTX_OR_FAIL(ctx);
    ViewPtr va;
    RMHIP_TRY(resolve_view(c, a, &va, true));
    Buffer pb;
    RMHIP_TRY(c->get(ipiv, &pb));
    // composed and applied on the device (no stream drain, no hipMalloc / hipFree); RMHIP_BLK_SWAP_DEVICE=0 or a view too tall for
    // the LDS map: the host composition, which also REPORTS an out-of-range target (the device form skips it)
    static const bool dev_path = !(std::getenv("RMHIP_BLK_SWAP_DEVICE") && std::getenv("RMHIP_BLK_SWAP_DEVICE")[0] == '0');
    if (dev_path) {
        const int rc = lu_swap_rows_from_device(c, va.ptr, va.ld, va.rows, va.cols, pb.data(), pb.numel);
        if (rc != RMHIP_ERR_UNSUPPORTED) return rc;
    }
    std::vector<double> host(pb.numel);
    if (pb.numel) {
        RMHIP_HIP_CHECK(hipMemcpyAsync(host.data(), pb.data(), pb.numel * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    std::vector<int> piv(host.size());
    for (size_t k = 0; k < host.size(); ++k) {
        if (!(host[k] >= 0.0) || host[k] >= (double)va.rows) return fail(RMHIP_ERR_INVALID, "swap_rows: pivot %zu out of range", k);
        piv[k] = (int)host[k];
    }
    return lu_swap_rows_device(c, va.ptr, va.ld, va.cols, piv);
}

int rmhip_set_rng_state(rmhip_ctx* ctx, uint64_t state) {
    CTX_OR_FAIL(ctx);
    c->rng_state = state;
    return RMHIP_OK;
}

int rmhip_set_lazy_random(rmhip_ctx* ctx, int enabled, size_t min_numel) {
    CTX_OR_FAIL(ctx);
    c->lazy_randn = enabled != 0;
    if (min_numel) c->lazy_randn_min = min_numel;
    return RMHIP_OK;
}

int rmhip_lazy_random_stats(rmhip_ctx* ctx, uint64_t* created, uint64_t* fused, uint64_t* materialised) {
    CTX_OR_FAIL(ctx);
    if (created) *created = c->lazy_randn_created;
    if (fused) *fused = c->lazy_randn_fused;
    if (materialised) *materialised = c->lazy_randn_materialised;
    return RMHIP_OK;
}

int rmhip_get_rng_state(rmhip_ctx* ctx, uint64_t* state) {
    CTX_OR_FAIL(ctx);
    if (!state) return fail(RMHIP_ERR_INVALID, "null state");
    *state = c->rng_state;
    return RMHIP_OK;
}

int rmhip_rng_seed(rmhip_ctx* ctx, uint64_t seed) {  // mix_seed, random.rs:128-141
    CTX_OR_FAIL(ctx);
    uint64_t s;
    if (seed == 0) s = 0x9e3779b97f4a7c15ULL;
    else {
        uint64_t z = seed + 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s = z ^ (z >> 31);
        if (s == 0) s = 0x9e3779b97f4a7c15ULL;
    }
    c->rng_state = s;
    return RMHIP_OK;
}

int rmhip_random_uniform(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Buffer ob;
    int rc;
    if (c->precision == 32) {
        RMHIP_TRY(c->new_buffer_f32(shape, rank, out, &ob));
        rc = launch_rng_uniform_f32(c, c->rng_state, ob.data_f32(), ob.numel);
    } else {
        RMHIP_TRY(c->new_buffer(shape, rank, out, &ob));
        rc = launch_rng_uniform(c, c->rng_state, ob.data(), ob.numel);
    }
    if (rc) {
        rmhip_free(ctx, *out);
        return rc;
    }
    c->rng_state = lcg_advance(c->rng_state, ob.numel);
    return RMHIP_OK;
}

int rmhip_random_normal(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Buffer ob;
    int rc;
    if (c->precision != 32 && c->lazy_randn && out && (rank == 0 || shape) && shape_numel(shape, rank) >= c->lazy_randn_min &&
        shape_numel(shape, rank) >= 2) {
        // Lazy record: no storage, no launch.  A streaming fused elementwise kernel that reads it generates the normals in registers
        // (8 B per sample never written and never read back); anything else materialises it under this id with k_rng_normal on the
        // recorded state.  The stream advances now, exactly as for an eager call.
        Buffer b;
        b.shape.assign(shape, shape + rank);
        b.numel = shape_numel(shape, rank);
        b.rng_lazy = true;
        b.rng_state = c->rng_state;
        const size_t numel = b.numel;
        RMHIP_TRY(c->register_buffer(std::move(b), out));
        c->lazy_randn_created++;
        c->rng_state = lcg_advance(c->rng_state, 2 * ((numel + 1) / 2));
        return RMHIP_OK;
    }
    if (c->precision == 32) {
        RMHIP_TRY(c->new_buffer_f32(shape, rank, out, &ob));
        rc = launch_rng_normal_f32(c, c->rng_state, ob.data_f32(), ob.numel);
    } else {
        RMHIP_TRY(c->new_buffer(shape, rank, out, &ob));
        rc = launch_rng_normal(c, c->rng_state, ob.data(), ob.numel);
    }
    if (rc) {
        rmhip_free(ctx, *out);
        return rc;
    }
    c->rng_state = lcg_advance(c->rng_state, 2 * ((ob.numel + 1) / 2));  // whole pairs are consumed
    return RMHIP_OK;
}

namespace {
// one transformed draw per element (or whole Box-Muller pairs): allocate in the provider's storage type, launch, advance the stream
int random_dist(rmhip_ctx* ctx, Context* c, const size_t* shape, size_t rank, rmhip_buf* out, bool pairs, bool consumes,
                const std::function<int(double*, float*, size_t)>& launch) {
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ob;
    int rc;
    if (c->precision == 32) {
        RMHIP_TRY(c->new_buffer_f32(shape, rank, out, &ob));
        rc = launch((double*)nullptr, ob.data_f32(), ob.numel);
    } else {
        RMHIP_TRY(c->new_buffer(shape, rank, out, &ob));
        rc = launch(ob.data(), (float*)nullptr, ob.numel);
    }
    if (rc) {
        rmhip_free(ctx, *out);
        return rc;
    }
    if (consumes) c->rng_state = lcg_advance(c->rng_state, pairs ? 2 * ((ob.numel + 1) / 2) : ob.numel);
    return RMHIP_OK;
}
}  // namespace

int rmhip_random_unifrnd(rmhip_ctx* ctx, double a, double b, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    return random_dist(ctx, c, shape, rank, out, false, true,
                       [&](double* o64, float* o32, size_t n) { return launch_rng_unifrnd(c, c->rng_state, a, b, o64, o32, n); });
}

int rmhip_random_exponential(rmhip_ctx* ctx, double mu, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    return random_dist(ctx, c, shape, rank, out, false, true,
                       [&](double* o64, float* o32, size_t n) { return launch_rng_exponential(c, c->rng_state, mu, o64, o32, n); });
}

int rmhip_random_normrnd(rmhip_ctx* ctx, double mu, double sigma, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    return random_dist(ctx, c, shape, rank, out, true, true,
                       [&](double* o64, float* o32, size_t n) { return launch_rng_normrnd(c, c->rng_state, mu, sigma, o64, o32, n); });
}

int rmhip_random_integer_range(rmhip_ctx* ctx, long long lower, long long upper, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    // simple_provider.rs:3689-3698: an empty range, or one of more than 2^53 values (not exactly representable), is an error
    if (lower > upper) return fail(RMHIP_ERR_INVALID, "random_integer_range: lower bound must be <= upper bound");
    const __int128 span128 = (__int128)upper - (__int128)lower + 1;
    if (span128 > ((__int128)1 << 53)) return fail(RMHIP_ERR_INVALID, "random_integer_range: integer range exceeds 2^53 and cannot be represented exactly");
    const unsigned long long span = (unsigned long long)span128;
    if (span == 1) {  // one value: no draws are consumed (simple_provider.rs:3703-3704)
        if (!out) return fail(RMHIP_ERR_INVALID, "null out");
        Buffer ob;
        RMHIP_TRY(c->new_buffer(shape, rank, out, &ob));  // (narrowed on return at precision 32, as rmhip_fill)
        const int rc = launch_fill(c, ob.data(), ob.numel, (double)lower);
        if (rc) rmhip_free(ctx, *out);
        return rc;
    }
    return random_dist(ctx, c, shape, rank, out, false, true,
                       [&](double* o64, float* o32, size_t n) { return launch_rng_integer_range(c, c->rng_state, lower, span, o64, o32, n); });
}

int rmhip_stochastic_evolution(rmhip_ctx* ctx, rmhip_buf state, double drift, double scale, uint32_t steps, rmhip_buf* out) {
    return rmhip_stochastic_evolution_sharded(ctx, state, drift, scale, steps, 0, out);
}

int rmhip_stochastic_evolution_sharded(rmhip_ctx* ctx, rmhip_buf state, double drift, double scale, uint32_t steps,
                                       uint64_t draws_per_step, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer sb;
    bool f32 = c->precision == 32;
    RMHIP_TRY(get_operand(c, state, &sb, &f32));
    Buffer ob;
    if (f32) RMHIP_TRY(c->new_buffer_f32(sb.shape.data(), sb.shape.size(), out, &ob));
    else RMHIP_TRY(c->new_buffer(sb.shape.data(), sb.shape.size(), out, &ob));
    if (sb.numel == 0) return RMHIP_OK;
    int rc = RMHIP_OK;
    if (steps == 0) {  // stochastic_evolution.rs:16-18: nothing drawn, state unchanged
        hipError_t e = hipMemcpyAsync(ob.data(), sb.data(), (f32 ? sizeof(float) : sizeof(double)) * sb.numel, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "stochastic_evolution: %s", hipGetErrorString(e));
    } else {
        const uint64_t local = 2ULL * ((sb.numel + 1) / 2);
        if (draws_per_step && draws_per_step < local)
            rc = fail(RMHIP_ERR_INVALID, "stochastic_evolution: draws_per_step %llu < the shard's own %llu",
                      (unsigned long long)draws_per_step, (unsigned long long)local);
        if (!rc)
            rc = f32 ? launch_stochastic_evolution_f32(c, c->rng_state, sb.data_f32(), ob.data_f32(), sb.numel, drift, scale, steps, draws_per_step)
                     : launch_stochastic_evolution(c, c->rng_state, sb.data(), ob.data(), sb.numel, drift, scale, steps, draws_per_step);
        if (!rc) c->rng_state = lcg_advance(c->rng_state, (uint64_t)steps * (draws_per_step ? draws_per_step : local));
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

}  // extern "C"
