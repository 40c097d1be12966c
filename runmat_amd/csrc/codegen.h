// codegen.h -- expression tape -> HIP source -> hipRTC code object -> cached hipFunction.
// Plays the role of the reference's pipeline cache keyed by shader hash
// (crates/runmat-accelerate/src/backend/wgpu/provider/ops/elementwise.rs:1608-1626).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "common.h"
#include "wgsl_front.h"

namespace rmhip {

struct EwTuning {
    int unroll = 0;        // independent 16-byte vectors in flight per thread; 0 = 1 (see unroll_for)
    int block = 1024;      // streaming fast path; interleaved A/B at 8192^2 f64 (scripts/tune_ew_ab.py), GB/s for block
                           // 256 / 512 / 1024: sin(A).*B+C 5325 / 5739 / 5989, A.*B+C 4538 / 5270 / 5444, A+B 4898 / 5562 /
                           // 5748, copy 5311 / 5832 / 6047, sin(A) 5649 / 5856 / 6098
    int bcast_block = 256; // general broadcast path: a block spans bcast_elems * bcast_block elements of dim 0
    int bcast_elems = 4;   // elements per thread there (their loads overlap).  Interleaved A/B (scripts/tune_bcast_ab.py),
                           // GB/s for `A - row` / `sin(A).*row + col` at 8192^2: 256x1 4422 / 3389, 256x2 5689 / 4218,
                           // 256x4 5885 / 4798, 1024x1 3605 / 2879, 1024x4 5648 / 4381, 256x8 5796 / 4724
    int blocks_per_cu = 16; // grid cap = blocks_per_cu * CUs (grid-stride beyond that)
    int nt_load = 1;       // non-temporal loads on the streaming fast path
    int nt_store = 1;      // non-temporal stores
    int chunked = 0;       // 1: each block walks one contiguous chunk instead of a grid-stride loop
    static EwTuning from_env();
    // One 16-byte vector per stream per thread.  Sequential sweeps (scripts/tune_ew.py) once suggested unroll 4 for
    // pure-arithmetic bodies; interleaved A/B runs (order effects on this hardware are as large as the differences)
    // show unroll 1 at least as fast for every body at every block size (e.g. block 1024: 5989 vs 5050 GB/s).
    int unroll_for(int n_streamed_inputs, bool heavy_math) const;
};

struct FusedKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn_fast = nullptr;    // elementwise: all inputs full-size or scalar, 16 B vectors
    hipFunction_t fn_fast1 = nullptr;   // elementwise: same, 8 B accesses (unaligned external memory)
    hipFunction_t fn_bcast = nullptr;   // elementwise: general broadcast, rank <= 8
    hipFunction_t fn_bcast_flat = nullptr;  // the same for a short dim 0: threads over the flat output
    hipFunction_t fn_contig = nullptr;  // reduction kernel A
    hipFunction_t fn_contig2 = nullptr; // reduction kernel A over 16-byte vectors (even slices, aligned full-size inputs)
    hipFunction_t fn_strided = nullptr; // reduction kernel B
    hipFunction_t fn_strided2 = nullptr; // reduction kernel B over 16-byte vectors (even `pre` >= 512, aligned full-size inputs)
    hipFunction_t fn_final = nullptr;   // reduction finalize
    hipFunction_t fn_final_flat = nullptr;  // the same, one thread per slice (many slices, a handful of partials each)
    int n_inputs = 0, n_outputs = 0;
    EwTuning tuning;
    std::string key_text;  // what the cache key hashes: compared on every hit
    ~FusedKernel();
};

// True if the body calls libm-class functions or divides (register hungry: prefers occupancy).
bool program_is_heavy(const ElementwiseProgram& p);

// Source generation (no GPU needed).
// `f32`: tensors are stored as f32 (precision-32 contexts); the body still computes in f64.
// `rng_mask` bit k set => input k is a lazy random_normal operand (f64, streaming kernel only; Buffer::rng_lazy).
std::string generate_elementwise_source(const ElementwiseProgram& p, const EwTuning& t, unsigned scalar_mask, bool f32 = false, unsigned rng_mask = 0);
std::string generate_reduction_source(const ReductionProgram& p, bool f32 = false);

// hipRTC compile for gfx950; on failure returns nonzero and sets the error string (with the log).
int compile_to_code_object(const std::string& source, std::vector<char>* code);

// Cached lookups (compile on miss). `scalar_mask` bit k set => input k is a 1-element tensor.
int get_elementwise_kernel(Context* c, const ElementwiseProgram& p, unsigned scalar_mask, bool f32,
                           std::shared_ptr<FusedKernel>* out, unsigned rng_mask = 0);
int get_reduction_kernel(Context* c, const ReductionProgram& p, bool f32, std::shared_ptr<FusedKernel>* out);

uint64_t fnv1a(const std::string& s);

}  // namespace rmhip
