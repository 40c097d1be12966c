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
    int unroll = 4;        // independent 16-byte vectors in flight per thread (fast path)
    int block = 256;
    int blocks_per_cu = 8; // grid cap = blocks_per_cu * CUs (grid-stride beyond that)
    int nontemporal = 1;   // non-temporal loads/stores on the streaming fast path
    static EwTuning from_env();
};

struct FusedKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn_fast = nullptr;    // elementwise: all inputs full-size or scalar, 16 B vectors
    hipFunction_t fn_fast1 = nullptr;   // elementwise: same, 8 B accesses (unaligned external memory)
    hipFunction_t fn_bcast = nullptr;   // elementwise: general broadcast, rank <= 8
    hipFunction_t fn_contig = nullptr;  // reduction kernel A
    hipFunction_t fn_strided = nullptr; // reduction kernel B
    hipFunction_t fn_final = nullptr;   // reduction finalize
    int n_inputs = 0, n_outputs = 0;
    EwTuning tuning;
    ~FusedKernel();
};

// Source generation (no GPU needed).
std::string generate_elementwise_source(const ElementwiseProgram& p, const EwTuning& t, unsigned scalar_mask);
std::string generate_reduction_source(const ReductionProgram& p);

// hipRTC compile for gfx950; on failure returns nonzero and sets the error string (with the log).
int compile_to_code_object(const std::string& source, std::vector<char>* code);

// Cached lookups (compile on miss). `scalar_mask` bit k set => input k is a 1-element tensor.
int get_elementwise_kernel(Context* c, const ElementwiseProgram& p, unsigned scalar_mask,
                           std::shared_ptr<FusedKernel>* out);
int get_reduction_kernel(Context* c, const ReductionProgram& p, std::shared_ptr<FusedKernel>* out);

uint64_t fnv1a(const std::string& s);

}  // namespace rmhip
