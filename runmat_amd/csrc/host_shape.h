// host_shape.h -- host-side shape logic of the operator entry points (no GPU dependency; unit-tested on the CPU by
// tests/cpp/broadcast_prep_check.cpp): broadcast stride preparation and dimension collapsing for the elementwise
// kernels.  Mirrors the provider duty of backend/wgpu/provider/ops/elementwise.rs:1655-1697.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace rmhip {

// Front-pad `shape` to `rank` (broadcast.rs:108-115, elementwise.rs:1681-1687) and derive strides
// with 0 on broadcast (extent 1) dims.  Returns false if the shape cannot broadcast to `out`.
inline bool padded_strides(const std::vector<size_t>& shape_in, const size_t* out, size_t rank, std::vector<uint64_t>* strides) {
    // MATLAB shapes carry implicit trailing singletons ([n,1] == [n]): an operand of higher rank
    // than the request is first stripped of trailing, then leading, extent-1 dims.
    std::vector<size_t> shape = shape_in;
    while (shape.size() > rank && !shape.empty() && shape.back() == 1) shape.pop_back();
    while (shape.size() > rank && !shape.empty() && shape.front() == 1) shape.erase(shape.begin());
    if (shape.size() > rank) return false;
    const size_t pad = rank - shape.size();
    strides->assign(rank, 0);
    uint64_t s = 1;
    for (size_t d = 0; d < rank; ++d) {
        const size_t ext = d < pad ? 1 : shape[d - pad];
        if (ext != 1 && ext != out[d]) return false;
        (*strides)[d] = ext == 1 ? 0 : s;
        s *= ext;
    }
    return true;
}

// Collapse dims: drop extent-1 dims, merge dim d+1 into d when every operand is contiguous across
// the boundary (stride[d+1] == stride[d]*shape[d]) or broadcast on both (0 and 0).
inline void collapse(std::vector<uint64_t>* shape, std::vector<std::vector<uint64_t>>* strides) {
    std::vector<uint64_t> ns;
    std::vector<std::vector<uint64_t>> nst(strides->size());
    for (size_t d = 0; d < shape->size(); ++d) {
        if ((*shape)[d] == 1) continue;
        bool merged = false;
        if (!ns.empty()) {
            bool ok = true;
            for (size_t k = 0; k < strides->size(); ++k) {
                const uint64_t prev = nst[k].back(), cur = (*strides)[k][d];
                if (!((prev == 0 && cur == 0) || (prev != 0 && cur == prev * ns.back()))) {
                    ok = false;
                    break;
                }
            }
            if (ok) {
                ns.back() *= (*shape)[d];
                merged = true;
            }
        }
        if (!merged) {
            ns.push_back((*shape)[d]);
            for (size_t k = 0; k < strides->size(); ++k) nst[k].push_back((*strides)[k][d]);
        }
    }
    if (ns.empty()) {
        ns.push_back(1);
        for (size_t k = 0; k < strides->size(); ++k) nst[k].push_back(0);
    }
    *shape = ns;
    *strides = nst;
}

}  // namespace rmhip
