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

// Operand of a broadcasting elementwise launch: its logical shape and, for a lazy repmat view (common.h `Buffer::rep_base`),
// the extents of the base tensor the storage really holds (same rank as `shape`; empty for a plain tensor).
struct OperandDims {
    std::vector<size_t> shape;
    std::vector<size_t> base;
};

// Strides of every operand over a REFINED output shape, so that repmat views are read in place: an output dimension of
// extent E that a view tiles from a base extent b (1 < b < E) is split into [b, E / b] - column-major, coordinate
// c = c_lo + b * c_hi - and the view gets strides (s, 0) there while a plain operand gets (s, s * b); a pure broadcast
// (b == 1) is stride 0, as for an extent-1 operand.  Shapes are front-padded to `rank` exactly like padded_strides.
// Returns 0 on success; 1 when operand *bad does not broadcast to `out`; 2 when operand *bad tiles a dimension that
// another view already splits differently (the caller materialises it and retries).
inline int refined_strides(const std::vector<OperandDims>& ops, const size_t* out, size_t rank, std::vector<uint64_t>* rshape,
                           std::vector<std::vector<uint64_t>>* strides, size_t* bad) {
    std::vector<OperandDims> nd(ops.size());
    for (size_t k = 0; k < ops.size(); ++k) {
        std::vector<size_t> shape = ops[k].shape, base = ops[k].base;
        const bool view = !base.empty();
        while (shape.size() > rank && shape.back() == 1) {
            shape.pop_back();
            if (view) base.pop_back();
        }
        while (shape.size() > rank && shape.front() == 1) {
            shape.erase(shape.begin());
            if (view) base.erase(base.begin());
        }
        if (shape.size() > rank) {
            *bad = k;
            return 1;
        }
        const size_t pad = rank - shape.size();
        nd[k].shape.assign(rank, 1);
        for (size_t d = 0; d < shape.size(); ++d) nd[k].shape[pad + d] = shape[d];
        if (view) {
            nd[k].base.assign(rank, 1);
            for (size_t d = 0; d < base.size(); ++d) nd[k].base[pad + d] = base[d];
        }
    }
    std::vector<size_t> split(rank, 1);
    for (size_t k = 0; k < ops.size(); ++k)
        for (size_t d = 0; d < rank; ++d) {
            const size_t e = nd[k].shape[d];
            if (e != 1 && e != out[d]) {
                *bad = k;
                return 1;
            }
            if (nd[k].base.empty() || e == 1) continue;
            const size_t b = nd[k].base[d];
            if (b == e || b == 1) continue;
            if (split[d] == 1) split[d] = b;
            else if (split[d] != b) {
                *bad = k;
                return 2;
            }
        }
    rshape->clear();
    for (size_t d = 0; d < rank; ++d) {
        if (split[d] > 1) {
            rshape->push_back(split[d]);
            rshape->push_back(out[d] / split[d]);
        } else {
            rshape->push_back(out[d]);
        }
    }
    strides->assign(ops.size(), std::vector<uint64_t>());
    for (size_t k = 0; k < ops.size(); ++k) {
        std::vector<uint64_t>& st = (*strides)[k];
        const bool view = !nd[k].base.empty();
        uint64_t s = 1;
        for (size_t d = 0; d < rank; ++d) {
            const size_t e = nd[k].shape[d];
            const size_t b = view ? nd[k].base[d] : e;  // extent the storage holds along d
            uint64_t lo, hi;                            // strides of the two halves of a split dimension
            if (e == 1 || b == 1) lo = hi = 0;          // broadcast operand, or a view replicating one element
            else if (b == e) {
                lo = s;
                hi = s * split[d];
            } else {  // tiled: b == split[d]
                lo = s;
                hi = 0;
            }
            st.push_back(lo);
            if (split[d] > 1) st.push_back(hi);
            s *= b;
        }
    }
    return 0;
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
