// reduce_kernels.hip -- ahead-of-time reductions: reduce_sum / reduce_sum_dim / reduce_mean(_dim) /
// reduce_min / reduce_max / reduce_prod (crates/runmat-accelerate-api/src/lib.rs:2709-2721,
// 2743-2792, 2858-2883; reference semantics crates/runmat-accelerate/src/simple_provider.rs:
// 6728-6806 and the CPU sum_tensor, runtime/.../reduction/sum.rs:996-1079).
// HBM-bound: coalesced loads, per-lane f64 accumulators, wave64 __shfl_down tree, LDS across the
// four waves of a block, deterministic two-stage combine (skel_reduce.h).
#include "common.h"
#include "reduce_plan.h"
#include "skel_common.h"
#include "skel_reduce.h"

namespace rmhip {

struct IdentityVal {
    const double* __restrict__ x;
    __device__ __forceinline__ double operator()(rm_u64 idx) const { return x[idx]; }
};

template <int OP>
__global__ void __launch_bounds__(RM_ABLOCK) k_reduce_contig(const double* x, rm_u64 red, rm_u64 nslices,
                                                             rm_u64 nsplit, double* pv, double* pn) {
    IdentityVal f{x};
    rm_reduce_contig<OP>(f, red, nslices, nsplit, pv, pn);
}
template <int OP>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_strided(const double* x, rm_u64 pre, rm_u64 red, rm_u64 nsplit,
                                                              int tx, double* pv, double* pn) {
    IdentityVal f{x};
    rm_reduce_strided<OP>(f, pre, red, nsplit, tx, pv, pn);
}
template <int OP>
__global__ void __launch_bounds__(RM_RBLOCK) k_reduce_final(const double* pv, const double* pn, rm_u64 nslices,
                                                            rm_u64 nsplit, rm_u64 red, int mean, int omitnan,
                                                            double scale, double* out) {
    rm_reduce_finalize<OP>(pv, pn, nslices, nsplit, red, mean, omitnan, scale, out);
}

template <int OP>
static int run_reduce(Context* c, int mean, int nan_mode, const double* x, size_t pre, size_t red, size_t post,
                      double* out) {
    if (pre == 0 || post == 0) return RMHIP_OK;  // no output slices
    const ReducePlan p = plan_reduction(pre, red, post, c->num_cus);
    if (!p.valid) return fail(RMHIP_ERR_UNSUPPORTED, "reduce: geometry [%zu,%zu,%zu] exceeds launch limits", pre, red, post);
    const size_t nparts = (size_t)(p.nslices * p.nsplit);
    RMHIP_TRY(c->ensure_scratch(2 * nparts * sizeof(double)));
    double* pv = c->scratch;
    double* pn = c->scratch + nparts;
    if (p.contiguous)
        hipLaunchKernelGGL((k_reduce_contig<OP>), dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, x,
                           (rm_u64)red, (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else
        hipLaunchKernelGGL((k_reduce_strided<OP>), dim3(p.gx, p.gy, p.gz), dim3(RM_RBLOCK), 0, c->stream, x,
                           (rm_u64)pre, (rm_u64)red, (rm_u64)p.nsplit, p.tx, pv, pn);
    RMHIP_HIP_CHECK(hipGetLastError());
    const unsigned fb = (unsigned)ceil_div_u64(p.nslices, RM_RBLOCK / 64);
    hipLaunchKernelGGL((k_reduce_final<OP>), dim3(fb), dim3(RM_RBLOCK), 0, c->stream, pv, pn, (rm_u64)p.nslices,
                       (rm_u64)p.nsplit, (rm_u64)red, mean, nan_mode, 1.0, out);
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}

int launch_reduce_mid(Context* c, int op, int nan_mode, const double* x, size_t pre, size_t red, size_t post,
                      double* out) {
    switch (op) {
        case RMHIP_RSUM: return run_reduce<RM_RSUM>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RMEAN: return run_reduce<RM_RSUM>(c, 1, nan_mode, x, pre, red, post, out);
        case RMHIP_RMIN: return run_reduce<RM_RMIN>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RMAX: return run_reduce<RM_RMAX>(c, 0, nan_mode, x, pre, red, post, out);
        case RMHIP_RPROD: return run_reduce<RM_RPROD>(c, 0, nan_mode, x, pre, red, post, out);
        default: return fail(RMHIP_ERR_UNSUPPORTED, "reduce op %d not supported by provider", op);
    }
}

// ---- dot: the producer a.*b folded into the same skeleton (no temporary array) -------------------
struct ProductVal {
    const double* __restrict__ a;
    const double* __restrict__ b;
    __device__ __forceinline__ double operator()(rm_u64 idx) const { return a[idx] * b[idx]; }
};
__global__ void __launch_bounds__(RM_ABLOCK) k_dot_contig(const double* a, const double* b, rm_u64 red, rm_u64 nslices,
                                                          rm_u64 nsplit, double* pv, double* pn) {
    ProductVal f{a, b};
    rm_reduce_contig<RM_RSUM>(f, red, nslices, nsplit, pv, pn);
}
__global__ void __launch_bounds__(RM_RBLOCK) k_dot_strided(const double* a, const double* b, rm_u64 pre, rm_u64 red,
                                                           rm_u64 nsplit, int tx, double* pv, double* pn) {
    ProductVal f{a, b};
    rm_reduce_strided<RM_RSUM>(f, pre, red, nsplit, tx, pv, pn);
}

int launch_reduce_dot(Context* c, const double* a, const double* b, size_t pre, size_t red, size_t post, double* out) {
    if (pre == 0 || post == 0) return RMHIP_OK;  // no output slices
    const ReducePlan p = plan_reduction(pre, red, post, c->num_cus);
    if (!p.valid) return fail(RMHIP_ERR_UNSUPPORTED, "dot: geometry [%zu,%zu,%zu] exceeds launch limits", pre, red, post);
    const size_t nparts = (size_t)(p.nslices * p.nsplit);
    RMHIP_TRY(c->ensure_scratch(2 * nparts * sizeof(double)));
    double* pv = c->scratch;
    double* pn = c->scratch + nparts;
    if (p.contiguous)
        hipLaunchKernelGGL(k_dot_contig, dim3(p.gx, p.gy, p.gz), dim3(p.tx), 0, c->stream, a, b, (rm_u64)red,
                           (rm_u64)p.nslices, (rm_u64)p.nsplit, pv, pn);
    else
        hipLaunchKernelGGL(k_dot_strided, dim3(p.gx, p.gy, p.gz), dim3(RM_RBLOCK), 0, c->stream, a, b, (rm_u64)pre,
                           (rm_u64)red, (rm_u64)p.nsplit, p.tx, pv, pn);
    RMHIP_HIP_CHECK(hipGetLastError());
    const unsigned fb = (unsigned)ceil_div_u64(p.nslices, RM_RBLOCK / 64);
    hipLaunchKernelGGL((k_reduce_final<RM_RSUM>), dim3(fb), dim3(RM_RBLOCK), 0, c->stream, pv, pn, (rm_u64)p.nslices,
                       (rm_u64)p.nsplit, (rm_u64)red, 0, 0, 1.0, out);
    RMHIP_HIP_CHECK(hipGetLastError());
    c->tel.kernel_launches += 2;
    return RMHIP_OK;
}

int launch_reduce_all(Context* c, int op, int nan_mode, const double* x, size_t n, double* out) {
    return launch_reduce_mid(c, op, nan_mode, x, 1, n, 1, out);
}

}  // namespace rmhip
