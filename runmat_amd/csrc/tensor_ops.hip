// tensor_ops.hip -- the shape / indexing hooks the hot-path builtins call around the arithmetic kernels
//   repmat            crates/runmat-accelerate-api/src/lib.rs:2689-2695  (semantics simple_provider.rs:2174-2240, 6681-6697)
//   permute           lib.rs:2579-2585   (simple_provider.rs:1645-1740)
//   zeros_like / ones_like / fill_like   lib.rs:1497, 1547, 1524-1545
//   read_scalar       lib.rs:1463        (simple_provider.rs:3415-3429)
//   gather_linear / scatter_linear       lib.rs:1423-1445   (simple_provider.rs:2609-2720)
//   linspace          lib.rs:1887        (simple_provider.rs:3488-3513)
// All of them are pure data movement: HBM-bound, one read and one write per output element.  The copies share one kernel
// family (IndexMap, common.h): the output is walked contiguously - dim 0 along the threads, the outer coordinates
// decoded once per block with scalar arithmetic - and the source index is rebuilt from per-dimension maps.  A permutation
// that moves source dim 0 away from output dim 0 goes through a 64 x 64 LDS tile so that both sides stay coalesced.
#include <algorithm>
#include <cstring>

#include "common.h"

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c);                                             \
    NarrowScope _ns(c)

namespace rmhip {

namespace {
constexpr int kBlock = 256;

struct MapParams {
    unsigned long long d0, nchunks;
    int rank;
    int id0;  // dim 0 is the identity map (off 0, mod >= extent, not reversed): no modulo per element
    unsigned long long shape[8], stride[8], off[8], mod[8];
    unsigned char rev[8];
};

__device__ __forceinline__ unsigned long long map_coord(const MapParams& p, int d, unsigned long long cd) {
    if (p.rev[d]) return p.off[d] - cd;
    unsigned long long t = p.off[d] + cd;
    if (t >= p.mod[d]) t = (t | p.mod[d]) >> 32 ? t % p.mod[d] : (unsigned long long)((unsigned)t % (unsigned)p.mod[d]);
    return t;
}

// One block: kBlock * E consecutive elements of dim 0 at one outer coordinate (E = 4, or less when dim 0 is short: a 512-element
// dim 0 on the four-element form leaves half of every block idle).
template <class T, int E>
__global__ void __launch_bounds__(kBlock) k_index_copy(const T* __restrict__ src, T* __restrict__ dst, MapParams p) {
    const unsigned long long blk = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;
    const unsigned long long chunk = blk % p.nchunks;
    const unsigned long long outer = blk / p.nchunks;
    unsigned long long base = 0, rem = outer;
    for (int d = 1; d < p.rank; ++d) {
        const unsigned long long cd = rem % p.shape[d];
        rem /= p.shape[d];
        base += map_coord(p, d, cd) * p.stride[d];
    }
    if (rem != 0) return;
    const unsigned long long obase = outer * p.d0;
    const unsigned long long i0 = chunk * (unsigned long long)(kBlock * E) + threadIdx.x;
    T v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long i = i0 + (unsigned long long)e * kBlock;
        if (i < p.d0) v[e] = src[base + (p.id0 ? i : map_coord(p, 0, i)) * p.stride[0]];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long i = i0 + (unsigned long long)e * kBlock;
        if (i < p.d0) dst[obase + i] = v[e];
    }
}

// Short dim 0 with many outer coordinates (repmat of a 1 x N row to 8 x N): flat threads over the output, 32-bit decode.
template <class T>
__global__ void __launch_bounds__(kBlock) k_index_copy_flat(const T* __restrict__ src, T* __restrict__ dst, MapParams p, unsigned n) {
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned idx = blockIdx.x * kBlock + threadIdx.x; idx < n; idx += stride) {
        unsigned rem = idx;
        unsigned long long s = 0;
        for (int d = 0; d < p.rank; ++d) {
            const unsigned sd = (unsigned)p.shape[d];
            const unsigned q = rem / sd, cd = rem - q * sd;
            rem = q;
            s += map_coord(p, d, cd) * p.stride[d];
        }
        dst[idx] = src[s];
    }
}

template <class T>
int index_copy(Context* c, const T* src, T* dst, size_t n, const IndexMap& m) {
    if (n == 0) return RMHIP_OK;
    if (m.rank < 1 || m.rank > 8) return fail(RMHIP_ERR_UNSUPPORTED, "index copy: rank %d", m.rank);
    MapParams p;
    p.rank = m.rank;
    p.d0 = m.shape[0];
    const int elems = p.d0 > 2ull * kBlock ? 4 : (p.d0 > (unsigned long long)kBlock ? 2 : 1);
    p.nchunks = (p.d0 + (unsigned long long)kBlock * elems - 1) / ((unsigned long long)kBlock * elems);
    unsigned long long outer = 1;
    for (int i = 0; i < 8; ++i) {
        const bool on = i < m.rank;
        p.shape[i] = on ? m.shape[i] : 1;
        p.stride[i] = on ? m.stride[i] : 0;
        p.off[i] = on ? m.off[i] : 0;
        p.mod[i] = on ? (m.mod[i] ? m.mod[i] : 1) : 1;
        p.rev[i] = on ? m.rev[i] : 0;
        if (i >= 1 && on) outer *= m.shape[i];
    }
    p.id0 = !p.rev[0] && p.off[0] == 0 && p.mod[0] >= p.shape[0];
    if (p.d0 < 128 && outer >= 64 && n < 0x80000000ULL) {
        const unsigned long long want = (n + kBlock - 1) / kBlock, cap = (unsigned long long)c->num_cus * 16;
        hipLaunchKernelGGL((k_index_copy_flat<T>), dim3((unsigned)std::min(want, cap)), dim3(kBlock), 0, c->stream, src, dst, p, (unsigned)n);
    } else {
        const unsigned long long blocks = p.nchunks * outer;
        const unsigned long long gx = std::min<unsigned long long>(blocks, 1048576ULL);
        const unsigned long long gy = (blocks + gx - 1) / gx;
        if (gy > 65535ULL) return fail(RMHIP_ERR_UNSUPPORTED, "index copy: grid too large");
        if (elems == 4) hipLaunchKernelGGL((k_index_copy<T, 4>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, c->stream, src, dst, p);
        else if (elems == 2) hipLaunchKernelGGL((k_index_copy<T, 2>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, c->stream, src, dst, p);
        else hipLaunchKernelGGL((k_index_copy<T, 1>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, c->stream, src, dst, p);
    }
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    return RMHIP_OK;
}

// ---- permutation that moves source dim 0 to output dim j > 0 -----------------------------------------------------------
// Tile = 64 (output dim 0; source stride s0) x 64 (output dim j = source dim 0, stride 1).  Loads run along the source's
// contiguous dimension, stores along the output's; the transposition happens in LDS (row padding: no bank conflicts).
struct PermParams {
    int rank, j;
    unsigned long long shape[8];    // output extents
    unsigned long long sstride[8];  // source stride of every output dim (sstride[j] == 1)
    unsigned long long ostride[8];  // output strides
    unsigned long long t0, tj;      // tiles along dim 0 / dim j
};
template <class T>
__global__ void __launch_bounds__(256) k_permute_tiled(const T* __restrict__ src, T* __restrict__ dst, PermParams p) {
    __shared__ T tile[64][65];
    unsigned long long blk = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;
    // consecutive workgroups advance along the SOURCE's contiguous dimension (output dim j): their loads continue each other's 512-byte
    // row pieces (the other order - along the output's contiguous dimension - measured 275 us against 225 for 8192^2)
    const unsigned long long bj = blk % p.tj;
    blk /= p.tj;
    const unsigned long long b0 = blk % p.t0;
    unsigned long long rem = blk / p.t0;
    unsigned long long sbase = 0, obase = 0;
    for (int d = 1; d < p.rank; ++d) {
        if (d == p.j) continue;
        const unsigned long long cd = rem % p.shape[d];
        rem /= p.shape[d];
        sbase += cd * p.sstride[d];
        obase += cd * p.ostride[d];
    }
    if (rem != 0) return;
    const unsigned long long i0 = b0 * 64, j0 = bj * 64;  // tile origin: output dim 0, output dim j
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    // load: tx runs along source dim 0 (= output dim j).  All 16 loads are issued before the first one is used (indices clamped into
    // the tensor instead of a branch per load - a guarded loop waits for every load before it issues the next: 268 us against 225 for
    // 8192^2), the stores are guarded.
    const unsigned long long jj = j0 + tx;
    const unsigned long long jc = jj < p.shape[p.j] ? jj : p.shape[p.j] - 1;
    T v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const unsigned long long ii = i0 + ty + 4 * q;
        const unsigned long long ic = ii < p.shape[0] ? ii : p.shape[0] - 1;
        v[q] = src[sbase + jc + ic * p.sstride[0]];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[ty + 4 * q][tx] = v[q];
    __syncthreads();
    // store: tx runs along output dim 0
    const unsigned long long io = i0 + tx;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const unsigned long long jo = j0 + ty + 4 * q;
        if (io < p.shape[0] && jo < p.shape[p.j]) dst[obase + io + jo * p.ostride[p.j]] = tile[tx][ty + 4 * q];
    }
}

template <class T>
__global__ void __launch_bounds__(kBlock) k_gather(const T* __restrict__ src, const unsigned* __restrict__ idx, T* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}
// large index sets: the bounds check rides along (smallest offending position in *bad; such an element reads nothing)
template <class T>
__global__ void __launch_bounds__(kBlock) k_gather_checked(const T* __restrict__ src, const unsigned* __restrict__ idx, T* __restrict__ dst, size_t n,
                                                           size_t numel, unsigned long long* __restrict__ bad) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const unsigned t = idx[i];
        if (t < numel) dst[i] = src[t];
        else atomicMin(bad, (unsigned long long)i);
    }
}
// Duplicates: the reference writes sequentially, so the LAST occurrence of an index wins (simple_provider.rs:2706-2717).
// `winner[k]` (host-computed) is 1 when position k is the last one carrying its index: the stores never race.
template <class T>
__global__ void __launch_bounds__(kBlock) k_scatter(T* __restrict__ target, const unsigned* __restrict__ idx, const unsigned char* __restrict__ winner,
                                                    const T* __restrict__ values, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        if (!winner || winner[i]) target[idx[i]] = values[i];
}
// Large index sets: bounds and duplicates resolved on the device.  Pass 1 leaves in owner[idx] the largest position + 1 that carries
// idx (the reference's sequential loop lets the LAST occurrence win, simple_provider.rs:2698-2711) and in *bad the smallest position
// whose index is out of bounds; pass 2 stores only from the owning position, so no two stores race.
__global__ void __launch_bounds__(kBlock) k_scatter_mark(const unsigned* __restrict__ idx, size_t n, size_t numel, unsigned* __restrict__ owner,
                                                         unsigned long long* __restrict__ bad) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const unsigned t = idx[i];
        if (t >= numel) atomicMin(bad, (unsigned long long)i);
        else atomicMax(owner + t, (unsigned)(i + 1));
    }
}
template <class T>
__global__ void __launch_bounds__(kBlock) k_scatter_owned(T* __restrict__ target, const unsigned* __restrict__ idx, const unsigned* __restrict__ owner,
                                                          const T* __restrict__ values, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const unsigned t = idx[i];
        if (owner[t] == (unsigned)(i + 1)) target[t] = values[i];
    }
}
// simple_provider.rs:3494-3503: start + idx * step with step = (stop - start) / (count - 1), the last element set to stop
template <class T>
__global__ void __launch_bounds__(kBlock) k_linspace(T* __restrict__ out, size_t n, double start, double step, double stop) {
    // write-only: one contiguous chunk of 1024 elements per block (ew_kernels.hip k_fill, scripts/micro/write_patterns.hip)
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i = b + e * 256;
        if (i < n) {
            const double prod = (double)i * step;  // separate multiply and add (-ffp-contract=off), as the CPU loop rounds
            out[i] = (T)(i + 1 == n ? stop : start + prod);
        }
    }
}

// identity_data (simple_provider.rs:2293-2336): 1 where row == col inside a page, written as one chunk of 1024 elements per block
__global__ void __launch_bounds__(kBlock) k_eye(double* __restrict__ out, size_t n, size_t rows, size_t cols) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i = b + e * 256;
        if (i < n) {
            const size_t r = i % rows, c = (i / rows) % cols;
            out[i] = r == c ? 1.0 : 0.0;
        }
    }
}
// tril_data / triu_data (simple_provider.rs:1974-2081)
template <class T>
__global__ void __launch_bounds__(kBlock) k_tri(const T* __restrict__ in, T* __restrict__ out, size_t n, size_t rows, size_t cols, int upper, long long offset) {
    const size_t b = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i = b + e * 256;
        if (i < n) {
            const long long r = (long long)(i % rows), c = (long long)((i / rows) % cols);
            const bool zero = upper ? (c - r < offset) : (r - c < -offset);
            out[i] = zero ? (T)0 : in[i];
        }
    }
}

unsigned flat_grid(const Context* c, size_t n) {
    const size_t want = (n + kBlock - 1) / kBlock, cap = (size_t)c->num_cus * 16;
    return (unsigned)std::max<size_t>(1, std::min(want, cap));
}
}  // namespace

int launch_index_copy(Context* c, const double* src, double* dst, size_t n, const IndexMap& m) { return index_copy(c, src, dst, n, m); }
int launch_index_copy_f32(Context* c, const float* src, float* dst, size_t n, const IndexMap& m) { return index_copy(c, src, dst, n, m); }

int materialize_repmat(Context* c, const Buffer& v, void* dst) {
    // collapse runs of dimensions the view does not tile (rep == 1 ... rep == 1) into one: rank 8 covers 4 tiled N-d axes and more
    std::vector<uint64_t> shape, stride, mod;
    uint64_t s = 1;
    for (size_t d = 0; d < v.shape.size(); ++d) {
        const uint64_t e = v.shape[d], b = v.rep_base[d];
        const bool plain = e == b;
        if (e == 1) {
            s *= b;
            continue;
        }
        if (plain && !shape.empty() && mod.back() == shape.back() && stride.back() * shape.back() == s) {
            shape.back() *= e;  // contiguous with the previous untiled dimension
            mod.back() = shape.back();
        } else {
            shape.push_back(e);
            stride.push_back(b == 1 ? 0 : s);
            mod.push_back(b == 1 ? e : b);  // a replicated single element: identity map, stride 0
        }
        s *= b;
    }
    if (shape.empty()) {
        shape.push_back(1);
        stride.push_back(0);
        mod.push_back(1);
    }
    if (shape.size() > 8) return fail(RMHIP_ERR_UNSUPPORTED, "repmat: more than 8 tiled dimensions");
    IndexMap m;
    m.rank = (int)shape.size();
    for (int d = 0; d < m.rank; ++d) {
        m.shape[d] = shape[d];
        m.stride[d] = stride[d];
        m.mod[d] = mod[d];
        m.off[d] = 0;
        m.rev[d] = 0;
    }
    return v.dtype == DT_F32 ? launch_index_copy_f32(c, v.data_f32(), (float*)dst, v.numel, m)
                             : launch_index_copy(c, v.data(), (double*)dst, v.numel, m);
}

}  // namespace rmhip

namespace {

// fetch an operand in its own storage type, views materialised (the copies below move bytes, they do not compute)
int get_settled(Context* c, rmhip_buf id, Buffer* out) {
    RMHIP_TRY(c->get_raw(id, out));
    if (out->lazy()) {
        RMHIP_TRY(c->settle_view(id));
        RMHIP_TRY(c->get_raw(id, out));
    }
    return RMHIP_OK;
}

// a new buffer of the operand's storage type (f32 results of a precision-32 context are not narrowed again)
int new_like(Context* c, const Buffer& like, const size_t* shape, size_t rank, rmhip_buf* id, Buffer* out) {
    return like.dtype == DT_F32 ? c->new_buffer_f32(shape, rank, id, out) : c->new_buffer(shape, rank, id, out);
}

// device copy of host u32 indices; the caller keeps `hold` alive until the kernel is enqueued (stream ordered pool)
int upload_indices(Context* c, const uint32_t* idx, size_t n, std::shared_ptr<Allocation>* hold, size_t extra_bytes = 0) {
    const size_t bytes = n * sizeof(uint32_t) + extra_bytes;
    RMHIP_TRY(c->alloc_device((bytes + 7) / 8 ? (bytes + 7) / 8 : 1, hold));
    RMHIP_HIP_CHECK(hipMemcpyAsync((*hold)->ptr, idx, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    return RMHIP_OK;
}

}  // namespace

extern "C" {

int rmhip_repmat(rmhip_ctx* ctx, rmhip_buf a, const size_t* reps, size_t n_reps, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || !reps) return fail(RMHIP_ERR_INVALID, "repmat: null argument");
    if (n_reps == 0) return fail(RMHIP_ERR_INVALID, "repmat: replication factors must be specified");  // simple_provider.rs:2175-2178
    Buffer ab;
    RMHIP_TRY(c->get_raw(a, &ab));
    if (ab.tview) {  // a view of a transpose view: the base first, in its own storage type
        RMHIP_TRY(c->settle_view(a));
        RMHIP_TRY(c->get_raw(a, &ab));
    }
    // simple_provider.rs:2179-2203: rank, base shape padded with 1s, one factor = every dimension
    const size_t orig_rank = ab.shape.empty() ? 1 : ab.shape.size();
    const size_t rank = n_reps == 1 ? std::max<size_t>(orig_rank, 2) : std::max(orig_rank, n_reps);
    std::vector<size_t> base(rank, 1), factors(rank, 1), shape(rank);
    for (size_t i = 0; i < ab.shape.size(); ++i) base[i] = ab.shape[i];
    if (n_reps == 1) std::fill(factors.begin(), factors.end(), reps[0]);
    else
        for (size_t i = 0; i < n_reps; ++i) factors[i] = reps[i];
    size_t total = 1;
    for (size_t i = 0; i < rank; ++i) {
        if (factors[i] && base[i] > SIZE_MAX / factors[i]) return fail(RMHIP_ERR_INVALID, "repmat: requested output exceeds maximum size");
        shape[i] = base[i] * factors[i];
        if (shape[i] && total > SIZE_MAX / shape[i]) return fail(RMHIP_ERR_INVALID, "repmat: requested output exceeds maximum size");
        total *= shape[i];
    }
    if (total == 0) {  // simple_provider.rs:2213-2216: an empty tensor of the tiled shape
        Buffer ob;
        return new_like(c, ab, shape.data(), rank, out, &ob);
    }
    Buffer r;
    r.alloc = ab.alloc;
    r.shape = shape;
    r.numel = total;
    r.dtype = ab.dtype;
    if (!ab.rep_base.empty()) {
        // a view of a view: coordinates reduce modulo the first base, so the combined view keeps that base as long as this
        // call's base (the first view's tiled shape) is a whole number of first-base periods in every dimension - it is, by construction
        r.rep_base.assign(rank, 1);
        for (size_t i = 0; i < ab.rep_base.size(); ++i) r.rep_base[i] = ab.rep_base[i];
    } else {
        r.rep_base = base;
    }
    bool any = false;
    for (size_t i = 0; i < rank; ++i) any = any || r.rep_base[i] != shape[i];
    if (!any) r.rep_base.clear();  // every factor 1: a plain alias (same bytes, the shape padded to `rank`)
    RMHIP_TRY(c->register_buffer(std::move(r), out));
    if (const char* e = std::getenv("RMHIP_EAGER_REPMAT"))
        if (e[0] == '1') RMHIP_TRY(c->settle_view(*out));
    return RMHIP_OK;
}

int rmhip_permute(rmhip_ctx* ctx, rmhip_buf a, const size_t* order, size_t n_order, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (n_order && !order)) return fail(RMHIP_ERR_INVALID, "permute: null argument");
    if (n_order == 0) return fail(RMHIP_ERR_INVALID, "permute: order must not be empty");  // simple_provider.rs:1651
    Buffer ab, ob;
    RMHIP_TRY(get_settled(c, a, &ab));
    const size_t rank = n_order;
    if (ab.shape.size() > rank) return fail(RMHIP_ERR_INVALID, "permute: order length must be at least the number of dimensions");
    std::vector<char> seen(rank, 0);
    for (size_t d = 0; d < rank; ++d) {
        if (order[d] >= rank) return fail(RMHIP_ERR_INVALID, "permute: invalid dimension index %zu", order[d] + 1);
        if (seen[order[d]]) return fail(RMHIP_ERR_INVALID, "permute: duplicate dimension index %zu encountered", order[d] + 1);
        seen[order[d]] = 1;
    }
    std::vector<size_t> src_shape = ab.shape;
    src_shape.resize(rank, 1);
    std::vector<uint64_t> src_stride(rank);
    uint64_t s = 1;
    for (size_t d = 0; d < rank; ++d) {
        src_stride[d] = s;
        s *= src_shape[d];
    }
    std::vector<size_t> dst_shape(rank);
    for (size_t d = 0; d < rank; ++d) dst_shape[d] = src_shape[order[d]];
    RMHIP_TRY(new_like(c, ab, dst_shape.data(), rank, out, &ob));
    if (ab.numel == 0) return RMHIP_OK;
    // drop extent-1 output dims, merge output dims whose source dims are adjacent too (order[d+1] == order[d] + 1 after dropping)
    std::vector<uint64_t> shape, sstr;
    for (size_t d = 0; d < rank; ++d) {
        if (dst_shape[d] == 1) continue;
        const uint64_t st = src_stride[order[d]];
        if (!shape.empty() && sstr.back() * shape.back() == st) shape.back() *= dst_shape[d];
        else {
            shape.push_back(dst_shape[d]);
            sstr.push_back(st);
        }
    }
    if (shape.empty()) {
        shape.push_back(1);
        sstr.push_back(1);
    }
    int rc;
    if (shape.size() > 8) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_UNSUPPORTED, "permute: more than 8 dimensions after merging");
    }
    int j = -1;  // output dim that is the source's contiguous one
    for (size_t d = 0; d < shape.size(); ++d)
        if (sstr[d] == 1) j = (int)d;
    if (j > 0 && shape[0] >= 8 && shape[j] >= 8) {
        PermParams p;
        p.rank = (int)shape.size();
        p.j = j;
        uint64_t os = 1;
        unsigned long long others = 1;
        for (int d = 0; d < 8; ++d) {
            const bool on = d < p.rank;
            p.shape[d] = on ? shape[d] : 1;
            p.sstride[d] = on ? sstr[d] : 0;
            p.ostride[d] = os;
            if (on) os *= shape[d];
            if (on && d != 0 && d != j) others *= shape[d];
        }
        p.t0 = (shape[0] + 63) / 64;
        p.tj = (shape[j] + 63) / 64;
        const unsigned long long blocks = p.t0 * p.tj * others;
        const unsigned long long gx = std::min<unsigned long long>(blocks, 1048576ULL), gy = (blocks + gx - 1) / gx;
        if (gy > 65535ULL) {
            rmhip_free(ctx, *out);
            return fail(RMHIP_ERR_UNSUPPORTED, "permute: grid too large");
        }
        if (ab.dtype == DT_F32) hipLaunchKernelGGL((k_permute_tiled<float>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, ab.data_f32(), ob.data_f32(), p);
        else hipLaunchKernelGGL((k_permute_tiled<double>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, ab.data(), ob.data(), p);
        c->tel.kernel_launches++;
        hipError_t e = hipGetLastError();
        rc = e == hipSuccess ? RMHIP_OK : fail(RMHIP_ERR_HIP, "permute launch: %s", hipGetErrorString(e));
    } else {
        IndexMap m;
        m.rank = (int)shape.size();
        for (int d = 0; d < m.rank; ++d) {
            m.shape[d] = shape[d];
            m.stride[d] = sstr[d];
            m.off[d] = 0;
            m.mod[d] = shape[d];
            m.rev[d] = 0;
        }
        rc = ab.dtype == DT_F32 ? launch_index_copy_f32(c, ab.data_f32(), ob.data_f32(), ab.numel, m)
                                : launch_index_copy(c, ab.data(), ob.data(), ab.numel, m);
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_fill_like(rmhip_ctx* ctx, rmhip_buf prototype, double value, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer pb;
    RMHIP_TRY(c->get_raw(prototype, &pb));  // shape only
    return rmhip_fill(ctx, value, pb.shape.data(), pb.shape.size(), out);
}

int rmhip_read_scalar(rmhip_ctx* ctx, rmhip_buf a, size_t linear_index, double* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab;
    RMHIP_TRY(c->get_raw(a, &ab));
    if (linear_index >= ab.numel)  // simple_provider.rs:3420-3426
        return fail(RMHIP_ERR_INVALID, "read_scalar: index %zu out of bounds (len %zu)", linear_index + 1, ab.numel);
    size_t at = linear_index;
    if (ab.tview) {  // logical [R, C] over storage C x R
        const size_t R = ab.shape[0], C = ab.shape[1];
        const size_t i = linear_index % R, j = linear_index / R;
        at = j + i * C;
    } else if (!ab.rep_base.empty()) {
        size_t rem = linear_index, s = 1;
        at = 0;
        for (size_t d = 0; d < ab.shape.size(); ++d) {
            const size_t cd = rem % ab.shape[d];
            rem /= ab.shape[d];
            at += (cd % ab.rep_base[d]) * s;
            s *= ab.rep_base[d];
        }
    }
    if (ab.dtype == DT_F32) {
        float v = 0.f;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&v, ab.data_f32() + at, sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        *out = (double)v;
    } else {
        RMHIP_HIP_CHECK(hipMemcpyAsync(out, ab.data() + at, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    c->tel.download_bytes += sizeof(double);
    return RMHIP_OK;
}

int rmhip_gather_linear(rmhip_ctx* ctx, rmhip_buf source, const uint32_t* indices, size_t n_indices, const size_t* out_shape, size_t rank,
                        rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (n_indices && !indices) || (rank && !out_shape)) return fail(RMHIP_ERR_INVALID, "gather_linear: null argument");
    if (shape_numel(out_shape, rank) != n_indices)
        return fail(RMHIP_ERR_SHAPE, "gather_linear: output shape holds %zu elements, %zu indices given", shape_numel(out_shape, rank), n_indices);
    Buffer sb, ob;
    RMHIP_TRY(get_settled(c, source, &sb));
    static const size_t device_min = std::getenv("RMHIP_SCATTER_DEVICE_MIN") ? (size_t)std::atol(std::getenv("RMHIP_SCATTER_DEVICE_MIN")) : 4096;
    const bool on_device = device_min && n_indices >= device_min;  // the bounds check inside the gather kernel instead of a host loop
    if (!on_device)
        for (size_t k = 0; k < n_indices; ++k)
            if (indices[k] >= sb.numel)  // simple_provider.rs:2636-2643
                return fail(RMHIP_ERR_INVALID, "gather_linear: index %u (position %zu) out of bounds for buffer %llu (logical_len=%zu)", indices[k], k,
                            (unsigned long long)source, sb.numel);
    RMHIP_TRY(new_like(c, sb, out_shape, rank, out, &ob));
    if (n_indices == 0) return RMHIP_OK;
    std::shared_ptr<Allocation> didx;
    int rc = upload_indices(c, indices, n_indices, &didx, sizeof(unsigned long long) + 8);
    if (rc == RMHIP_OK && on_device) {
        const unsigned* di = reinterpret_cast<const unsigned*>(didx->ptr);
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(didx->ptr) + ((n_indices * sizeof(uint32_t) + 7) & ~(size_t)7));
        unsigned long long first_bad = ~0ULL;
        hipError_t e = hipMemsetAsync(bad, 0xff, sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) {
            if (sb.dtype == DT_F32) hipLaunchKernelGGL((k_gather_checked<float>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, sb.data_f32(), di, ob.data_f32(), n_indices, sb.numel, bad);
            else hipLaunchKernelGGL((k_gather_checked<double>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, sb.data(), di, ob.data(), n_indices, sb.numel, bad);
            c->tel.kernel_launches++;
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&first_bad, bad, sizeof(first_bad), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "gather_linear: %s", hipGetErrorString(e));
        else if (first_bad != ~0ULL)
            rc = fail(RMHIP_ERR_INVALID, "gather_linear: index %u (position %zu) out of bounds for buffer %llu (logical_len=%zu)", indices[first_bad],
                      (size_t)first_bad, (unsigned long long)source, sb.numel);
    } else if (rc == RMHIP_OK) {
        const unsigned* di = reinterpret_cast<const unsigned*>(didx->ptr);
        if (sb.dtype == DT_F32) hipLaunchKernelGGL((k_gather<float>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, sb.data_f32(), di, ob.data_f32(), n_indices);
        else hipLaunchKernelGGL((k_gather<double>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, sb.data(), di, ob.data(), n_indices);
        c->tel.kernel_launches++;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller's index slice (pageable host memory) may go away on return
        if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "gather_linear: %s", hipGetErrorString(e));
    }
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_scatter_linear(rmhip_ctx* ctx, rmhip_buf target, const uint32_t* indices, size_t n_indices, rmhip_buf values) {
    CTX_OR_FAIL(ctx);
    if (n_indices && !indices) return fail(RMHIP_ERR_INVALID, "scatter_linear: null indices");
    Buffer tb, vb;
    RMHIP_TRY(get_settled(c, target, &tb));
    RMHIP_TRY(c->detach_views_of(target));  // written in place: a repmat / transpose view of the target keeps the values it was made from
    RMHIP_TRY(get_settled(c, values, &vb));
    if (tb.dtype != vb.dtype)  // simple_provider.rs:2663-2669 (storage mismatch)
        return fail(RMHIP_ERR_UNSUPPORTED, "scatter_linear: storage mismatch target=%s values=%s", tb.dtype == DT_F32 ? "f32" : "f64", vb.dtype == DT_F32 ? "f32" : "f64");
    if (vb.numel != n_indices)
        return fail(RMHIP_ERR_SHAPE, "scatter_linear: values raw length %zu does not match index count %zu for lane factor 1", vb.numel, n_indices);
    if (n_indices == 0) return RMHIP_OK;
    // From kScatterDeviceMin indices on (and fewer than 2^32 - 1 of them) the host does nothing per index: an O(n) bounds loop, a sort
    // and a hash map per call were the cost of a large `A(idx) = v` (1e7 indices: seconds on the host against two short kernels).
    // The owner table costs 4 bytes per target element, cleared per call (RMHIP_SCATTER_DEVICE_MIN, 0 = never).
    static const size_t device_min = std::getenv("RMHIP_SCATTER_DEVICE_MIN") ? (size_t)std::atol(std::getenv("RMHIP_SCATTER_DEVICE_MIN")) : 4096;
    if (device_min && n_indices >= device_min && n_indices < 0xffffffffULL && tb.numel <= 64 * n_indices) {
        std::shared_ptr<Allocation> didx, owner;
        RMHIP_TRY(upload_indices(c, indices, n_indices, &didx, sizeof(unsigned long long) + 8));
        RMHIP_TRY(c->alloc_device((tb.numel + 1) / 2 + 1, &owner));
        const unsigned* di = reinterpret_cast<const unsigned*>(didx->ptr);
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(didx->ptr) + ((n_indices * sizeof(uint32_t) + 7) & ~(size_t)7));
        unsigned* own = reinterpret_cast<unsigned*>(owner->ptr);
        RMHIP_HIP_CHECK(hipMemsetAsync(own, 0, tb.numel * sizeof(unsigned), c->stream));
        RMHIP_HIP_CHECK(hipMemsetAsync(bad, 0xff, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(k_scatter_mark, dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, di, n_indices, tb.numel, own, bad);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        unsigned long long first_bad = 0;
        RMHIP_HIP_CHECK(hipMemcpyAsync(&first_bad, bad, sizeof(first_bad), hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // (also: the caller's index array has been read)
        if (first_bad != ~0ULL)  // nothing was stored (the reference's loop stores the positions before the offending one, then fails: simple_provider.rs:2698-2706; both paths here validate first)
            return fail(RMHIP_ERR_INVALID, "scatter_linear: index %u (position %zu) out of bounds for target (logical_len=%zu)", indices[first_bad],
                        (size_t)first_bad, tb.numel);
        if (tb.dtype == DT_F32) hipLaunchKernelGGL((k_scatter_owned<float>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, tb.data_f32(), di, own, vb.data_f32(), n_indices);
        else hipLaunchKernelGGL((k_scatter_owned<double>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, tb.data(), di, own, vb.data(), n_indices);
        c->tel.kernel_launches++;
        RMHIP_HIP_CHECK(hipGetLastError());
        return RMHIP_OK;  // (didx / owner return to the pool in stream order)
    }
    for (size_t k = 0; k < n_indices; ++k)
        if (indices[k] >= tb.numel)
            return fail(RMHIP_ERR_INVALID, "scatter_linear: index %u (position %zu) out of bounds for target (logical_len=%zu)", indices[k], k, tb.numel);
    // last occurrence of an index wins, as in the reference's sequential loop: mark winners on the host (only when duplicates exist)
    std::vector<unsigned char> winner;
    {
        std::vector<uint32_t> sorted(indices, indices + n_indices);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
            winner.assign(n_indices, 0);
            std::unordered_map<uint32_t, size_t> last;
            last.reserve(n_indices);
            for (size_t k = 0; k < n_indices; ++k) last[indices[k]] = k;
            for (const auto& kv : last) winner[kv.second] = 1;
        }
    }
    std::shared_ptr<Allocation> didx;
    RMHIP_TRY(upload_indices(c, indices, n_indices, &didx, winner.size()));
    const unsigned* di = reinterpret_cast<const unsigned*>(didx->ptr);
    const unsigned char* dw = nullptr;
    if (!winner.empty()) {
        dw = reinterpret_cast<const unsigned char*>(didx->ptr) + n_indices * sizeof(uint32_t);
        RMHIP_HIP_CHECK(hipMemcpyAsync((void*)dw, winner.data(), winner.size(), hipMemcpyHostToDevice, c->stream));
    }
    if (tb.dtype == DT_F32) hipLaunchKernelGGL((k_scatter<float>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, tb.data_f32(), di, dw, vb.data_f32(), n_indices);
    else hipLaunchKernelGGL((k_scatter<double>), dim3(flat_grid(c, n_indices)), dim3(kBlock), 0, c->stream, tb.data(), di, dw, vb.data(), n_indices);
    c->tel.kernel_launches++;
    RMHIP_HIP_CHECK(hipGetLastError());
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));  // host index / winner arrays go out of scope
    return RMHIP_OK;
}

int rmhip_linspace(rmhip_ctx* ctx, double start, double stop, size_t count, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    const size_t shape[2] = {1, count};
    Buffer ob;
    RMHIP_TRY(c->new_buffer(shape, 2, out, &ob));  // f64, narrowed on return by a precision-32 context
    if (count == 0) return RMHIP_OK;
    const double step = count > 1 ? (stop - start) / (double)(count - 1) : 0.0;
    if ((count + 1023) / 1024 > 0x7fffffffULL) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_UNSUPPORTED, "linspace: %zu elements exceed the launch limits", count);
    }
    hipLaunchKernelGGL((k_linspace<double>), dim3((unsigned)((count + 1023) / 1024)), dim3(kBlock), 0, c->stream, ob.data(), count, start, step, stop);
    c->tel.kernel_launches++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_HIP, "linspace launch: %s", hipGetErrorString(e));
    }
    return RMHIP_OK;
}

int rmhip_eye(rmhip_ctx* ctx, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (rank && !shape)) return fail(RMHIP_ERR_INVALID, "eye: null argument");
    std::vector<size_t> s(shape, shape + rank);
    if (s.empty()) s = {1, 1};
    else if (s.size() == 1) s = {s[0], s[0]};  // normalize_shape, simple_provider.rs:779-788
    Buffer ob;
    RMHIP_TRY(c->new_buffer(s.data(), s.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    if ((ob.numel + 1023) / 1024 > 0x7fffffffULL) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_UNSUPPORTED, "eye: %zu elements exceed the launch limits", ob.numel);
    }
    hipLaunchKernelGGL(k_eye, dim3((unsigned)((ob.numel + 1023) / 1024)), dim3(kBlock), 0, c->stream, ob.data(), ob.numel, s[0], s[1]);
    c->tel.kernel_launches++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_HIP, "eye launch: %s", hipGetErrorString(e));
    }
    return RMHIP_OK;
}

namespace {
// out = in with per-dimension maps (flip: rev, circshift: off): one IndexMap copy over the dimensions that are not trivial
int mapped_copy(rmhip_ctx* ctx, Context* c, const Buffer& ab, const std::vector<size_t>& shape, const std::vector<uint64_t>& off,
                const std::vector<uint8_t>& rev, rmhip_buf* out) {
    Buffer ob;
    RMHIP_TRY(new_like(c, ab, ab.shape.data(), ab.shape.size(), out, &ob));  // the operand's own shape (flip / circshift keep it)
    if (ab.numel == 0) return RMHIP_OK;
    // merge runs of identity dimensions (contiguous in source and output alike), drop extent-1 dimensions
    IndexMap m;
    m.rank = 0;
    uint64_t stride = 1;
    bool prev_identity = false;
    for (size_t d = 0; d < shape.size(); ++d) {
        const uint64_t e = shape[d];
        const bool identity = !rev[d] && off[d] == 0;
        if (e != 1) {
            if (identity && prev_identity && m.rank > 0) {
                m.shape[m.rank - 1] *= e;
                m.mod[m.rank - 1] = m.shape[m.rank - 1];
            } else {
                if (m.rank == 8) {
                    rmhip_free(ctx, *out);
                    return fail(RMHIP_ERR_UNSUPPORTED, "more than 8 mapped dimensions");
                }
                m.shape[m.rank] = e;
                m.stride[m.rank] = stride;
                m.off[m.rank] = off[d];
                m.mod[m.rank] = e;
                m.rev[m.rank] = rev[d];
                ++m.rank;
            }
            prev_identity = identity;
        }
        stride *= e;
    }
    if (m.rank == 0) {
        m.rank = 1;
        m.shape[0] = 1;
        m.stride[0] = 1;
        m.off[0] = 0;
        m.mod[0] = 1;
        m.rev[0] = 0;
    }
    const int rc = ab.dtype == DT_F32 ? launch_index_copy_f32(c, ab.data_f32(), ob.data_f32(), ab.numel, m)
                                      : launch_index_copy(c, ab.data(), ob.data(), ab.numel, m);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}
}  // namespace

int rmhip_flip(rmhip_ctx* ctx, rmhip_buf a, const size_t* axes, size_t n_axes, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (n_axes && !axes)) return fail(RMHIP_ERR_INVALID, "flip: null argument");
    Buffer ab;
    RMHIP_TRY(get_settled(c, a, &ab));
    std::vector<size_t> shape = ab.shape;
    for (size_t k = 0; k < n_axes; ++k)
        if (axes[k] >= shape.size()) shape.resize(axes[k] + 1, 1);  // flip_data: axes beyond the rank are extent-1 dimensions
    std::vector<uint8_t> rev(shape.size(), 0);
    std::vector<uint64_t> off(shape.size(), 0);
    for (size_t k = 0; k < n_axes; ++k) rev[axes[k]] ^= 1;  // named twice: flipped back (simple_provider.rs:1757-1762)
    for (size_t d = 0; d < shape.size(); ++d) {
        if (shape[d] <= 1) rev[d] = 0;
        if (rev[d]) off[d] = shape[d] - 1;
    }
    return mapped_copy(ctx, c, ab, shape, off, rev, out);
}

int rmhip_circshift(rmhip_ctx* ctx, rmhip_buf a, const long long* shifts, size_t n_shifts, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (n_shifts && !shifts)) return fail(RMHIP_ERR_INVALID, "circshift: null argument");
    Buffer ab;
    RMHIP_TRY(get_settled(c, a, &ab));
    std::vector<size_t> shape = ab.shape;
    if (n_shifts > shape.size()) shape.resize(n_shifts, 1);  // simple_provider.rs:6439-6442
    std::vector<uint8_t> rev(shape.size(), 0);
    std::vector<uint64_t> off(shape.size(), 0);
    for (size_t d = 0; d < shape.size() && d < n_shifts; ++d) {
        const long long len = (long long)shape[d];
        if (len <= 1) continue;
        long long v = shifts[d] % len;  // circshift_data: normalised into [0, len)
        if (v < 0) v += len;
        off[d] = (uint64_t)((len - v) % len);  // src = (coord + len - shift) % len
    }
    return mapped_copy(ctx, c, ab, shape, off, rev, out);
}

int rmhip_tri(rmhip_ctx* ctx, rmhip_buf a, int upper, long long offset, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer ab, ob;
    RMHIP_TRY(get_settled(c, a, &ab));
    RMHIP_TRY(new_like(c, ab, ab.shape.data(), ab.shape.size(), out, &ob));
    if (ab.numel == 0) return RMHIP_OK;
    const size_t rows = ab.shape.empty() ? 1 : ab.shape[0], cols = ab.shape.size() > 1 ? ab.shape[1] : 1;
    if ((ab.numel + 1023) / 1024 > 0x7fffffffULL) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_UNSUPPORTED, "tril / triu: %zu elements exceed the launch limits", ab.numel);
    }
    const dim3 grid((unsigned)((ab.numel + 1023) / 1024));
    if (ab.dtype == DT_F32) hipLaunchKernelGGL((k_tri<float>), grid, dim3(kBlock), 0, c->stream, ab.data_f32(), ob.data_f32(), ab.numel, rows, cols, upper, offset);
    else hipLaunchKernelGGL((k_tri<double>), grid, dim3(kBlock), 0, c->stream, ab.data(), ob.data(), ab.numel, rows, cols, upper, offset);
    c->tel.kernel_launches++;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rmhip_free(ctx, *out);
        return fail(RMHIP_ERR_HIP, "tril / triu launch: %s", hipGetErrorString(e));
    }
    return RMHIP_OK;
}

int rmhip_cat(rmhip_ctx* ctx, size_t dim, const rmhip_buf* inputs, size_t n_inputs, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || !inputs) return fail(RMHIP_ERR_INVALID, "cat: null argument");
    if (n_inputs < 2) return fail(RMHIP_ERR_INVALID, "cat: at least two input arrays are required");  // tensor.rs:462-465
    if (dim < 1) return fail(RMHIP_ERR_INVALID, "cat: dimension must be >= 1");
    const size_t dz = dim - 1;
    std::vector<Buffer> in(n_inputs);
    size_t rank = dz + 1;
    for (size_t k = 0; k < n_inputs; ++k) {
        RMHIP_TRY(get_settled(c, inputs[k], &in[k]));
        if (in[k].dtype != in[0].dtype) return fail(RMHIP_ERR_UNSUPPORTED, "cat: input precision mismatch");
        rank = std::max(rank, in[k].shape.size());
    }
    std::vector<std::vector<size_t>> shapes(n_inputs);
    for (size_t k = 0; k < n_inputs; ++k) {
        shapes[k] = in[k].shape;
        shapes[k].resize(rank, 1);
    }
    for (size_t ax = 0; ax < rank; ++ax) {
        if (ax == dz) continue;
        for (size_t k = 1; k < n_inputs; ++k)
            if (shapes[k][ax] != shapes[0][ax])
                return fail(RMHIP_ERR_SHAPE, "cat: dimension %zu mismatch between input 1 (size %zu) and input %zu (size %zu)", ax + 1, shapes[0][ax], k + 1,
                            shapes[k][ax]);
    }
    std::vector<size_t> oshape = shapes[0];
    size_t cat_extent = 0;
    for (size_t k = 0; k < n_inputs; ++k) cat_extent += shapes[k][dz];
    oshape[dz] = cat_extent;
    size_t inner = 1, outer = 1;
    for (size_t ax = 0; ax < dz; ++ax) inner *= oshape[ax];
    for (size_t ax = dz + 1; ax < rank; ++ax) outer *= oshape[ax];
    std::vector<size_t> nshape = oshape;  // normalize_concat_shape (backend_shared.rs:330-339)
    const size_t min_len = std::min(std::max<size_t>(dz + 1, 2), nshape.size());
    while (nshape.size() > min_len && nshape.back() == 1) nshape.pop_back();
    if (nshape.size() == 1) nshape.push_back(1);
    Buffer ob;
    RMHIP_TRY(new_like(c, in[0], nshape.data(), nshape.size(), out, &ob));
    if (ob.numel == 0) return RMHIP_OK;
    const size_t esz = in[0].dtype == DT_F32 ? sizeof(float) : sizeof(double);
    size_t at = 0;  // offset along the concatenated dimension
    for (size_t k = 0; k < n_inputs; ++k) {
        const size_t w = inner * shapes[k][dz];  // contiguous elements per outer index
        if (w && outer) {
            char* dst = (char*)ob.alloc->ptr + at * inner * esz;
            const hipError_t e = hipMemcpy2DAsync(dst, inner * cat_extent * esz, in[k].alloc->ptr, w * esz, w * esz, outer, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) {
                rmhip_free(ctx, *out);
                return fail(RMHIP_ERR_HIP, "cat copy: %s", hipGetErrorString(e));
            }
        }
        at += shapes[k][dz];
    }
    return RMHIP_OK;
}

}  // extern "C"
