// rmhip_core.cpp -- context, buffer table, device-memory pool, telemetry and the memory half of the
// C ABI (include/rmhip.h).  Mirrors the provider-side bookkeeping the reference keeps per
// provider: buffer table + residency pool (crates/runmat-accelerate/src/backend/wgpu/provider,
// SURVEY.md 2.1) and the InProcessProvider registry (simple_provider.rs:678-728).
#include <chrono>
#include <cstring>

#include "common.h"

namespace rmhip {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int fail(int code, const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

ScopedTimer::ScopedTimer(std::atomic<uint64_t>* c, std::atomic<uint64_t>* n) : count(c), ns(n), t0(now_ns()) {}
ScopedTimer::~ScopedTimer() {
    (*count)++;
    (*ns) += now_ns() - t0;
}

// ---- pool --------------------------------------------------------------------------------------
// Buckets: round up to 256 B below 1 MiB, to 1 MiB above. Frees go back to the pool (bounded), so
// the steady state of "every op returns a new buffer" costs no hipMalloc.
static size_t bucket_bytes(size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (bytes < (1u << 20)) return (bytes + 255) & ~(size_t)255;
    return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
}

Allocation::~Allocation() {
    if (ptr && !external && ctx) ctx->release_device(ptr, bytes);
}

int Context::alloc_device(size_t numel, std::shared_ptr<Allocation>* out) {
    const size_t bytes = bucket_bytes(numel * sizeof(double));
    double* p = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = pool.find(bytes);
        if (it != pool.end()) {
            p = it->second;
            pool.erase(it);
            pooled_bytes -= bytes;
            tel.bytes_pooled = pooled_bytes;
        }
    }
    if (!p) {
        hipError_t e = hipMalloc((void**)&p, bytes);
        if (e != hipSuccess) {
            // drop the pool and retry once
            std::vector<double*> victims;
            {
                std::lock_guard<std::mutex> lk(mu);
                for (auto& kv : pool) victims.push_back(kv.second);
                pool.clear();
                pooled_bytes = 0;
                tel.bytes_pooled = 0;
            }
            (void)hipStreamSynchronize(stream);
            for (double* v : victims) (void)hipFree(v);
            e = hipMalloc((void**)&p, bytes);
            if (e != hipSuccess) return fail(RMHIP_ERR_OOM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        }
        tel.bytes_allocated += bytes;
    }
    auto a = std::make_shared<Allocation>();
    a->ctx = this;
    a->ptr = p;
    a->bytes = bytes;
    *out = std::move(a);
    return RMHIP_OK;
}

void Context::release_device(double* ptr, size_t bytes) {
    // All work is stream-ordered on one stream, so a pooled block can be handed out again
    // immediately: any later kernel that writes it is ordered after the kernels that read it.
    std::unique_lock<std::mutex> lk(mu);
    if (pooled_bytes + bytes <= pool_limit_bytes) {
        pool.emplace(bytes, ptr);
        pooled_bytes += bytes;
        tel.bytes_pooled = pooled_bytes;
        return;
    }
    lk.unlock();
    (void)hipStreamSynchronize(stream);
    (void)hipFree(ptr);
    tel.bytes_allocated -= bytes;
}

int Context::register_buffer(Buffer&& b, uint64_t* id) {
    std::lock_guard<std::mutex> lk(mu);
    *id = next_id++;
    if (b.lazy()) ++n_lazy;
    if (b.rng_lazy) ++n_rng_lazy;
    table.emplace(*id, std::move(b));
    return RMHIP_OK;
}

int Context::new_buffer(const size_t* shape, size_t rank, uint64_t* id, Buffer* out) {
    Buffer b;
    b.shape.assign(shape, shape + rank);
    b.numel = shape_numel(shape, rank);
    RMHIP_TRY(alloc_device(b.numel, &b.alloc));
    if (out) *out = b;
    RMHIP_TRY(register_buffer(std::move(b), id));
    if (precision == 32) narrow_pending.push_back(*id);
    return RMHIP_OK;
}

int Context::new_buffer_complex(const size_t* shape, size_t rank, uint64_t* id, Buffer* out) {
    Buffer b;
    b.shape.assign(shape, shape + rank);
    b.numel = shape_numel(shape, rank);
    b.cplx = true;
    RMHIP_TRY(alloc_device(2 * b.numel, &b.alloc));
    if (out) *out = b;
    return register_buffer(std::move(b), id);  // f64 storage in either precision mode: never queued for narrowing
}

int Context::get_any(uint64_t id, Buffer* out) {
    RMHIP_TRY(lookup(id, out));
    return out->cplx ? RMHIP_OK : get(id, out);
}

int Context::new_buffer_f32(const size_t* shape, size_t rank, uint64_t* id, Buffer* out) {
    Buffer b;
    b.shape.assign(shape, shape + rank);
    b.numel = shape_numel(shape, rank);
    b.dtype = DT_F32;
    RMHIP_TRY(alloc_device((b.numel + 1) / 2, &b.alloc));
    if (out) *out = b;
    return register_buffer(std::move(b), id);
}

int Context::get_raw(uint64_t id, Buffer* out, bool keep_rng) {
    RMHIP_TRY(lookup(id, out));
    if (out->cplx) return fail(RMHIP_ERR_UNSUPPORTED, "complex-interleaved tensor %llu: this entry point takes real tensors", (unsigned long long)id);
    if (out->rng_lazy && !keep_rng) {
        RMHIP_TRY(settle_rng(id));
        RMHIP_TRY(lookup(id, out));
    }
    return RMHIP_OK;
}

int Context::settle_rng(uint64_t id) {
    Buffer raw;
    RMHIP_TRY(lookup(id, &raw));
    if (!raw.rng_lazy) return RMHIP_OK;
    RMHIP_TRACEF("materialise lazy random_normal id %llu numel %zu", (unsigned long long)id, raw.numel);
    std::shared_ptr<Allocation> fresh;
    RMHIP_TRY(alloc_device(raw.numel ? raw.numel : 1, &fresh));
    RMHIP_TRY(launch_rng_normal(this, raw.rng_state, fresh->ptr, raw.numel));
    lazy_randn_materialised++;
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find(id);
    if (it != table.end() && it->second.rng_lazy) {
        it->second.alloc = fresh;
        it->second.rng_lazy = false;
        --n_rng_lazy;
    }
    return RMHIP_OK;
}

int Context::lookup(uint64_t id, Buffer* out) {
    Buffer rec;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = table.find(id);
        if (it == table.end()) return fail(RMHIP_ERR_NOT_FOUND, "buffer not found: %llu", (unsigned long long)id);
        rec = it->second;
    }
    // assigned outside the lock: `*out` may hold the last reference to a temporary (a widened or materialised copy),
    // whose release takes `mu` again
    *out = std::move(rec);
    return RMHIP_OK;
}

int Context::get_view(uint64_t id, Buffer* out) {
    RMHIP_TRY(get_raw(id, out));
    if (!out->rep_base.empty()) {  // a consumer that cannot index a repmat view: tile it once, keep the copy under this id
        RMHIP_TRY(settle_view(id));
        RMHIP_TRY(get_raw(id, out));
    }
    if (out->dtype == DT_F64) return RMHIP_OK;
    RMHIP_TRACEF("widen id %llu numel %zu tview %d", (unsigned long long)id, out->numel, (int)out->tview);
    // f32 storage read by an f64 kernel: widen into a temporary that lives as long as the caller's Buffer copy
    std::shared_ptr<Allocation> wide;
    RMHIP_TRY(alloc_device(out->numel ? out->numel : 1, &wide));
    RMHIP_TRY(launch_widen(this, out->data_f32(), wide->ptr, out->numel));
    out->alloc = wide;
    out->dtype = DT_F64;
    return RMHIP_OK;
}

int Context::settle_view(uint64_t id) {
    Buffer raw;
    RMHIP_TRY(get_raw(id, &raw));
    if (!raw.rep_base.empty()) {
        RMHIP_TRACEF("materialise repmat view id %llu numel %zu (base %zu)", (unsigned long long)id, raw.numel, raw.stored_numel());
        std::shared_ptr<Allocation> tiled;
        const size_t words = raw.dtype == DT_F32 ? (raw.numel + 1) / 2 : raw.numel;
        RMHIP_TRY(alloc_device(words ? words : 1, &tiled));
        RMHIP_TRY(materialize_repmat(this, raw, tiled->ptr));
        std::lock_guard<std::mutex> lk(mu);
        auto it = table.find(id);
        if (it != table.end() && !it->second.rep_base.empty() && it->second.alloc == raw.alloc) {
            it->second.alloc = tiled;  // `raw` still references the base storage: nothing is released under the lock
            it->second.rep_base.clear();
            --n_lazy;
        }
        return RMHIP_OK;
    }
    if (!raw.tview) return RMHIP_OK;
    const size_t R = raw.shape[0], C = raw.shape[1];  // logical R x C, storage = base C x R
    RMHIP_TRACEF("materialise view id %llu %zux%zu (%s storage)", (unsigned long long)id, R, C, raw.dtype == DT_F32 ? "f32" : "f64");
    std::shared_ptr<Allocation> fresh;
    if (raw.dtype == DT_F32) {
        RMHIP_TRY(alloc_device((raw.numel + 1) / 2 ? (raw.numel + 1) / 2 : 1, &fresh));
        RMHIP_TRY(transpose_device_f32(this, raw.data_f32(), C, C, R, reinterpret_cast<float*>(fresh->ptr), R));
    } else {
        RMHIP_TRY(alloc_device(raw.numel ? raw.numel : 1, &fresh));
        RMHIP_TRY(transpose_device(this, raw.data(), C, C, R, fresh->ptr, R));
    }
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find(id);
    if (it != table.end() && it->second.tview && it->second.alloc == raw.alloc) {
        it->second.alloc = fresh;  // `raw` still references the base storage: nothing is released under the lock
        it->second.tview = false;
        --n_lazy;
    }
    return RMHIP_OK;
}

int Context::detach_views_of(uint64_t id) {
    std::vector<uint64_t> sharers;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = table.find(id);
        // O(1) in the common case: no lazy view is alive in this context (the count is kept where records enter / leave the table and
        // where views are settled).  The earlier `use_count() <= 1` test never fired - every caller holds a copy of the record.
        if (n_lazy == 0 || it == table.end() || !it->second.alloc) return RMHIP_OK;
        for (auto& kv : table)
            if (kv.first != id && kv.second.alloc == it->second.alloc && kv.second.lazy()) sharers.push_back(kv.first);
    }
    for (uint64_t v : sharers) RMHIP_TRY(settle_view(v));
    return RMHIP_OK;
}

int Context::narrow(uint64_t id) {
    Buffer b;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = table.find(id);
        if (it == table.end()) return RMHIP_OK;  // freed on an error path
        b = it->second;
    }
    if (b.dtype != DT_F64 || !b.alloc || b.alloc->external) return RMHIP_OK;
    if (!b.rep_base.empty()) return RMHIP_OK;  // a view of somebody else's storage: not this entry point's to convert
    RMHIP_TRACEF("narrow id %llu numel %zu", (unsigned long long)id, b.numel);
    std::shared_ptr<Allocation> slim;
    RMHIP_TRY(alloc_device((b.numel + 1) / 2 ? (b.numel + 1) / 2 : 1, &slim));
    RMHIP_TRY(launch_narrow(this, b.data(), reinterpret_cast<float*>(slim->ptr), b.numel));
    std::lock_guard<std::mutex> lk(mu);
    // every alias of this storage created inside the same entry point (reshape of a fresh result) moves with it
    for (auto& kv : table)
        if (kv.second.alloc == b.alloc && kv.second.dtype == DT_F64) {
            kv.second.alloc = slim;
            kv.second.dtype = DT_F32;
        }
    return RMHIP_OK;
}

void Context::finish_outputs(size_t mark) {
    for (size_t i = mark; i < narrow_pending.size(); ++i) (void)narrow(narrow_pending[i]);  // on failure the buffer stays f64: still valid
    narrow_pending.resize(mark);
}

int Context::get(uint64_t id, Buffer* out) {
    // a consumer that cannot address a transposed operand: materialise once and keep the plain copy under this id
    RMHIP_TRY(get_raw(id, out));
    if (out->lazy()) RMHIP_TRY(settle_view(id));
    return get_view(id, out);
}

void Context::record_launch(const char* kernel, std::initializer_list<std::pair<const char*, uint64_t>> shape,
                            std::initializer_list<std::pair<const char*, uint64_t>> tuning) {
    LaunchRecord& r = launch_log[launch_seq++ % kLaunchLog];
    r.kernel = kernel;
    r.bits = precision;
    r.n_shape = r.n_tuning = 0;
    for (const auto& kv : shape)
        if (r.n_shape < 6) {
            r.shape_key[r.n_shape] = kv.first;
            r.shape_val[r.n_shape++] = kv.second;
        }
    for (const auto& kv : tuning)
        if (r.n_tuning < 6) {
            r.tuning_key[r.n_tuning] = kv.first;
            r.tuning_val[r.n_tuning++] = kv.second;
        }
}

void Context::record_solve_fallback(const char* reason) {
    for (auto& kv : solve_fallbacks)
        if (std::strcmp(kv.first, reason) == 0) {
            kv.second++;
            return;
        }
    solve_fallbacks.emplace_back(reason, 1);
}

void Context::ensure_max_lds(const void* kernel, size_t bytes) {
    for (const void* k : lds_opt_in)
        if (k == kernel) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    lds_opt_in.push_back(kernel);
}

int Context::ensure_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return RMHIP_OK;
    if (scratch) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(scratch);
        scratch = nullptr;
        scratch_bytes = 0;
    }
    size_t want = bucket_bytes(bytes);
    RMHIP_HIP_CHECK(hipMalloc((void**)&scratch, want));
    scratch_bytes = want;
    return RMHIP_OK;
}

}  // namespace rmhip

using namespace rmhip;

struct rmhip_ctx {
    Context c;
};

#define CTX_OR_FAIL(ctx)                                                  \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");           \
    Context* c = &(ctx)->c;                                               \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);              \
    DeviceGuard _dg(c);                                                   \
    NarrowScope _ns(c)

// Which XCD does workgroup b of a launch run on?  The LU places small panels on ONE XCD by launching 8x the grid and keeping
// workgroups b % 8 == 0, and its late-phase update kernels leave the panel's XCD: both assume eight dies and a round-robin
// dispatcher.  A CPX partition (one die, 32 CUs) passes the gfx950 check and breaks both - every update workgroup would leave.
// Probe instead of assuming: the placement tricks stay on only for exactly 8 dies visited round robin.
namespace {
__global__ void k_probe_xcc(int* out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (int)(xcc & 0xf);
    }
}
void probe_xcds(rmhip::Context* c) {
    constexpr int kBlocks = 64;
    int* d = nullptr;
    int h[kBlocks];
    c->num_xcc = 1;
    c->one_xcd_ok = false;
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess) return;
    hipLaunchKernelGGL(k_probe_xcc, dim3(kBlocks), dim3(64), 0, c->stream, d);
    const bool ok = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
    (void)hipFree(d);
    if (!ok) {
        (void)hipGetLastError();
        return;
    }
    bool seen[16] = {false};
    int n = 0;
    for (int b = 0; b < kBlocks; ++b)
        if (!seen[h[b] & 15]) {
            seen[h[b] & 15] = true;
            ++n;
        }
    bool round_robin = n == 8;
    for (int b = 8; b < kBlocks && round_robin; ++b) round_robin = h[b] == h[b - 8];
    c->num_xcc = n;
    c->one_xcd_ok = round_robin;
}
}  // namespace

extern "C" {

const char* rmhip_version(void) { return "rmhip 0.1.0 (gfx950)"; }
const char* rmhip_last_error(void) { return g_last_error.c_str(); }

int rmhip_init(int device_ordinal, rmhip_ctx** out_ctx) {
    if (!out_ctx) return fail(RMHIP_ERR_INVALID, "null out_ctx");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(RMHIP_ERR_NO_DEVICE, "no HIP device available (%s); librmhip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(RMHIP_ERR_INVALID, "device ordinal %d out of range (0..%d)", device_ordinal, count - 1);
    RMHIP_HIP_CHECK(hipSetDevice(device_ordinal));
    auto* h = new rmhip_ctx();
    Context* c = &h->c;
    c->device = device_ordinal;
    e = hipGetDeviceProperties(&c->props, device_ordinal);
    if (e != hipSuccess) {
        delete h;
        return fail(RMHIP_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    if (std::strncmp(c->props.gcnArchName, "gfx950", 6) != 0 && !std::getenv("RMHIP_ALLOW_ANY_ARCH")) {
        std::string arch = c->props.gcnArchName;
        delete h;
        return fail(RMHIP_ERR_NO_DEVICE, "device %d is %s; librmhip kernels are built for gfx950 only",
                    device_ordinal, arch.c_str());
    }
    c->num_cus = c->props.multiProcessorCount > 0 ? c->props.multiProcessorCount : 256;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(RMHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->owns_stream = true;
    (void)hipEventCreate(&c->ev_begin);
    (void)hipEventCreate(&c->ev_end);
    // Keep at most a quarter of HBM parked in the pool (288 GB parts: plenty for 512 MiB operands).
    c->pool_limit_bytes = (size_t)(c->props.totalGlobalMem / 4);
    if (const char* v = std::getenv("RMHIP_POOL_LIMIT_MB")) c->pool_limit_bytes = (size_t)std::atoll(v) << 20;
    if (const char* v = std::getenv("RMHIP_LAZY_RANDN")) c->lazy_randn = *v != '0';
    if (const char* v = std::getenv("RMHIP_LAZY_RANDN_MIN")) c->lazy_randn_min = (size_t)std::atoll(v);
    probe_xcds(c);
    *out_ctx = h;
    return RMHIP_OK;
}

int rmhip_shutdown(rmhip_ctx* ctx) {
    if (!ctx) return RMHIP_OK;
    Context* c = &ctx->c;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->kernel_cache.clear();
    }
    c->pool_limit_bytes = 0;  // frees bypass the pool from here on
    c->table.clear();
    c->n_lazy = 0;
    c->n_rng_lazy = 0;
    c->fft_tables.clear();  // (cached twiddle / chirp tables hold allocations of this context: released while it is still whole)
    for (auto& kv : c->pool) (void)hipFree(kv.second);
    c->pool.clear();
    if (c->scratch) (void)hipFree(c->scratch);
    for (hipEvent_t e : c->lu_events)
        if (e) (void)hipEventDestroy(e);
    if (c->lu_side_stream) (void)hipStreamDestroy(c->lu_side_stream);
    if (c->lu_prep_stream) (void)hipStreamDestroy(c->lu_prep_stream);
    if (c->lu_aux_stream) (void)hipStreamDestroy(c->lu_aux_stream);
    if (c->lu_far_stream) (void)hipStreamDestroy(c->lu_far_stream);
    if (c->lu_mid_stream) (void)hipStreamDestroy(c->lu_mid_stream);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete ctx;
    return RMHIP_OK;
}

int rmhip_device_info(rmhip_ctx* ctx, rmhip_device_info_t* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    std::memset(out, 0, sizeof *out);
    // (hipDeviceProp_t::name comes back empty on some driver / container combinations: say at least what the part is)
    if (c->props.name[0]) std::snprintf(out->name, sizeof out->name, "%s", c->props.name);
    else std::snprintf(out->name, sizeof out->name, "AMD Instinct (%s, %d CUs)", c->props.gcnArchName, c->num_cus);
    std::snprintf(out->vendor, sizeof out->vendor, "AMD");
    std::snprintf(out->backend, sizeof out->backend, "hip");
    out->xcd_count = c->num_xcc;
    std::snprintf(out->arch, sizeof out->arch, "%s", c->props.gcnArchName);
    out->device_ordinal = c->device;
    out->compute_units = c->num_cus;
    out->wavefront_size = c->props.warpSize;
    out->clock_mhz = c->props.clockRate / 1000;
    out->memory_clock_khz = c->props.memoryClockRate;
    out->memory_bus_width_bits = c->props.memoryBusWidth;
    out->l2_cache_bytes = c->props.l2CacheSize;
    out->total_memory_bytes = c->props.totalGlobalMem;
    out->precision_bits = c->precision;
    out->reduction_workgroup_size = 256;
    out->two_pass_threshold = 262144;
    return RMHIP_OK;
}

int rmhip_set_stream(rmhip_ctx* ctx, void* hip_stream) {
    CTX_OR_FAIL(ctx);
    (void)hipStreamSynchronize(c->stream);
    if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->owns_stream = false;
    return RMHIP_OK;
}

int rmhip_set_precision(rmhip_ctx* ctx, int bits) {
    CTX_OR_FAIL(ctx);
    if (bits != 32 && bits != 64) return fail(RMHIP_ERR_INVALID, "precision must be 32 or 64 bits, got %d", bits);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->table.empty() && bits != c->precision)
        return fail(RMHIP_ERR_INVALID, "precision is a property of the provider: set it before the first buffer exists");
    c->precision = bits;
    return RMHIP_OK;
}

int rmhip_buffer_bits(rmhip_ctx* ctx, rmhip_buf id, int* bits) {
    CTX_OR_FAIL(ctx);
    if (!bits) return fail(RMHIP_ERR_INVALID, "null bits");
    Buffer b;
    RMHIP_TRY(c->get_raw(id, &b));
    *bits = b.dtype == DT_F32 ? 32 : 64;
    return RMHIP_OK;
}

void* rmhip_get_stream(rmhip_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

int rmhip_synchronize(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RMHIP_OK;
}

int rmhip_upload(rmhip_ctx* ctx, const double* host, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (rank && !shape)) return fail(RMHIP_ERR_INVALID, "upload: null argument");
    const size_t n = shape_numel(shape, rank);
    if (n && !host) return fail(RMHIP_ERR_INVALID, "upload: null host data");
    Buffer b;
    RMHIP_TRY(c->new_buffer(shape, rank, out, &b));
    if (n) {
        hipError_t e = hipMemcpyAsync(b.data(), host, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // host buffer may be reused on return
        if (e != hipSuccess) {
            rmhip_free(ctx, *out);
            return fail(RMHIP_ERR_HIP, "upload memcpy: %s", hipGetErrorString(e));
        }
    }
    c->tel.upload_bytes += n * sizeof(double);
    return RMHIP_OK;
}

int rmhip_download(rmhip_ctx* ctx, rmhip_buf id, double* out_host, size_t n) {
    CTX_OR_FAIL(ctx);
    Buffer b;
    RMHIP_TRY(c->get_any(id, &b));
    // a complex-interleaved tensor comes back as 2 * numel doubles (re, im, ...), as `HostTensorOwned` carries it (lib.rs:3362-3366)
    if (n != (b.cplx ? 2 * b.numel : b.numel))
        return fail(RMHIP_ERR_SHAPE, "download: expected %zu elements, got %zu", b.cplx ? 2 * b.numel : b.numel, n);
    if (n && !out_host) return fail(RMHIP_ERR_INVALID, "download: null destination");
    if (n) {
        RMHIP_HIP_CHECK(hipMemcpyAsync(out_host, b.data(), n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->tel.download_bytes += n * sizeof(double);
    return RMHIP_OK;
}

int rmhip_free(rmhip_ctx* ctx, rmhip_buf id) {
    CTX_OR_FAIL(ctx);
    Buffer victim;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->table.find(id);
        if (it == c->table.end()) return fail(RMHIP_ERR_NOT_FOUND, "free: buffer not found: %llu", (unsigned long long)id);
        if (it->second.lazy()) --c->n_lazy;
        if (it->second.rng_lazy) --c->n_rng_lazy;
        victim = std::move(it->second);
        c->table.erase(it);
    }
    return RMHIP_OK;  // `victim` drops the allocation reference outside the lock
}

int rmhip_storage(rmhip_ctx* ctx, rmhip_buf id, int* complex_interleaved) {
    CTX_OR_FAIL(ctx);
    if (!complex_interleaved) return fail(RMHIP_ERR_INVALID, "null result");
    Buffer b;
    RMHIP_TRY(c->lookup(id, &b));
    *complex_interleaved = b.cplx ? 1 : 0;
    return RMHIP_OK;
}

int rmhip_shape(rmhip_ctx* ctx, rmhip_buf id, size_t* rank_inout, size_t* shape_out) {
    CTX_OR_FAIL(ctx);
    if (!rank_inout) return fail(RMHIP_ERR_INVALID, "null rank");
    Buffer b;
    RMHIP_TRY(c->lookup(id, &b));
    if (*rank_inout < b.shape.size() || !shape_out) {
        *rank_inout = b.shape.size();
        return shape_out ? fail(RMHIP_ERR_INVALID, "shape buffer too small") : RMHIP_OK;
    }
    for (size_t i = 0; i < b.shape.size(); ++i) shape_out[i] = b.shape[i];
    *rank_inout = b.shape.size();
    return RMHIP_OK;
}

int rmhip_numel(rmhip_ctx* ctx, rmhip_buf id, size_t* out) {
    CTX_OR_FAIL(ctx);
    Buffer b;
    RMHIP_TRY(c->lookup(id, &b));
    *out = b.numel;
    return RMHIP_OK;
}

int rmhip_fill(rmhip_ctx* ctx, double value, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Buffer b;
    RMHIP_TRY(c->new_buffer(shape, rank, out, &b));
    int rc = launch_fill(c, b.data(), b.numel, value);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_fill_uniform(rmhip_ctx* ctx, uint64_t seed, double lo, double hi, const size_t* shape, size_t rank,
                       rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Buffer b;
    RMHIP_TRY(c->new_buffer(shape, rank, out, &b));
    int rc = launch_fill_uniform(c, b.data(), b.numel, seed, lo, hi);
    if (rc) rmhip_free(ctx, *out);
    return rc;
}

int rmhip_reshape(rmhip_ctx* ctx, rmhip_buf id, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out || (rank && !shape)) return fail(RMHIP_ERR_INVALID, "reshape: null argument");
    Buffer b;
    RMHIP_TRY(c->get_raw(id, &b));
    if (shape_numel(shape, rank) != b.numel)
        return fail(RMHIP_ERR_SHAPE, "reshape: element count mismatch (%zu vs %zu)", shape_numel(shape, rank), b.numel);
    // Same buffer, new shape: the trait default (lib.rs:2676-2684) and the wgpu provider (ops/tensor.rs reshape_exec)
    // return the SAME buffer_id, and callers such as the reshape builtin consume the source handle without freeing
    // it - a second table entry would orphan the first and pin the allocation.  A transpose view is materialised first
    // (the bytes of a view are those of its base matrix).
    if (b.lazy()) RMHIP_TRY(c->settle_view(id));
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->table.find(id);
        if (it == c->table.end()) return fail(RMHIP_ERR_NOT_FOUND, "buffer not found: %llu", (unsigned long long)id);
        it->second.shape.assign(shape, shape + rank);
    }
    *out = id;
    return RMHIP_OK;
}

int rmhip_wrap_external(rmhip_ctx* ctx, void* device_ptr, const size_t* shape, size_t rank, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!device_ptr && shape_numel(shape, rank)) return fail(RMHIP_ERR_INVALID, "wrap_external: null pointer");
    Buffer b;
    b.alloc = std::make_shared<Allocation>();
    b.alloc->ctx = c;
    b.alloc->ptr = (double*)device_ptr;
    b.alloc->external = true;
    b.shape.assign(shape, shape + rank);
    b.numel = shape_numel(shape, rank);
    b.alloc->bytes = b.numel * sizeof(double);
    return c->register_buffer(std::move(b), out);
}

void* rmhip_device_ptr(rmhip_ctx* ctx, rmhip_buf id) {
    if (!ctx) return nullptr;
    std::lock_guard<std::recursive_mutex> _call(ctx->c.call_mu);
    DeviceGuard _dg(&ctx->c);
    Buffer b;
    if (ctx->c.get_raw(id, &b) != RMHIP_OK) return nullptr;
    if (ctx->c.detach_views_of(id) != RMHIP_OK) return nullptr;  // the caller may write through the pointer: views of this storage keep their values
    if (b.dtype == DT_F32 && !b.rep_base.empty()) {  // tile first: the pointer must address numel elements
        if (ctx->c.settle_view(id) != RMHIP_OK || ctx->c.get_raw(id, &b) != RMHIP_OK) return nullptr;
    }
    if (b.dtype == DT_F32) return b.tview ? nullptr : (void*)b.data();  // the f32 storage itself (rmhip_buffer_bits says which)
    if (ctx->c.get(id, &b) != RMHIP_OK) return nullptr;
    return b.data();
}

int rmhip_telemetry(rmhip_ctx* ctx, rmhip_telemetry_t* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Telemetry& t = c->tel;
    out->fused_elementwise_count = t.fused_elementwise_count;
    out->fused_elementwise_ns = t.fused_elementwise_ns;
    out->fused_reduction_count = t.fused_reduction_count;
    out->fused_reduction_ns = t.fused_reduction_ns;
    out->matmul_count = t.matmul_count;
    out->matmul_ns = t.matmul_ns;
    out->mldivide_count = t.mldivide_count;
    out->mldivide_ns = t.mldivide_ns;
    out->upload_bytes = t.upload_bytes;
    out->download_bytes = t.download_bytes;
    out->fusion_cache_hits = t.cache_hits;
    out->fusion_cache_misses = t.cache_misses;
    out->kernel_launches = t.kernel_launches;
    out->bytes_allocated = t.bytes_allocated;
    out->bytes_pooled = t.bytes_pooled;
    out->linsolve_count = t.linsolve_count;
    out->linsolve_ns = t.linsolve_ns;
    out->mrdivide_count = t.mrdivide_count;
    out->mrdivide_ns = t.mrdivide_ns;
    return RMHIP_OK;
}

int rmhip_telemetry_solve_fallback(rmhip_ctx* ctx, size_t index, char* reason, size_t cap, uint64_t* count) {
    CTX_OR_FAIL(ctx);
    if (index >= c->solve_fallbacks.size()) return fail(RMHIP_ERR_NOT_FOUND, "solve_fallbacks: index %zu of %zu", index, c->solve_fallbacks.size());
    if (reason && cap) std::snprintf(reason, cap, "%s", c->solve_fallbacks[index].first);
    if (count) *count = c->solve_fallbacks[index].second;
    return RMHIP_OK;
}

int rmhip_telemetry_kernel_launch(rmhip_ctx* ctx, size_t index, rmhip_kernel_launch_t* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    const uint64_t have = c->launch_seq < (uint64_t)kLaunchLog ? c->launch_seq : (uint64_t)kLaunchLog;
    if (index >= have) return fail(RMHIP_ERR_NOT_FOUND, "kernel_launches: index %zu of %llu", index, (unsigned long long)have);
    const LaunchRecord& r = c->launch_log[(c->launch_seq - have + index) % kLaunchLog];  // oldest first
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->kernel, sizeof out->kernel, "%s", r.kernel ? r.kernel : "");
    std::snprintf(out->precision, sizeof out->precision, "%s", r.bits == 32 ? "f32" : "f64");
    out->n_shape = (uint32_t)r.n_shape;
    out->n_tuning = (uint32_t)r.n_tuning;
    for (int i = 0; i < r.n_shape; ++i) {
        std::snprintf(out->shape[i].key, sizeof out->shape[i].key, "%s", r.shape_key[i]);
        out->shape[i].value = r.shape_val[i];
    }
    for (int i = 0; i < r.n_tuning; ++i) {
        std::snprintf(out->tuning[i].key, sizeof out->tuning[i].key, "%s", r.tuning_key[i]);
        out->tuning[i].value = r.tuning_val[i];
    }
    return RMHIP_OK;
}

int rmhip_lu_stats(rmhip_ctx* ctx, rmhip_lu_stats_t* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    std::memset(out, 0, sizeof *out);
    out->solve_path_factorizations = c->lu_fast_count;
    out->pivot_growth_fallbacks = c->lu_growth_fallbacks;
    out->panel_exchange_timeouts = c->lu_exchange_timeouts;
    out->subst_chain_timeouts = c->lu_subst_timeouts;
    out->last_max_multiplier = c->lu_last_growth;
    out->tau = c->lu_tau;
    out->one_xcd_panels = c->one_xcd_ok ? 1 : 0;
    out->conservative_panels = c->lu_conservative ? 1 : 0;
    out->svd_solves = c->svd_solves;
    return RMHIP_OK;
}

int rmhip_reset_telemetry(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    Telemetry& t = c->tel;
    t.fused_elementwise_count = t.fused_elementwise_ns = 0;
    t.fused_reduction_count = t.fused_reduction_ns = 0;
    t.matmul_count = t.matmul_ns = 0;
    t.mldivide_count = t.mldivide_ns = 0;
    t.upload_bytes = t.download_bytes = 0;
    t.cache_hits = t.cache_misses = 0;
    t.kernel_launches = 0;
    t.linsolve_count = t.linsolve_ns = 0;
    t.mrdivide_count = t.mrdivide_ns = 0;
    c->solve_fallbacks.clear();  // telemetry.rs:95-125: reset clears the fallback map and the launch log
    c->launch_seq = 0;
    return RMHIP_OK;
}

int rmhip_timer_begin(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    RMHIP_HIP_CHECK(hipEventRecord(c->ev_begin, c->stream));
    return RMHIP_OK;
}

int rmhip_timer_end(rmhip_ctx* ctx, double* elapsed_ms) {
    CTX_OR_FAIL(ctx);
    RMHIP_HIP_CHECK(hipEventRecord(c->ev_end, c->stream));
    RMHIP_HIP_CHECK(hipEventSynchronize(c->ev_end));
    float ms = 0.f;
    RMHIP_HIP_CHECK(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return RMHIP_OK;
}

}  // extern "C"

// rmhip_ctx is an opaque wrapper; the other translation units reach the Context through this.
namespace rmhip {
Context* context_of(rmhip_ctx* h) { return h ? &h->c : nullptr; }
}  // namespace rmhip
