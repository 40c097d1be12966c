// comm.cpp -- the collective layer behind the C ABI (include/rmhip.h "multi-GPU collectives"): what a non-Python host
// needs to shard the hot path the way SURVEY.md 8(e) lays it out - one process per GPU, a row-block all-gather for a
// replicated C = A*B, a one-value ordered exchange for reductions / Monte-Carlo, a panel broadcast per block column of
// the block-cyclic A\b.  The reference has no multi-device code at all: nothing here mirrors it.
//
// Two transports behind the same entry points, chosen by the id rank 0 creates (rmhip_comm_unique_id):
//   * RCCL (one rank per GPU, xGMI): librccl is loaded on first use with dlopen - the library proper does not link it -
//     and every collective is enqueued on a HIP stream of the context (the call stream, or the communication stream
//     for the asynchronous form).
//   * host shared memory (several ranks on ONE GPU, or a box without xGMI peers: the control-flow tests): a POSIX
//     shared-memory segment with one staging slot per rank and a sense-reversing barrier; device -> slot -> device.
//     Slow and node-local by construction; it exists so that the same entry points run on the single-GPU test box.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

#include "common.h"
#include <memory>
#include <mutex>
#include <condition_variable>

namespace rmhip {

namespace {

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------------
// The handful of NCCL-API types and constants used here, declared locally (they are fixed by the NCCL ABI: rccl.h
// `ncclUniqueId` = 128 opaque bytes, `ncclResult_t` 0 = success, `ncclDataType_t` 8 = f64): the library is only ever loaded at run
// time, so it should not need RCCL's development headers to BUILD either.
struct ncclComm;
typedef ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclResult_t ncclInProgress = 7;  // (nccl.h: ncclInProgress)
static constexpr ncclDataType_t ncclFloat64 = 8;

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string load_error;  // why `ok` is false: captured ONCE where it happened (dlerror() clears itself when read)
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // RCCL must sit on the SAME HIP runtime as this library.  A host process can hold two (PyTorch wheels bundle
        // their own libamdhip64 / librccl next to the system ROCm): a librccl bound to the other runtime reports
        // "no ROCm-capable device".  So: first the librccl that lives beside the libamdhip64 our HIP entry points resolve to,
        // then the loader's default search.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string& name : names) {
            api.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char* e = dlerror();
            api.load_error = e ? e : "dlopen failed";
        }
        if (!api.handle) return;
        api.load_error.clear();
        auto sym = [&](const char* n) { return dlsym(api.handle, n); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.CommGetAsyncError = (decltype(api.CommGetAsyncError))sym("ncclCommGetAsyncError");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.Broadcast && api.GetErrorString;
        if (!api.ok) api.load_error = "librccl lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllGather / Broadcast / GetErrorString";
    });
    return api;
}

int nccl_fail(const char* what, ncclResult_t r) {
    return fail(RMHIP_ERR_HIP, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
}

// ---- host shared-memory transport ------------------------------------------------------------------------------------
static constexpr char kShmMagic[8] = {'R', 'M', 'H', 'S', 'H', 'M', '0', '1'};
static constexpr size_t kShmSlotBytes = (size_t)8 << 20;  // staging per rank and round
struct ShmHeader {
    std::atomic<uint32_t> ready;     // set by rank 0 once the header is initialised
    std::atomic<uint32_t> attached;  // ranks that mapped the segment
    std::atomic<uint32_t> arrived;   // barrier: arrivals of the current generation
    std::atomic<uint32_t> generation;
    uint32_t world;
    std::atomic<uint32_t> aborted;   // a rank gave up inside a collective sequence (rmhip_comm_abort): every barrier fails from now on
    uint32_t pad[10];
};
static_assert(sizeof(ShmHeader) == 64, "header is one cache line");

}  // namespace

struct Comm {
    int rank = 0, world = 1;
    bool host = false;
    // RCCL
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;       // communication stream of the asynchronous form
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool pending = false;
    bool aborted = false;  // rmhip_comm_abort was called here: every later collective fails at once
    // host shared memory
    std::string shm_name;
    ShmHeader* hdr = nullptr;
    char* slots = nullptr;
    size_t map_bytes = 0;
};

namespace {

int shm_barrier(Comm* cm) {
    ShmHeader* h = cm->hdr;
    if (h->aborted.load(std::memory_order_acquire)) return fail(RMHIP_ERR_HIP, "comm: a rank aborted the communicator");
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)cm->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return RMHIP_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
        if (++spins > 2000) {
            if (h->aborted.load(std::memory_order_acquire)) return fail(RMHIP_ERR_HIP, "comm: a rank aborted the communicator");
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                return fail(RMHIP_ERR_HIP, "comm: a rank did not reach the host barrier within 60 s");
        }
    }
    return RMHIP_OK;
}

char* shm_slot(Comm* cm, int rank) { return cm->slots + (size_t)rank * kShmSlotBytes; }

// every rank contributes `bytes` from `send`; `recv` receives world blocks in rank order
int shm_allgather(Context* c, Comm* cm, const void* send, void* recv, size_t bytes) {
    for (size_t off = 0; off < bytes || off == 0; off += kShmSlotBytes) {
        const size_t n = bytes - off < kShmSlotBytes ? bytes - off : kShmSlotBytes;
        if (n) RMHIP_HIP_CHECK(hipMemcpyAsync(shm_slot(cm, cm->rank), (const char*)send + off, n, hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        RMHIP_TRY(shm_barrier(cm));
        for (int r = 0; r < cm->world && n; ++r)
            RMHIP_HIP_CHECK(hipMemcpyAsync((char*)recv + (size_t)r * bytes + off, shm_slot(cm, r), n, hipMemcpyHostToDevice, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        RMHIP_TRY(shm_barrier(cm));
        if (bytes == 0) break;
    }
    return RMHIP_OK;
}

int shm_bcast(Context* c, Comm* cm, void* buf, size_t bytes, int root) {
    for (size_t off = 0; off < bytes || off == 0; off += kShmSlotBytes) {
        const size_t n = bytes - off < kShmSlotBytes ? bytes - off : kShmSlotBytes;
        if (cm->rank == root && n) RMHIP_HIP_CHECK(hipMemcpyAsync(shm_slot(cm, 0), (const char*)buf + off, n, hipMemcpyDeviceToHost, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        RMHIP_TRY(shm_barrier(cm));
        if (cm->rank != root && n) RMHIP_HIP_CHECK(hipMemcpyAsync((char*)buf + off, shm_slot(cm, 0), n, hipMemcpyHostToDevice, c->stream));
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        RMHIP_TRY(shm_barrier(cm));
        if (bytes == 0) break;
    }
    return RMHIP_OK;
}

// balanced contiguous split in units of `granule` (runmat_amd/sharding.py `partition`)
void partition(size_t total, int world, int rank, size_t granule, size_t* start, size_t* stop) {
    const size_t units = (total + granule - 1) / granule;
    const size_t base = units / (size_t)world, extra = units % (size_t)world;
    const size_t u0 = (size_t)rank * base + ((size_t)rank < extra ? (size_t)rank : extra);
    const size_t u1 = u0 + base + ((size_t)rank < extra ? 1 : 0);
    *start = u0 * granule < total ? u0 * granule : total;
    *stop = u1 * granule < total ? u1 * granule : total;
}

// the stream a collective runs on: the call stream, or - asynchronous form - the communication stream, ordered behind
// what the call stream has enqueued so far
int begin_collective(Context* c, Comm* cm, bool async, hipStream_t* s) {
    if (!async || cm->host) {
        // a broadcast posted on the communication stream may still be in flight: RCCL does not order two operations of one
        // communicator that sit on different streams, so the call stream waits for it first
        if (cm->pending) {
            RMHIP_HIP_CHECK(hipStreamWaitEvent(c->stream, cm->ev_done, 0));
            cm->pending = false;
        }
        *s = c->stream;
        return RMHIP_OK;
    }
    RMHIP_HIP_CHECK(hipEventRecord(cm->ev_ready, c->stream));
    RMHIP_HIP_CHECK(hipStreamWaitEvent(cm->stream, cm->ev_ready, 0));
    *s = cm->stream;
    return RMHIP_OK;
}
int end_collective(Context* c, Comm* cm, bool async) {
    (void)c;
    if (!async || cm->host) return RMHIP_OK;
    RMHIP_HIP_CHECK(hipEventRecord(cm->ev_done, cm->stream));
    cm->pending = true;
    return RMHIP_OK;
}

// synchronous collectives run on the call stream: first join a broadcast still in flight on the communication stream
int join_pending(Context* c, Comm* cm) {
    if (cm->pending && !cm->host) {
        RMHIP_HIP_CHECK(hipStreamWaitEvent(c->stream, cm->ev_done, 0));
        cm->pending = false;
    }
    return RMHIP_OK;
}

int require_comm(Context* c, Comm** out) {
    if (!c->comm) return fail(RMHIP_ERR_INVALID, "no communicator on this context: call rmhip_comm_init first");
    if (c->comm->aborted) return fail(RMHIP_ERR_HIP, "comm: this communicator was aborted (rmhip_comm_destroy, then a new rmhip_comm_init)");
    *out = c->comm;
    return RMHIP_OK;
}

}  // namespace

void comm_destroy(Context* c) {
    Comm* cm = c->comm;
    if (!cm) return;
    if (cm->nccl) (void)rccl().CommDestroy(cm->nccl);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    if (cm->ev_ready) (void)hipEventDestroy(cm->ev_ready);
    if (cm->ev_done) (void)hipEventDestroy(cm->ev_done);
    if (cm->hdr) {
        (void)munmap((void*)cm->hdr, cm->map_bytes);
        if (cm->rank == 0 && !cm->shm_name.empty()) (void)shm_unlink(cm->shm_name.c_str());  // only if init never got to its barrier
    }
    delete cm;
    c->comm = nullptr;
}

}  // namespace rmhip

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c)

extern "C" {

int rmhip_comm_unique_id(int transport, void* id_out) {
    if (!id_out) return fail(RMHIP_ERR_INVALID, "null id");
    std::memset(id_out, 0, RMHIP_COMM_ID_BYTES);
    if (transport == RMHIP_COMM_HOST_SHM) {
        static std::atomic<unsigned> counter{0};
        char* p = (char*)id_out;
        std::memcpy(p, kShmMagic, 8);
        const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
        std::snprintf(p + 8, RMHIP_COMM_ID_BYTES - 8, "/rmhip-%ld-%llx-%u", (long)getpid(), (unsigned long long)now, counter++);
        return RMHIP_OK;
    }
    if (transport != RMHIP_COMM_RCCL) return fail(RMHIP_ERR_INVALID, "unknown transport %d", transport);
    if (!rccl().ok) return fail(RMHIP_ERR_UNSUPPORTED, "librccl could not be loaded (%s)", rccl().load_error.c_str());
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(id) == RMHIP_COMM_ID_BYTES, "RMHIP_COMM_ID_BYTES is ncclUniqueId's size");
    std::memcpy(id_out, &id, sizeof id);
    return RMHIP_OK;
}

int rmhip_comm_init(rmhip_ctx* ctx, const void* unique_id, int rank, int world) {
    CTX_OR_FAIL(ctx);
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(RMHIP_ERR_INVALID, "comm_init: bad id / rank %d / world %d", rank, world);
    if (c->comm) return fail(RMHIP_ERR_INVALID, "comm_init: this context already has a communicator");
    Comm* cm = new Comm();
    cm->rank = rank;
    cm->world = world;
    c->comm = cm;
    if (std::memcmp(unique_id, kShmMagic, 8) == 0) {
        cm->host = true;
        cm->shm_name = (const char*)unique_id + 8;
        cm->map_bytes = sizeof(ShmHeader) + (size_t)world * kShmSlotBytes;
        int fd = -1;
        if (rank == 0) {
            fd = shm_open(cm->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)cm->map_bytes) != 0) {
                if (fd >= 0) close(fd);
                comm_destroy(c);
                return fail(RMHIP_ERR_HIP, "comm_init: cannot create the shared segment (%s)", std::strerror(errno));
            }
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            struct stat st;
            for (;;) {  // wait until rank 0 created AND sized the segment
                fd = shm_open(cm->shm_name.c_str(), O_RDWR, 0600);
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= cm->map_bytes) break;
                if (fd >= 0) close(fd);
                fd = -1;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                    comm_destroy(c);
                    return fail(RMHIP_ERR_HIP, "comm_init: rank 0 did not create the shared segment within 60 s");
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        void* m = mmap(nullptr, cm->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            comm_destroy(c);
            return fail(RMHIP_ERR_HIP, "comm_init: mmap failed (%s)", std::strerror(errno));
        }
        cm->hdr = (ShmHeader*)m;
        cm->slots = (char*)m + sizeof(ShmHeader);
        if (rank == 0) {  // a fresh segment is zero-filled: counters start at 0
            cm->hdr->world = (uint32_t)world;
            cm->hdr->ready.store(1, std::memory_order_release);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            while (cm->hdr->ready.load(std::memory_order_acquire) == 0) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                    comm_destroy(c);
                    return fail(RMHIP_ERR_HIP, "comm_init: shared segment never became ready");
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (cm->hdr->world != (uint32_t)world) {
                comm_destroy(c);
                return fail(RMHIP_ERR_INVALID, "comm_init: world size mismatch between ranks");
            }
        }
        cm->hdr->attached.fetch_add(1);
        const int brc = shm_barrier(cm);
        // every rank has the segment mapped now (or the barrier timed out): the NAME can go - a rank that crashes later, or a
        // communicator nobody destroys, no longer leaves world x 8 MiB behind in /dev/shm
        if (rank == 0) {
            (void)shm_unlink(cm->shm_name.c_str());
            cm->shm_name.clear();
        }
        if (brc != RMHIP_OK) comm_destroy(c);
        return brc;
    }
    if (!rccl().ok) {
        comm_destroy(c);
        return fail(RMHIP_ERR_UNSUPPORTED, "librccl could not be loaded (%s)", rccl().load_error.c_str());
    }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    {
        // RCCL checks hipGetLastError() at places and reports whatever non-sticky error an EARLIER, unrelated call left
        // behind (an attribute query that is allowed to fail, ...) as its own "unhandled cuda error": start clean
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess) RMHIP_TRACEF("comm_init: cleared a stale HIP error (%s)", hipGetErrorString(stale));
    }
    // ncclCommInitRank returns when EVERY rank has joined: a peer that failed before it got here (librccl missing there, a device
    // error, a crashed process) would leave this rank inside the call for good.  The call runs on a helper thread and is waited for
    // with a bound (RMHIP_COMM_INIT_TIMEOUT_S, default 60): past it this rank gives up - the caller's ranks then agree on a fall-back
    // through their control plane (runmat_amd/sharding.py try_native_comm) - and the helper, if the call ever returns, destroys the
    // communicator it got.  RMHIP_COMM_TEST_FAIL_RANK=r makes rank r fail here at once (test hook for exactly that situation).
    if (const char* tf = std::getenv("RMHIP_COMM_TEST_FAIL_RANK"))
        if (std::atoi(tf) == rank) {
            comm_destroy(c);
            return fail(RMHIP_ERR_HIP, "comm_init: forced failure on rank %d (RMHIP_COMM_TEST_FAIL_RANK)", rank);
        }
    struct InitState {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false, abandoned = false;
        ncclComm_t comm = nullptr;
        ncclResult_t result = ncclSuccess;
    };
    auto st = std::make_shared<InitState>();
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::thread([st, id, world, rank, dev]() {
        (void)hipSetDevice(dev);
        ncclComm_t comm = nullptr;
        const ncclResult_t r = rccl().CommInitRank(&comm, world, id, rank);
        std::unique_lock<std::mutex> lk(st->mu);
        st->result = r;
        st->comm = comm;
        st->done = true;
        if (st->abandoned && r == ncclSuccess && comm) {  // nobody is waiting any more
            lk.unlock();
            (void)rccl().CommDestroy(comm);
            return;
        }
        st->cv.notify_all();
    }).detach();
    double timeout_s = 60.0;
    if (const char* v = std::getenv("RMHIP_COMM_INIT_TIMEOUT_S")) timeout_s = std::atof(v) > 0 ? std::atof(v) : timeout_s;
    ncclResult_t r = ncclSuccess;
    {
        std::unique_lock<std::mutex> lk(st->mu);
        if (!st->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return st->done; })) {
            st->abandoned = true;
            lk.unlock();
            comm_destroy(c);
            return fail(RMHIP_ERR_HIP, "comm_init: ncclCommInitRank did not return within %.0f s (a peer never joined?)", timeout_s);
        }
        r = st->result;
        cm->nccl = st->comm;
    }
    if (r != ncclSuccess) {
        cm->nccl = nullptr;
        comm_destroy(c);
        return nccl_fail("ncclCommInitRank", r);
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e = hipStreamCreateWithPriority(&cm->stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        comm_destroy(c);
        return fail(RMHIP_ERR_HIP, "comm_init: %s", hipGetErrorString(e));
    }
    return RMHIP_OK;
}

int rmhip_comm_destroy(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    if (c->comm && c->comm->pending) (void)hipStreamSynchronize(c->comm->stream);
    (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    return RMHIP_OK;
}

int rmhip_comm_abort(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    Comm* cm = c->comm;
    if (!cm || cm->aborted) return RMHIP_OK;
    cm->aborted = true;
    if (cm->host && cm->hdr) cm->hdr->aborted.store(1, std::memory_order_release);
    if (cm->nccl && rccl().CommAbort) {  // frees the communicator: nothing left for rmhip_comm_destroy to destroy
        (void)rccl().CommAbort(cm->nccl);
        cm->nccl = nullptr;
    }
    cm->pending = false;
    return RMHIP_OK;
}

int rmhip_comm_wait_bounded(rmhip_ctx* ctx, double timeout_s) {
    CTX_OR_FAIL(ctx);
    Comm* cm = c->comm;
    if (!cm || cm->aborted || cm->world <= 1) {
        RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
        return RMHIP_OK;
    }
    RMHIP_TRY(join_pending(c, cm));
    if (!(timeout_s > 0.0)) timeout_s = std::getenv("RMHIP_COMM_TIMEOUT_S") ? std::atof(std::getenv("RMHIP_COMM_TIMEOUT_S")) : 300.0;
    const bool test_expire = std::getenv("RMHIP_COMM_TEST_EXPIRE") != nullptr;  // test hook: behave as if the wait had expired
    hipEvent_t ev = nullptr;
    RMHIP_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, c->stream);
    const auto t0 = std::chrono::steady_clock::now();
    bool expired = false, comm_error = false;
    while (e == hipSuccess) {
        const hipError_t q = test_expire ? hipErrorNotReady : hipEventQuery(ev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            e = q;
            break;
        }
        if (cm->nccl && rccl().CommGetAsyncError) {
            ncclResult_t ar = ncclSuccess;
            if (rccl().CommGetAsyncError(cm->nccl, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) comm_error = true;
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (comm_error || waited > timeout_s || test_expire) {
            expired = true;
            break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.01 ? 20 : 500));
    }
    if (expired) {
        // a peer left (or the fabric failed): abort locally - the collective kernels of this rank exit - and drain the stream
        cm->aborted = true;
        if (cm->host && cm->hdr) cm->hdr->aborted.store(1, std::memory_order_release);
        if (cm->nccl && rccl().CommAbort) {
            (void)rccl().CommAbort(cm->nccl);
            cm->nccl = nullptr;
        }
        cm->pending = false;
        (void)hipStreamSynchronize(c->stream);
        (void)hipEventDestroy(ev);
        return fail(RMHIP_ERR_HIP, comm_error ? "comm: the communicator reported an asynchronous error; aborted"
                                               : "comm: timed out after %.0f s waiting for a collective (a peer left?); aborted", timeout_s);
    }
    (void)hipEventDestroy(ev);
    if (e != hipSuccess) return fail(RMHIP_ERR_HIP, "comm: %s", hipGetErrorString(e));
    return RMHIP_OK;
}

int rmhip_comm_rank(rmhip_ctx* ctx, int* rank, int* world) {
    CTX_OR_FAIL(ctx);
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (world) *world = c->comm ? c->comm->world : 1;
    return RMHIP_OK;
}

int rmhip_comm_wait(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    Comm* cm;
    if (c->comm && c->comm->aborted) return RMHIP_OK;  // nothing of it is in flight any more
    RMHIP_TRY(require_comm(c, &cm));
    if (cm->pending) {
        RMHIP_HIP_CHECK(hipStreamWaitEvent(c->stream, cm->ev_done, 0));
        cm->pending = false;
    }
    return RMHIP_OK;
}

int rmhip_comm_bcast(rmhip_ctx* ctx, const rmhip_view_t* v, int root, int async) {
    CTX_OR_FAIL(ctx);
    Comm* cm;
    RMHIP_TRY(require_comm(c, &cm));
    if (!v) return fail(RMHIP_ERR_INVALID, "null view");
    if (root < 0 || root >= cm->world) return fail(RMHIP_ERR_INVALID, "bcast: root %d of %d", root, cm->world);
    Buffer b;
    RMHIP_TRY(c->get_raw(v->buf, &b));
    if (b.dtype != DT_F64 || b.lazy()) return fail(RMHIP_ERR_UNSUPPORTED, "bcast: plain f64 buffers only");
    std::vector<size_t> s = b.shape;
    if (s.empty()) s = {1, 1};
    if (s.size() == 1) s.push_back(1);
    size_t cols_all = 1;
    for (size_t i = 1; i < s.size(); ++i) cols_all *= s[i];
    if (v->row_off + v->rows > s[0] || v->col_off + v->cols > cols_all)
        return fail(RMHIP_ERR_SHAPE, "bcast: view [%zu+%zu, %zu+%zu] exceeds buffer %zux%zu", v->row_off, v->rows, v->col_off, v->cols, s[0], cols_all);
    if ((cm->world == 1 && cm->host) || v->rows * v->cols == 0) return RMHIP_OK;  // a one-rank RCCL communicator still runs the collective
    const size_t ld = s[0], count = v->rows * v->cols;
    double* base = b.data() + v->row_off + v->col_off * ld;
    const bool dense = v->rows == ld;  // whole columns: the view is contiguous
    hipStream_t st;
    RMHIP_TRY(begin_collective(c, cm, async != 0, &st));
    std::shared_ptr<Allocation> pack;
    double* wire = base;
    if (!dense) {  // pack the sub-block, send, unpack on the receivers
        RMHIP_TRY(c->alloc_device(count, &pack));
        wire = pack->ptr;
        if (cm->rank == root)
            RMHIP_HIP_CHECK(hipMemcpy2DAsync(wire, v->rows * sizeof(double), base, ld * sizeof(double), v->rows * sizeof(double), v->cols,
                                             hipMemcpyDeviceToDevice, st));
    }
    if (cm->host) {
        RMHIP_TRY(shm_bcast(c, cm, wire, count * sizeof(double), root));
    } else {
        const ncclResult_t r = rccl().Broadcast(wire, wire, count, ncclFloat64, root, cm->nccl, st);
        if (r != ncclSuccess) return nccl_fail("ncclBroadcast", r);
    }
    if (!dense && cm->rank != root)
        RMHIP_HIP_CHECK(hipMemcpy2DAsync(base, ld * sizeof(double), wire, v->rows * sizeof(double), v->rows * sizeof(double), v->cols,
                                         hipMemcpyDeviceToDevice, st));
    if (pack && st != c->stream) {
        // the staging block goes back to the pool when this call returns, and the pool hands blocks out in call-stream
        // order: the call stream must not reuse it before the communication stream is done with it
        RMHIP_TRY(end_collective(c, cm, true));
        RMHIP_HIP_CHECK(hipStreamWaitEvent(c->stream, cm->ev_done, 0));
        cm->pending = false;
        return RMHIP_OK;
    }
    return end_collective(c, cm, async != 0);
}

int rmhip_comm_allgather_f64(rmhip_ctx* ctx, rmhip_buf local, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Comm* cm;
    RMHIP_TRY(require_comm(c, &cm));
    RMHIP_TRY(join_pending(c, cm));
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    Buffer b;
    RMHIP_TRY(c->get(local, &b));  // f64, plain layout (a precision-32 buffer is widened: the exchange is f64)
    const size_t k = b.numel;
    const size_t oshape[2] = {k, (size_t)cm->world};
    Buffer ob;
    uint64_t oid = 0;
    {  // the gathered values stay f64 whatever the provider precision (they are partial sums on their way to the host)
        Buffer nb;
        nb.shape.assign(oshape, oshape + 2);
        nb.numel = k * (size_t)cm->world;
        RMHIP_TRY(c->alloc_device(nb.numel ? nb.numel : 1, &nb.alloc));
        ob = nb;
        RMHIP_TRY(c->register_buffer(std::move(nb), &oid));
    }
    int rc = RMHIP_OK;
    if (k) {
        if (cm->world == 1 && cm->host) {
            hipError_t e = hipMemcpyAsync(ob.data(), b.data(), k * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) rc = fail(RMHIP_ERR_HIP, "allgather_f64: %s", hipGetErrorString(e));
        } else if (cm->host) {
            rc = shm_allgather(c, cm, b.data(), ob.data(), k * sizeof(double));
        } else {
            const ncclResult_t r = rccl().AllGather(b.data(), ob.data(), k, ncclFloat64, cm->nccl, c->stream);
            if (r != ncclSuccess) rc = nccl_fail("ncclAllGather", r);
        }
    }
    if (rc) {
        rmhip_free(ctx, oid);
        return rc;
    }
    *out = oid;
    return RMHIP_OK;
}

int rmhip_comm_allgather_rows(rmhip_ctx* ctx, rmhip_buf local, size_t rows_total, size_t granule, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    Comm* cm;
    RMHIP_TRY(require_comm(c, &cm));
    RMHIP_TRY(join_pending(c, cm));
    if (!out || granule == 0) return fail(RMHIP_ERR_INVALID, "allgather_rows: bad argument");
    Buffer b;
    RMHIP_TRY(c->get_raw(local, &b));
    if (b.dtype != DT_F64) return fail(RMHIP_ERR_UNSUPPORTED, "allgather_rows: needs a precision-64 provider");
    RMHIP_TRY(c->get(local, &b));
    if (b.shape.size() != 2) return fail(RMHIP_ERR_UNSUPPORTED, "allgather_rows: 2-D blocks only");
    const size_t rows_g = b.shape[0], n = b.shape[1];
    size_t r0, r1;
    partition(rows_total, cm->world, cm->rank, granule, &r0, &r1);
    if (rows_g != r1 - r0) return fail(RMHIP_ERR_SHAPE, "allgather_rows: rank %d owns rows [%zu, %zu) but the block has %zu rows", cm->rank, r0, r1, rows_g);
    size_t width = 0;
    for (int r = 0; r < cm->world; ++r) {
        size_t a0, a1;
        partition(rows_total, cm->world, r, granule, &a0, &a1);
        width = a1 - a0 > width ? a1 - a0 : width;
    }
    const size_t oshape[2] = {rows_total, n};
    Buffer ob;
    uint64_t oid = 0;
    RMHIP_TRY(c->new_buffer(oshape, 2, &oid, &ob));
    auto bail = [&](int rc) {
        rmhip_free(ctx, oid);
        return rc;
    };
    if (rows_total * n == 0) {
        *out = oid;
        return RMHIP_OK;
    }
    // every rank sends `width` x n doubles (ragged blocks padded), the blocks land in rank order in a staging buffer and
    // are then placed: rank r's block is rows [a0, a1) of every column of the column-major result
    std::shared_ptr<Allocation> send, stage;
    const double* wire = b.data();
    if (rows_g != width) {
        if (int rc = c->alloc_device(width * n, &send)) return bail(rc);
        hipError_t e = hipMemsetAsync(send->ptr, 0, width * n * sizeof(double), c->stream);
        if (e == hipSuccess && rows_g)
            e = hipMemcpy2DAsync(send->ptr, width * sizeof(double), b.data(), rows_g * sizeof(double), rows_g * sizeof(double), n,
                                 hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) return bail(fail(RMHIP_ERR_HIP, "allgather_rows: %s", hipGetErrorString(e)));
        wire = send->ptr;
    }
    if (int rc = c->alloc_device((size_t)cm->world * width * n, &stage)) return bail(rc);
    if (cm->world == 1 && cm->host) {
        hipError_t e = hipMemcpyAsync(stage->ptr, wire, width * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) return bail(fail(RMHIP_ERR_HIP, "allgather_rows: %s", hipGetErrorString(e)));
    } else if (cm->host) {
        if (int rc = shm_allgather(c, cm, wire, stage->ptr, width * n * sizeof(double))) return bail(rc);
    } else {
        const ncclResult_t r = rccl().AllGather(wire, stage->ptr, width * n, ncclFloat64, cm->nccl, c->stream);
        if (r != ncclSuccess) return bail(nccl_fail("ncclAllGather", r));
    }
    for (int r = 0; r < cm->world; ++r) {
        size_t a0, a1;
        partition(rows_total, cm->world, r, granule, &a0, &a1);
        if (a1 == a0) continue;
        hipError_t e = hipMemcpy2DAsync(ob.data() + a0, rows_total * sizeof(double), stage->ptr + (size_t)r * width * n, width * sizeof(double),
                                        (a1 - a0) * sizeof(double), n, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) return bail(fail(RMHIP_ERR_HIP, "allgather_rows: %s", hipGetErrorString(e)));
    }
    *out = oid;
    return RMHIP_OK;
}

int rmhip_comm_barrier(rmhip_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    Comm* cm;
    RMHIP_TRY(require_comm(c, &cm));
    RMHIP_TRY(join_pending(c, cm));
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (cm->world == 1) return RMHIP_OK;
    if (cm->host) return shm_barrier(cm);
    // RCCL: a one-element all-gather is the barrier
    std::shared_ptr<Allocation> tmp;
    RMHIP_TRY(c->alloc_device((size_t)cm->world + 1, &tmp));
    const ncclResult_t r = rccl().AllGather(tmp->ptr + cm->world, tmp->ptr, 1, ncclFloat64, cm->nccl, c->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
    RMHIP_HIP_CHECK(hipStreamSynchronize(c->stream));
    return RMHIP_OK;
}

}  // extern "C"
