// sharded.cpp -- the sharded forms of the hot path behind the C ABI, for hosts that are not Python:
//   rmhip_matmul_row_sharded        C[rows_g, :] = A[rows_g, :] * B, optionally followed by the row-block all-gather
//   rmhip_mldivide_row_partitioned  x = A \ b with [A | b] distributed by row blocks (BASELINE.json configs[4])
//   rmhip_blk_absmax                max |a_ij| over a view (the multiplier guard)
// The reference has no multi-device code: nothing here replaces a trait method.  One process per GPU, one
// context per process, a communicator attached with rmhip_comm_init (RCCL over xGMI, or host shared memory for tests); every
// rank calls the same entry point with its local block.  The drivers are written over the library's own block-level entry
// points (rmhip_blk_* on sub-blocks, rmhip_comm_bcast / allgather) - the same sequence runmat_amd/sharding.py issues from
// Python, with which tests/test_gpu_multirank.py compares them bit for bit - plus a depth-1 look-ahead in the solver.
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "common.h"

using namespace rmhip;

#define CTX_OR_FAIL(ctx)                                            \
    if (!(ctx)) return fail(RMHIP_ERR_INVALID, "null context");     \
    Context* c = context_of(ctx);                                   \
    std::lock_guard<std::recursive_mutex> _call(c->call_mu);        \
    DeviceGuard _dg(c);                                             \
    NarrowScope _ns(c)

namespace {

rmhip_view_t view(rmhip_buf buf, size_t r0, size_t c0, size_t rows, size_t cols) { return rmhip_view_t{buf, r0, c0, rows, cols}; }

// frees every buffer it was handed when it goes out of scope (error paths included)
struct Temps {
    rmhip_ctx* ctx;
    std::vector<rmhip_buf> ids;
    explicit Temps(rmhip_ctx* x) : ctx(x) {}
    rmhip_buf keep(rmhip_buf id) {
        ids.push_back(id);
        return id;
    }
    void drop(rmhip_buf id) {
        for (auto& v : ids)
            if (v == id) {
                (void)rmhip_free(ctx, id);
                v = 0;
                return;
            }
    }
    ~Temps() {
        // an early way out may leave an asynchronous broadcast in flight on the communication stream whose target is one of these
        // tiles: the context's stream waits for it before any of them goes back to the pool
        int r = 0, w = 1;
        if (rmhip_comm_rank(ctx, &r, &w) == RMHIP_OK && w > 1) (void)rmhip_comm_wait(ctx);
        for (rmhip_buf id : ids)
            if (id) (void)rmhip_free(ctx, id);
    }
};

// max |a_ij| over a view as a DEVICE scalar (1 x 1 tensor): the per-panel multiplier guard no longer reads it back
int absmax_dev(rmhip_ctx* ctx, const rmhip_view_t* v, rmhip_buf* out) {
    rmhip_buf blk = 0, ab = 0;
    RMHIP_TRY(rmhip_blk_copy(ctx, v, &blk));
    int rc = rmhip_unary(ctx, RMHIP_ABS, blk, &ab);
    (void)rmhip_free(ctx, blk);
    if (rc != RMHIP_OK) return rc;
    rc = rmhip_reduce(ctx, RMHIP_RMAX, ab, -1, /*include NaN: a NaN multiplier must fail the guard*/ 0, out);
    (void)rmhip_free(ctx, ab);
    return rc;
}

// Per-phase device time of one row-partitioned solve (rmhip_rp_phase_ms): pairs of timed events around the phases' launches on the stream
// they run on, summed after the solve.  panel = the owner's factorisation, interchanges, U12 and tile copy; wait = the stream idling for
// a tile broadcast (rmhip_comm_wait); update = multipliers and trailing updates; exchange = the guard's gather and the replicated tail.
struct PhaseTimers {
    enum { PANEL = 0, WAIT, UPDATE, EXCHANGE, NPHASE };
    struct Span {
        int phase;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    bool on = true;
    hipEvent_t begin(hipStream_t st) {
        if (!on) return nullptr;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        (void)hipEventRecord(e, st);
        return e;
    }
    void end(int phase, hipEvent_t a, hipStream_t st) {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            return;
        }
        (void)hipEventRecord(b, st);
        spans.push_back(Span{phase, a, b});
    }
    void collect(double out[NPHASE]) {
        for (int i = 0; i < NPHASE; ++i) out[i] = 0.0;
        for (Span& sp : spans) {
            float ms = 0.f;
            if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) out[sp.phase] += ms;
            (void)hipEventDestroy(sp.a);
            (void)hipEventDestroy(sp.b);
        }
        spans.clear();
    }
    ~PhaseTimers() {
        for (Span& sp : spans) {
            (void)hipEventDestroy(sp.a);
            (void)hipEventDestroy(sp.b);
        }
    }
};

int zeros(rmhip_ctx* ctx, size_t rows, size_t cols, rmhip_buf* out) {
    const size_t shape[2] = {rows, cols};
    return rmhip_fill(ctx, 0.0, shape, 2, out);
}

}  // namespace

extern "C" {

int rmhip_blk_absmax(rmhip_ctx* ctx, const rmhip_view_t* v, double* out) {
    CTX_OR_FAIL(ctx);
    if (!v || !out) return fail(RMHIP_ERR_INVALID, "blk_absmax: null argument");
    *out = 0.0;
    if (v->rows * v->cols == 0) return RMHIP_OK;
    rmhip_buf blk = 0, ab = 0, mx = 0;
    Temps t(ctx);
    RMHIP_TRY(rmhip_blk_copy(ctx, v, &blk));
    t.keep(blk);
    RMHIP_TRY(rmhip_unary(ctx, RMHIP_ABS, blk, &ab));
    t.keep(ab);
    RMHIP_TRY(rmhip_reduce(ctx, RMHIP_RMAX, ab, -1, /*include NaN: a NaN multiplier must fail the guard*/ 0, &mx));
    t.keep(mx);
    return rmhip_read_scalar(ctx, mx, 0, out);
}

int rmhip_matmul_row_sharded(rmhip_ctx* ctx, rmhip_buf a_rows, rmhip_buf b, size_t rows_total, size_t granule, int gather, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    rmhip_buf local = 0;
    RMHIP_TRY(rmhip_matmul(ctx, a_rows, b, &local));  // the embarrassingly parallel part: no exchange
    int rank = 0, world = 1;
    RMHIP_TRY(rmhip_comm_rank(ctx, &rank, &world));
    if (!gather || world == 1) {
        *out = local;
        return RMHIP_OK;
    }
    const int rc = rmhip_comm_allgather_rows(ctx, local, rows_total, granule ? granule : 128, out);
    (void)rmhip_free(ctx, local);
    return rc;
}

int rmhip_rp_phase_ms(rmhip_ctx* ctx, double* out4) {
    CTX_OR_FAIL(ctx);
    if (!out4) return fail(RMHIP_ERR_INVALID, "null out");
    for (int i = 0; i < 4; ++i) out4[i] = c->rp_phase_ms[i];
    return RMHIP_OK;
}

// Row block q (height rb) of the n x (n + nrhs) augmented matrix lives on rank q % world, blocks in ownership order in `ab_local`
// (overwritten with this rank's rows of the factors).  Algorithm, guard and failure modes: include/rmhip.h.
int rmhip_mldivide_row_partitioned(rmhip_ctx* ctx, rmhip_buf ab_local, size_t n, size_t nrhs, size_t rb, double tau, rmhip_buf* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return fail(RMHIP_ERR_INVALID, "null out");
    if (c->precision != 64) return fail(RMHIP_ERR_UNSUPPORTED, "mldivide_row_partitioned: needs a precision-64 provider (block views update f64 storage in place)");
    if (n == 0 || nrhs == 0 || rb == 0) return fail(RMHIP_ERR_INVALID, "mldivide_row_partitioned: empty system");
    int rank = 0, world = 1;
    RMHIP_TRY(rmhip_comm_rank(ctx, &rank, &world));
    const size_t ncols = n + nrhs, nblocks = (n + rb - 1) / rb;
    std::vector<size_t> mine;  // owned row blocks, ascending = local storage order
    for (size_t q = 0; q < nblocks; ++q)
        if ((int)(q % (size_t)world) == rank) mine.push_back(q);
    size_t nloc = 0;
    for (size_t q : mine) nloc += std::min(rb, n - q * rb);
    {
        size_t shape[8], r = 8;
        RMHIP_TRY(rmhip_shape(ctx, ab_local, &r, shape));
        if (r != 2 || shape[0] != nloc || shape[1] != ncols)
            return fail(RMHIP_ERR_SHAPE, "mldivide_row_partitioned: rank %d of %d owns %zu rows of the %zu x %zu augmented matrix, the local block is not %zu x %zu",
                        rank, world, nloc, n, ncols, nloc, ncols);
    }
    auto local_row_offset = [&](size_t q) { return (q / (size_t)world) * rb; };
    auto first_local_row_at_or_after = [&](size_t q) {
        for (size_t b : mine)
            if (b >= q) return local_row_offset(b);
        return nloc;
    };
    // the panels are factored with the solve path's kernels (RMHIP_RP_SOLVE_PATH=0: the grid-wide panel kernels of `lu`); the flag goes
    // back on every way out
    struct SolvePathScope {
        Context* c;
        bool saved;
        ~SolvePathScope() { c->blk_lu_solve_path = saved; }
    } sp_scope{c, c->blk_lu_solve_path};
    {
        const char* v = std::getenv("RMHIP_RP_SOLVE_PATH");
        c->blk_lu_solve_path = !(v && v[0] == '0');
    }
    const size_t n_direct = nblocks > (size_t)world ? nblocks - (size_t)world : 0;  // panels whose owner still has a block below the tile
    const bool overlap = world > 1;  // asynchronous broadcasts on the communication stream (look-ahead)
    // Any way out that the ranks did not agree on (a local allocation or device failure between two collectives) aborts the
    // communicator, so that the peers fail in their next barrier instead of waiting for a broadcast that never comes; declared before
    // `temps`, whose destructor (tiles that may still be the target of a broadcast in flight) runs first.
    struct AbortUnlessAgreed {
        rmhip_ctx* ctx;
        int world;
        bool agreed = false;
        ~AbortUnlessAgreed() {
            if (!agreed && world > 1) (void)rmhip_comm_abort(ctx);
        }
    } exit_guard{ctx, world};
    if (const char* v = std::getenv("RMHIP_RP_TEST_FAIL_RANK"))  // test hook: this rank leaves before its first collective
        if (world > 1 && std::atoi(v) == rank) return fail(RMHIP_ERR_HIP, "mldivide_row_partitioned: injected local failure on rank %d", rank);
    Temps temps(ctx);
    // A failure on one rank (singular pivot inside its domain, an allocation) must not leave the others blocked in the panel
    // broadcast: the failing rank poisons what it sends with NaN and keeps taking part in every collective; the NaN reaches every
    // rank's multiplier guard, so all of them leave together after the one exchange at the end.
    bool failed = false;
    std::string why;
    auto poison = [&](rmhip_buf tile, size_t rows, size_t cols) {
        rmhip_buf nanb = 0;
        const size_t shape[2] = {rows, cols};
        if (rmhip_fill(ctx, std::numeric_limits<double>::quiet_NaN(), shape, 2, &nanb) != RMHIP_OK) return;
        const rmhip_view_t dst = view(tile, 0, 0, rows, cols);
        (void)rmhip_blk_assign(ctx, &dst, nanb);
        (void)rmhip_free(ctx, nanb);
    };
    // the largest multiplier outside the diagonal domains, accumulated ON THE DEVICE (elementwise max keeps a NaN): one read at the guard
    rmhip_buf growth_dev = 0;
    {
        const size_t one[2] = {1, 1};
        RMHIP_TRY(rmhip_fill(ctx, 0.0, one, 2, &growth_dev));
    }
    struct GrowthScope {  // (replaced as it accumulates: not in `temps`)
        rmhip_ctx* ctx;
        rmhip_buf* id;
        ~GrowthScope() {
            if (*id) (void)rmhip_free(ctx, *id);
        }
    } growth_scope{ctx, &growth_dev};
    const bool deferred_on = !(std::getenv("RMHIP_RP_DEFERRED") && std::getenv("RMHIP_RP_DEFERRED")[0] == '0');
    PhaseTimers timers;
    timers.on = !(std::getenv("RMHIP_RP_TIMERS") && std::getenv("RMHIP_RP_TIMERS")[0] == '0');
    for (double& v : c->rp_phase_ms) v = 0.0;
    // second stream: the trailing update of panel p beyond panel p + 1's columns runs there while this stream factors panel p + 1
    // (RMHIP_RP_OVERLAP=0: everything on the context's stream, as in round 5)
    const bool side_on = !(std::getenv("RMHIP_RP_OVERLAP") && std::getenv("RMHIP_RP_OVERLAP")[0] == '0');
    hipStream_t const main_stream = c->stream;
    if (side_on && !c->lu_side_stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        RMHIP_HIP_CHECK(hipStreamCreateWithPriority(&c->lu_side_stream, hipStreamNonBlocking, lo));
    }
    hipStream_t const side = side_on ? c->lu_side_stream : main_stream;
    const size_t side_pad = std::getenv("RMHIP_RP_SIDE_PAD") ? (size_t)std::atol(std::getenv("RMHIP_RP_SIDE_PAD")) : (size_t)(84 * 1024 - 73728);
    struct StreamRestore {  // whatever happens, the context's stream comes back and the side stream is drained
        Context* c;
        hipStream_t main_stream, side;
        ~StreamRestore() {
            c->stream = main_stream;
            if (side != main_stream) (void)hipStreamSynchronize(side);
        }
    } stream_restore{c, main_stream, side};
    hipEvent_t ev_side_done = nullptr, ev_main_ready = nullptr;  // (reused: a wait enqueued earlier has captured the earlier record)
    if (side != main_stream) {
        RMHIP_HIP_CHECK(hipEventCreateWithFlags(&ev_side_done, hipEventDisableTiming));
        RMHIP_HIP_CHECK(hipEventCreateWithFlags(&ev_main_ready, hipEventDisableTiming));
    }
    struct EventScope {
        hipEvent_t *a, *b;
        ~EventScope() {
            if (*a) (void)hipEventDestroy(*a);
            if (*b) (void)hipEventDestroy(*b);
        }
    } event_scope{&ev_side_done, &ev_main_ready};
    // cooperative yield, as in the single-GPU two-level driver: the panels' k_rp_top counts itself into a table and the side stream's
    // eight-wave block on its CU sleeps meanwhile (RMHIP_RP_YIELD=1 enables)
    std::shared_ptr<Allocation> yield_tab;
    struct YieldScope {
        Context* c;
        ~YieldScope() {
            c->ext_yield_tab = nullptr;
            c->gemm_yield_word = nullptr;
        }
    } yield_scope{c};
    if (side != main_stream && std::getenv("RMHIP_RP_YIELD") && std::getenv("RMHIP_RP_YIELD")[0] == '1') {  // (measured neutral: 95.7-96.1 with, 96.1-97.3 without - off by default)
        RMHIP_TRY(c->alloc_device(kYieldSlots / 2, &yield_tab));
        RMHIP_HIP_CHECK(hipMemsetAsync(yield_tab->ptr, 0, sizeof(unsigned) * kYieldSlots, main_stream));
        c->ext_yield_tab = (unsigned*)yield_tab->ptr;
        c->gemm_yield_word = c->ext_yield_tab;
    }
    bool side_pending = false;  // the side stream holds an update the main stream has not waited for yet
    auto join_side = [&]() {
        if (side_pending) (void)hipStreamWaitEvent(main_stream, ev_side_done, 0);
        side_pending = false;
    };
    struct Tile {
        size_t j, w;
        rmhip_buf id;
    };
    std::vector<Tile> tiles;  // every direct panel's tile row [w x (ncols - j)], kept for the back substitution

    // the owner's share of panel p: factor among its own rows from the tile down, interchanges, [pending work on the interchanged tile
    // rows], U12 and the y part; returns the tile row.  `before_trsm` runs between the interchanges and the triangular solve: the
    // look-ahead applies the previous panel's update to the (now final) tile rows there - rows exchanged above carry their
    // multipliers with them (the left part is swapped too), so update-after-swap equals swap-after-update.
    auto factor_panel = [&](size_t p, rmhip_buf* tile_out, const std::function<int()>& before_trsm) -> int {
        const size_t j = p * rb, w = rb, width = ncols - j;
        const size_t lr = local_row_offset(p);
        rmhip_buf ipiv = 0;
        int info = 0;
        const rmhip_view_t pan = view(ab_local, lr, j, nloc - lr, w);
        hipEvent_t t_panel = timers.begin(main_stream);
        struct PanelSpan {
            PhaseTimers& t;
            hipEvent_t a;
            hipStream_t st;
            ~PanelSpan() { t.end(PhaseTimers::PANEL, a, st); }
        } panel_span{timers, t_panel, main_stream};
        // (round 6) the panel without any host round trip when the solve path's kernels are in use: its status - a singular pivot, its
        // largest multiplier - is folded into this rank's guard value on the device and seen by every rank at the one exchange
        int rc = c->blk_lu_solve_path && deferred_on ? rmhip_blk_lu_deferred(ctx, &pan, growth_dev, &ipiv) : rmhip_blk_lu(ctx, &pan, &ipiv, &info);
        // the interchanges move rows the side stream's update of the previous panel reads (its multipliers, left of the panel) and
        // writes (right of it): that update has to be through first - it ran under the factorisation above
        join_side();
        if (rc == RMHIP_OK && info > 0) {
            (void)rmhip_free(ctx, ipiv);
            return fail(RMHIP_ERR_GROWTH, "panel %zu: %d pivot(s) at the singular cut-off inside the diagonal domain", p, info);
        }
        if (rc != RMHIP_OK) return rc;
        if (j > 0) {
            const rmhip_view_t left = view(ab_local, lr, 0, nloc - lr, j);
            rc = rmhip_blk_swap_rows(ctx, &left, ipiv);
        }
        if (rc == RMHIP_OK) {
            const rmhip_view_t right = view(ab_local, lr, j + w, nloc - lr, width - w);
            rc = rmhip_blk_swap_rows(ctx, &right, ipiv);
        }
        (void)rmhip_free(ctx, ipiv);
        if (rc != RMHIP_OK) return rc;
        if (before_trsm) RMHIP_TRY(before_trsm());
        const rmhip_view_t t11 = view(ab_local, lr, j, w, w), a12 = view(ab_local, lr, j + w, w, width - w);
        RMHIP_TRY(rmhip_blk_trsm(ctx, 0, &t11, &a12));  // U12 and the y part: L11^-1 [A12 | b]
        const rmhip_view_t row = view(ab_local, lr, j, w, width);
        return rmhip_blk_copy(ctx, &row, tile_out);
    };
    // post panel p: the owner factors and sends, everybody else posts the receive (asynchronous when there is someone to talk to)
    auto post_panel = [&](size_t p, rmhip_buf* tile_out, const std::function<int()>& before_trsm) -> int {
        const size_t j = p * rb, w = rb, width = ncols - j;
        const int owner = (int)(p % (size_t)world);
        rmhip_buf tile = 0;
        if (rank == owner && !failed) {
            const int rc = factor_panel(p, &tile, before_trsm);
            if (rc != RMHIP_OK) {
                failed = true;
                why = rmhip_last_error();
                tile = 0;
            }
        }
        if (!tile) {
            RMHIP_TRY(zeros(ctx, w, width, &tile));
            if (rank == owner) poison(tile, w, width);  // failed owner: everybody learns through the guard
        }
        temps.keep(tile);
        const rmhip_view_t tv = view(tile, 0, 0, w, width);
        if (world > 1) RMHIP_TRY(rmhip_comm_bcast(ctx, &tv, owner, overlap ? 1 : 0));
        *tile_out = tile;
        return RMHIP_OK;
    };
    // this rank's rows below `below`: multipliers against the tile (not on the owner, whose rows were factored with it) and the
    // trailing update restricted to tile columns [c0, c1) (tile-relative, c0 >= w)
    auto multipliers = [&](rmhip_buf tile, size_t j, size_t w, size_t below, bool is_owner) -> int {
        const size_t mb = nloc - below;
        if (mb == 0 || is_owner) return RMHIP_OK;
        const rmhip_view_t t11 = view(tile, 0, 0, w, w), a21 = view(ab_local, below, j, mb, w);
        RMHIP_TRY(rmhip_blk_trsm(ctx, 2, &t11, &a21));  // L21 = A21 U11^-1
        rmhip_buf g = 0, acc = 0;
        RMHIP_TRY(absmax_dev(ctx, &a21, &g));
        const int rc = rmhip_binary(ctx, RMHIP_MAX, growth_dev, g, &acc);
        (void)rmhip_free(ctx, g);
        RMHIP_TRY(rc);
        (void)rmhip_free(ctx, growth_dev);
        growth_dev = acc;
        return RMHIP_OK;
    };
    auto update = [&](rmhip_buf tile, size_t j, size_t w, size_t r0, size_t r1, size_t c0, size_t c1, bool on_side = false) -> int {
        if (r1 <= r0 || c1 <= c0) return RMHIP_OK;
        const rmhip_view_t l21 = view(ab_local, r0, j, r1 - r0, w), u12 = view(tile, 0, c0, w, c1 - c0), a22 = view(ab_local, r0, j + c0, r1 - r0, c1 - c0);
        if (on_side && side != main_stream) {
            // everything this product reads is final on the main stream by now (multipliers, the tile): the side stream starts behind it
            (void)hipEventRecord(ev_main_ready, main_stream);
            (void)hipStreamWaitEvent(side, ev_main_ready, 0);
            c->stream = side;
            // one eight-wave block per CU (84 KiB of LDS asked for) as on the look-ahead LU's update stream: the panel's one-workgroup
            // kernels take a CU whenever a block retires instead of queueing behind two resident blocks per CU
            const size_t keep_pad = c->gemm_lds_pad;
            c->gemm_lds_pad = side_pad;
            hipEvent_t t0 = timers.begin(side);
            const int rc = rmhip_blk_gemm(ctx, -1.0, &l21, &u12, 1.0, &a22);
            timers.end(PhaseTimers::UPDATE, t0, side);
            c->gemm_lds_pad = keep_pad;
            c->stream = main_stream;
            (void)hipEventRecord(ev_side_done, side);
            side_pending = true;
            return rc;
        }
        hipEvent_t t0 = timers.begin(main_stream);
        const int rc = rmhip_blk_gemm(ctx, -1.0, &l21, &u12, 1.0, &a22);
        timers.end(PhaseTimers::UPDATE, t0, main_stream);
        return rc;
    };

    rmhip_buf cur = 0;
    if (n_direct > 0) RMHIP_TRY(post_panel(0, &cur, nullptr));
    for (size_t p = 0; p < n_direct; ++p) {
        const size_t j = p * rb, w = rb, width = ncols - j;
        const int owner = (int)(p % (size_t)world);
        if (overlap) {
            hipEvent_t t0 = timers.begin(main_stream);
            const int wrc = rmhip_comm_wait(ctx);
            timers.end(PhaseTimers::WAIT, t0, main_stream);
            RMHIP_TRY(wrc);
        }
        const rmhip_buf tile = cur;
        const bool is_owner = rank == owner;
        const size_t below = is_owner ? local_row_offset(p) + w : first_local_row_at_or_after(p + 1);
        rmhip_buf nxt = 0;
        if (!failed) {
            int rc = multipliers(tile, j, w, below, is_owner);
            // depth-1 look-ahead: the owner of panel p + 1 brings that panel's columns (all its rows) and its tile's rows (all
            // columns) up to date first, factors and posts the broadcast; the rest of update p then runs under the transfer
            const bool next_mine = p + 1 < n_direct && (int)((p + 1) % (size_t)world) == rank;
            if (rc == RMHIP_OK && next_mine && side != main_stream) {
                // round 6: panel p + 1's columns on this stream, EVERYTHING else of update p (the next tile's rows included) on the side
                // stream - it runs while this stream factors panel p + 1; factor_panel joins the side stream before the interchanges.
                // Same products on the same operands as the one-stream order below (update-after-swap equals swap-after-update, row
                // by row): the results are bit-identical.
                const size_t lr1 = local_row_offset(p + 1);
                rc = update(tile, j, w, lr1, nloc, w, 2 * w);
                if (rc == RMHIP_OK) rc = update(tile, j, w, lr1, nloc, 2 * w, width, /*on_side=*/true);
                if (rc == RMHIP_OK) rc = post_panel(p + 1, &nxt, nullptr);
            } else if (rc == RMHIP_OK && next_mine) {
                const size_t lr1 = local_row_offset(p + 1);  // == below: the next owner's first block at or after p + 1 is p + 1 itself
                rc = update(tile, j, w, lr1, nloc, w, 2 * w);  // panel p + 1's columns, every row from its tile down
                if (rc == RMHIP_OK)
                    rc = post_panel(p + 1, &nxt, [&]() { return update(tile, j, w, lr1, lr1 + rb, 2 * w, width); });  // its tile's rows, the other columns
                if (rc == RMHIP_OK) rc = update(tile, j, w, lr1 + rb, nloc, 2 * w, width);  // everything else, under the transfer
            } else if (rc == RMHIP_OK) {
                if (p + 1 < n_direct) rc = post_panel(p + 1, &nxt, nullptr);
                if (rc == RMHIP_OK) rc = update(tile, j, w, below, nloc, w, width);
            }
            if (rc != RMHIP_OK) {
                failed = true;
                why = rmhip_last_error();
            }
        }
        if (failed && !nxt && p + 1 < n_direct) RMHIP_TRY(post_panel(p + 1, &nxt, nullptr));  // keep the collectives in step
        tiles.push_back(Tile{j, w, tile});
        cur = nxt;
    }
    // ---- the guard: one exchange, every rank decides the same way (a failed rank reports NaN)
    {
        join_side();
        hipEvent_t t_exch = timers.begin(main_stream);
        rmhip_buf mineb = 0, all = 0;
        const size_t one[2] = {1, 1};
        double growth = 0.0;
        if (failed) {
            RMHIP_TRY(rmhip_fill(ctx, std::numeric_limits<double>::quiet_NaN(), one, 2, &mineb));
            temps.keep(mineb);
        } else {
            mineb = growth_dev;  // the accumulated device scalar itself goes into the exchange
            if (world == 1) RMHIP_TRY(rmhip_read_scalar(ctx, growth_dev, 0, &growth));
        }
        double worst = failed ? std::numeric_limits<double>::quiet_NaN() : growth;
        if (world > 1) {
            RMHIP_TRY(rmhip_comm_allgather_f64(ctx, mineb, &all));
            temps.keep(all);
            // the first point where this rank's HOST blocks on the result of a collective: bounded, so that a peer that left without
            // being able to say so (killed, a device fault) costs a timeout and an error here instead of a hang
            RMHIP_TRY(rmhip_comm_wait_bounded(ctx, 0.0));
            std::vector<double> h((size_t)world);
            RMHIP_TRY(rmhip_download(ctx, all, h.data(), h.size()));
            worst = 0.0;
            for (double v : h)
                if (v != v || v > worst) worst = v;
        }
        timers.end(PhaseTimers::EXCHANGE, t_exch, main_stream);
        if (failed || !(worst <= tau)) exit_guard.agreed = true;  // every rank holds the same `worst` (NaN from a failed rank): all leave here
        if (failed) return fail(RMHIP_ERR_GROWTH, "mldivide_row_partitioned: rank %d failed (%s)", rank, why.c_str());
        if (!(worst <= tau))
            return fail(RMHIP_ERR_GROWTH, "largest multiplier outside the diagonal domains %.3g > %g (or a rank failed): use the block-column form (grid-wide pivot rule)",
                        worst, tau);
    }
    // ---- the remaining rows: gathered, then the single-GPU solve on every rank
    const size_t j0 = n_direct * rb, m_rem = n - j0;
    hipEvent_t t_tail = timers.begin(main_stream);
    rmhip_buf x = 0;
    RMHIP_TRY(zeros(ctx, n, nrhs, &x));
    temps.keep(x);
    if (m_rem > 0) {
        rmhip_buf trailing = 0;
        RMHIP_TRY(zeros(ctx, m_rem, m_rem + nrhs, &trailing));
        temps.keep(trailing);
        for (size_t q = n_direct; q < nblocks; ++q) {
            const size_t h = std::min(rb, n - q * rb);
            const int owner = (int)(q % (size_t)world);
            rmhip_buf blk = 0;
            if (rank == owner) {
                const rmhip_view_t src = view(ab_local, local_row_offset(q), j0, h, m_rem + nrhs);
                RMHIP_TRY(rmhip_blk_copy(ctx, &src, &blk));
            } else {
                RMHIP_TRY(zeros(ctx, h, m_rem + nrhs, &blk));
            }
            temps.keep(blk);
            const rmhip_view_t bv = view(blk, 0, 0, h, m_rem + nrhs);
            if (world > 1) RMHIP_TRY(rmhip_comm_bcast(ctx, &bv, owner, 0));
            const rmhip_view_t dst = view(trailing, q * rb - j0, 0, h, m_rem + nrhs);
            RMHIP_TRY(rmhip_blk_assign(ctx, &dst, blk));
            temps.drop(blk);
        }
        exit_guard.agreed = true;  // the last collective is behind us: what follows is local, on replicated data
        rmhip_buf a_rem = 0, b_rem = 0, x_rem = 0;
        const rmhip_view_t av = view(trailing, 0, 0, m_rem, m_rem), bv = view(trailing, 0, m_rem, m_rem, nrhs);
        RMHIP_TRY(rmhip_blk_copy(ctx, &av, &a_rem));
        temps.keep(a_rem);
        RMHIP_TRY(rmhip_blk_copy(ctx, &bv, &b_rem));
        temps.keep(b_rem);
        RMHIP_TRY(rmhip_mldivide(ctx, a_rem, b_rem, &x_rem));  // identical inputs on every rank: identical result, same status everywhere
        temps.keep(x_rem);
        const rmhip_view_t xd = view(x, j0, 0, m_rem, nrhs);
        RMHIP_TRY(rmhip_blk_assign(ctx, &xd, x_rem));
        temps.drop(trailing);
        temps.drop(a_rem);
        temps.drop(b_rem);
        temps.drop(x_rem);
    }
    exit_guard.agreed = true;  // (no remaining rows: the guard was the last collective)
    // ---- back substitution over the direct panels: every rank holds every tile row, so it is redundant and needs no exchange
    for (size_t k = tiles.size(); k-- > 0;) {
        const size_t j = tiles[k].j, w = tiles[k].w, width = ncols - j;
        const rmhip_buf tile = tiles[k].id;
        rmhip_buf rhs = 0;
        const rmhip_view_t yv = view(tile, 0, width - nrhs, w, nrhs);
        RMHIP_TRY(rmhip_blk_copy(ctx, &yv, &rhs));  // y_p
        temps.keep(rhs);
        const size_t later = n - j - w;
        const rmhip_view_t rv = view(rhs, 0, 0, w, nrhs);
        if (later > 0) {
            const rmhip_view_t u12 = view(tile, 0, w, w, later), xl = view(x, j + w, 0, later, nrhs);
            RMHIP_TRY(rmhip_blk_gemm(ctx, -1.0, &u12, &xl, 1.0, &rv));  // y_p - U12 x_later
        }
        const rmhip_view_t u11 = view(tile, 0, 0, w, w);
        RMHIP_TRY(rmhip_blk_trsm(ctx, 1, &u11, &rv));  // U11^-1
        const rmhip_view_t xd = view(x, j, 0, w, nrhs);
        RMHIP_TRY(rmhip_blk_assign(ctx, &xd, rhs));
        temps.drop(rhs);
        temps.drop(tile);
    }
    for (auto& id : temps.ids)
        if (id == x) id = 0;  // the result outlives the scope
    *out = x;
    timers.end(PhaseTimers::EXCHANGE, t_tail, main_stream);
    if (timers.on) timers.collect(c->rp_phase_ms);  // (synchronises on the last event: the caller reads x next anyway)
    return RMHIP_OK;
}

}  // extern "C"
