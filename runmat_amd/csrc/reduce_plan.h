// reduce_plan.h -- launch geometry for the [pre, red, post] reduction skeletons (skel_reduce.h).
// Shared by the ahead-of-time reductions and the hipRTC fused reductions.
//
// Contrast with the reference: its generated reduction runs ONE workgroup per slice
// (crates/runmat-accelerate/src/fusion.rs:1983-2030), i.e. a single CU for `sum(x,'all')`.
// Here every shape is split until the grid covers the 256 CUs several times over, with a
// deterministic second stage.
#pragma once

#include <cstddef>
#include <cstdint>

namespace rmhip {

struct ReducePlan {
    bool contiguous;  // kernel A (pre == 1) or kernel B
    uint64_t nslices; // pre * post
    uint64_t nsplit;  // partials per slice
    int tx;           // kernel B: threads along `pre` per block (power of two)
    unsigned gx, gy, gz;
    bool valid;       // false: geometry exceeds grid limits (caller reports UNSUPPORTED)
};

inline uint64_t ceil_div_u64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Kernel B walks `red` in nsplit chunks of ceil(red / nsplit) columns and the blocks of all chunks run in lockstep.  When a
// chunk spans a multiple of 256 KiB (power-of-two shapes: 8192 x 8192 f64 in 64 chunks = 8 MiB each) every block is at the
// same offset of its chunk at the same time; 65 chunks of 127 columns instead measured 112 -> 105 us for sum(x,2) at 8192^2
// and 115 -> 102 us at 16384 x 4096 (scripts/red_chunk_ab.sh).  The block COUNT matters more (reduce_kernels.hip); this
// only moves the generic kernel off its worst point.
inline uint64_t dealias_nsplit(uint64_t red, uint64_t nsplit, uint64_t column_stride_bytes, uint64_t max_split) {
    if (nsplit <= 1 || column_stride_bytes == 0) return nsplit;
    for (int tries = 0; tries < 8 && nsplit < max_split; ++tries) {
        const uint64_t chunk = ceil_div_u64(red, nsplit);
        if (chunk <= 8 || (chunk * column_stride_bytes) % (256u * 1024u) != 0) break;
        ++nsplit;
    }
    return nsplit;
}

// Requires pre >= 1 and post >= 1 (callers return early when there are no output slices).
// `elem_bytes`: storage width of the reduced tensor (8, or 4 on a precision-32 provider).
inline ReducePlan plan_reduction(uint64_t pre, uint64_t red, uint64_t post, int num_cus, unsigned elem_bytes = 8) {
    ReducePlan p{};
    if (pre == 0 || post == 0) {
        p.valid = false;
        return p;
    }
    const uint64_t target_blocks = (uint64_t)num_cus * 8;  // ~2048 blocks of 256 threads
    p.nslices = pre * post;
    if (pre == 1) {
        p.contiguous = true;
        // slices of 64 KiB and more stream with RM_ABLOCK threads, shorter ones keep RM_RBLOCK (more resident blocks hide
        // the fixed per-block reduction latency)
        const uint64_t bs = red * elem_bytes >= 65536 ? 1024 : 256;
        uint64_t max_split = ceil_div_u64(red, bs * 8);  // >= 8 elements per thread per block
        if (max_split < 1) max_split = 1;
        uint64_t want = ceil_div_u64(target_blocks, post ? post : 1);
        if (want < 1) want = 1;
        p.nsplit = want < max_split ? want : max_split;
        if (p.nsplit > 4096) p.nsplit = 4096;
        p.tx = (int)bs;  // block size of kernel A
        p.gx = (unsigned)p.nsplit;
        // slices spread over (y, z)
        uint64_t gy = post < 65535 ? post : 65535;
        if (gy < 1) gy = 1;
        p.gy = (unsigned)gy;
        p.gz = (unsigned)ceil_div_u64(post ? post : 1, gy);
    } else {
        p.contiguous = false;
        int tx = 256;
        while (tx > 1 && (uint64_t)tx / 2 >= pre) tx /= 2;  // smallest power of two >= pre, capped at 256
        p.tx = tx;
        const uint64_t bx = ceil_div_u64(pre, (uint64_t)tx);
        const uint64_t ty = 256 / tx;
        uint64_t max_split = ceil_div_u64(red, 16 * ty);  // >= 16 elements per thread
        if (max_split < 1) max_split = 1;
        uint64_t want = ceil_div_u64(target_blocks, bx * (post ? post : 1));
        if (want < 1) want = 1;
        p.nsplit = want < max_split ? want : max_split;
        p.nsplit = dealias_nsplit(red, p.nsplit, pre * elem_bytes, max_split);
        if (p.nsplit > 65535) p.nsplit = 65535;
        p.gx = (unsigned)bx;
        p.gy = (unsigned)p.nsplit;
        p.gz = (unsigned)(post ? post : 1);
    }
    p.valid = p.gz <= 65535u && p.gy <= 65535u && p.gx >= 1;
    if (!p.contiguous && post > 65535) p.valid = false;
    return p;
}

// Geometry of kernel B over 16-byte vectors (a thread owns two adjacent lines; reduce_kernels.hip k_reduce_strided_v2, reduce2.hip,
// generated rm_red_strided2): `bx` windows of `win` pairs along `pre`, their number a multiple of the XCD count - workgroups go to the
// XCDs round robin in launch order (x fastest), so with bx % xcds == 0 a window is always walked by the same XCD whatever the chunk -,
// balanced (128-byte granules), a block of as many waves as its window needs, and `blocks_per_cu` blocks per CU over (bx, nsplit, post).
struct StridedWidePlan {
    unsigned bx, win, threads;
    uint64_t nsplit;
};
inline StridedWidePlan plan_strided_wide(uint64_t pre, uint64_t red, uint64_t post, int num_cus, int xcds_probed, unsigned elem_bytes,
                                         int blocks_per_cu = 3, bool pin_xcds = true) {
    StridedWidePlan w{};
    const unsigned xcds = xcds_probed > 0 ? (unsigned)xcds_probed : 8u;
    const uint64_t npairs = (pre + 1) / 2;
    const bool pin = pin_xcds && pre / 2 >= (uint64_t)xcds * 64;
    w.bx = (unsigned)ceil_div_u64(npairs, 256);
    if (pin) w.bx = (w.bx + xcds - 1) / xcds * xcds;
    w.win = (unsigned)((ceil_div_u64(npairs, w.bx) + 7) / 8 * 8);
    if (w.win > 256) w.win = 256;
    w.bx = (unsigned)ceil_div_u64(npairs, w.win);
    if (pin) w.bx = (w.bx + xcds - 1) / xcds * xcds;  // (trailing windows may be empty)
    w.threads = (w.win + 63) / 64 * 64;
    uint64_t want = ceil_div_u64((uint64_t)num_cus * blocks_per_cu, (uint64_t)w.bx * (post ? post : 1));
    const uint64_t max_split = ceil_div_u64(red, 16);
    w.nsplit = want < 1 ? 1 : want;
    if (w.nsplit > max_split) w.nsplit = max_split;
    w.nsplit = dealias_nsplit(red, w.nsplit, pre * elem_bytes, max_split);
    if (w.nsplit > 65535) w.nsplit = 65535;
    return w;
}

}  // namespace rmhip
