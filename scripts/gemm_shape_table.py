"""Developer tool (GPU box): the special-shape GEMM table of round 6 (profiles/r06_gemm_shapes.txt) - every class of product the hot path
issues, at the final kernels, each with its fraction of the fp64 matrix-core peak of the box: square, tall-skinny, small-k (the LU's own
inner updates, C -= A*B in place inside a padded leading dimension), few-tile / long-k (split-K), transposed views, the epilogue form.
Reference kernels these stand in for: crates/runmat-accelerate/src/backend/wgpu/shaders/matmul_smallk.rs, matmul_tall_skinny.rs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider

prov = HipProvider(0)
info = prov.device_info_struct()
peak = int(info["compute_units"]) * 4 * 32 * int(info["clock_mhz"]) * 1e6 / 1e12
print(f"# fp64 matrix-core peak of this box: {peak:.1f} TFLOP/s ({info['compute_units']} CUs x {info['clock_mhz']} MHz); times by HIP events on the library's stream")
print(f"# {'class':34s} {'shape (m x n x k)':26s} {'time':>10s} {'TFLOP/s':>9s} {'frac':>6s}  kernel path")


def timed(fn, flops):
    reps = 30 if flops < 2e10 else (10 if flops < 3e11 else 4)
    for _ in range(2):
        fn()
    prov.synchronize()
    prov.timer_begin()
    for _ in range(reps):
        fn()
    return prov.timer_end() / reps


def row(cls, m, n, k, ms, path):
    tf = 2.0 * m * n * k / ms / 1e9
    print(f"  {cls:34s} {f'{m} x {n} x {k}':26s} {ms * 1e3:8.1f} us {tf:9.2f} {tf / peak:6.3f}  {path}", flush=True)


def matmul(cls, m, n, k, path, ta=False, tb=False):
    a = prov.fill_uniform(1, -1, 1, (k, m) if ta else (m, k))
    b = prov.fill_uniform(2, -1, 1, (n, k) if tb else (k, n))
    va = prov.transpose(a) if ta else a
    vb = prov.transpose(b) if tb else b
    ms = timed(lambda: prov.free(prov.matmul(va, vb)), 2.0 * m * n * k)
    row(cls, m, n, k, ms, path)
    for h in {id(x): x for x in (a, b, va, vb)}.values():
        prov.free(h)


# dense / square
for nn in (2048, 4096, 8192):
    matmul("square", nn, nn, nn, "k_dgemm_w8 (eight waves, 128x128x16)")
# tall-skinny outputs (reference: matmul_tall_skinny.rs) and skinny products
matmul("tall-skinny", 100000, 64, 64, "k_dgemm_small (64x64 tiles)")
matmul("tall-skinny", 65536, 128, 128, "k_dgemm_small / w8")
matmul("tall-skinny", 16384, 256, 256, "k_dgemm_w8")
matmul("tall-skinny", 262144, 32, 32, "k_dgemm_small guarded")
matmul("wide-short", 64, 100000, 64, "k_dgemm_small")
# small k (reference: matmul_smallk.rs): outer-product-like updates
for k in (16, 64, 128, 256, 512):
    matmul("small k, full output", 8192, 8192, k, "k_dgemm_w8")
# few output tiles, long k: split-K
matmul("split-K", 512, 512, 131072, "k_dgemm_w8 over k slices + k_reduce_splits")
matmul("split-K", 1024, 1024, 65536, "k_dgemm_w8 over k slices + k_reduce_splits")
matmul("split-K", 128, 128, 262144, "k_dgemm_w8 over k slices + k_reduce_splits")
# transposed views (A'*B, A*B')
matmul("A' * B view", 8192, 8192, 8192, "k_dgemm_w8<TA>", ta=True)
matmul("A * B' view", 8192, 8192, 8192, "k_dgemm<TB>", tb=True)
matmul("A' * B tall (Gram)", 1024, 1024, 131072, "k_dgemm_w8<TA> split-K", ta=True)
# ragged (guarded tiles)
matmul("ragged", 8191, 8193, 8190, "k_dgemm_w8g (guarded)")
matmul("ragged", 5000, 3000, 1000, "k_dgemm_w8g (guarded)")

# the LU's own in-place updates C -= A*B inside one padded buffer (ld = 16384 + 32), alone on the device
N = 16384
ld = N + 32
buf = prov.fill_uniform(11, -1, 1, (ld, N))
for (m, n, k, what) in [(16128, 256, 256, "look-ahead update of one 256-column panel"), (16128, 64, 64, "in-panel update, 64-deep"),
                        (16000, 128, 128, "in-panel update, 128-deep"), (14336, 1792, 256, "inner rank-256 update of a super-panel's rest"),
                        (12288, 2048, 2048, "boundary: rank-2048 update of the next super-panel"), (12288, 12288, 2048, "far stream: deep rank-2048 update"),
                        (8192, 8192, 512, "one-level driver: rank-512 update"), (2048, 256, 1024, "W-wide solve: inner product of the recursion"),
                        (1024, 8192, 1024, "W-wide solve: inner product, many columns"), (128, 8192, 128, "W-wide solve: leaf-level product")]:
    va, vb, vc = (buf, k, 0, m, k), (buf, 0, k, k, n), (buf, k, k, m, n)
    ms = timed(lambda: prov.blk_gemm(-1e-9, va, vb, 1.0, vc), 2.0 * m * n * k)
    row("LU update: " + what[:22], m, n, k, ms, what)
prov.free(buf)
prov.close()
