"""Developer tool (GPU box, ONE GPU): component timings behind DESIGN.md section 6's predicted-time model of the two N > 1 paths the
driver's SCALE run will exercise, and the model's predictions for N = 1, 2, 4, 8.  Nothing here runs on more than one GPU: the model is
what a measured SCALE_rNN.json is to be compared with.

  row-sharded dgemm (BASELINE configs[2]): rank g computes C[rows_g, :] = A[rows_g, :] * B with B replicated; no collective in the timed
      region.  Component: the (8192 / N) x 8192 x 8192 product on one GPU.
  row-partitioned x = A\\b (configs[4], csrc/sharded.cpp): per panel p (512 columns) the owner factors its (n - j) / N rows of the panel,
      solves its tile row and broadcasts it (512 x (n + 1 - j) doubles); every rank then forms its multipliers and applies the rank-512
      update to its (n - j) / N rows; depth-1 look-ahead overlaps the broadcast with the previous update.  Components: the tall panel
      factorisation, the tile-row solve, the update at (n - j) / N rows, and the broadcast priced at the xGMI link rate.
Usage: python scripts/multi_gpu_model.py > profiles/r05_multi_gpu_model.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider

LINK_GBS = 153.0 * 0.8   # one xGMI link, one direction, at 80 % of its 153 GB/s (MI355X_MICROARCH.md): a broadcast leaves over 7 links at once
prov = HipProvider(0)


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    prov.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    prov.synchronize()
    return (time.perf_counter() - t0) / reps


print("== row-sharded dgemm 8192^3: per-GPU product (8192/N) x 8192 x 8192, B replicated, no collective in the timed region")
n = 8192
hb = prov.fill_uniform(12, -1, 1, (n, n))
t1 = None
for N in (1, 2, 4, 8):
    ha = prov.fill_uniform(11, -1, 1, (n // N, n))
    t = timed(lambda: prov.free(prov.matmul(ha, hb)))
    prov.free(ha)
    t1 = t1 or t
    print(f"N={N}: per-GPU {t*1e3:7.3f} ms = {2.0*(n//N)*n*n/t/1e12:5.1f} TFLOP/s per GPU; predicted whole-job {2.0*n**3/t/1e12:6.1f} TFLOP/s, "
          f"speed-up {t1/t:4.2f}x (all-gather of C for a replicated result, outside the timed region: {n*n*8*(N-1)/N/ (7*LINK_GBS*1e9) *1e3 if N>1 else 0:5.2f} ms over 7 links)")
prov.free(hb)

print("\n== row-partitioned x = A\\b, n = 16384, rb = 512: components on one GPU")
n, rb = 16384, 512
ab = prov.fill_uniform(41, -1, 1, (n, n + 1))
tile = prov.fill_uniform(43, -1, 1, (rb, n + 1))
comp = {}
for N in (1, 2, 4, 8):
    # mid-factorisation shapes: j = n / 2 (the sums below integrate the measured rate over j)
    rows = []
    for j in (0, n // 4, n // 2, 3 * n // 4):
        m = max(rb, (n - j) // N)
        width = n + 1 - j - rb
        t_upd = timed(lambda: prov.blk_gemm(-1.0, (ab, 0, 0, m, rb), (tile, 0, rb, rb, width), 1.0, (ab, 0, rb, m, width)), reps=3, warm=1)
        t_mult = timed(lambda: prov.blk_trsm(2, (tile, 0, 0, rb, rb), (ab, 0, 0, m, rb)), reps=3, warm=1)
        rows.append((j, m, width, t_upd, t_mult))
    comp[N] = rows
    print(f"N={N}: " + "; ".join(f"j={j}: update {m}x{w}x512 {tu*1e3:6.2f} ms ({2.0*m*w*rb/tu/1e12:4.1f} TF/s), multipliers {tm*1e3:5.2f} ms" for j, m, w, tu, tm in rows))
# panel factorisation + tile-row solve of the owner (rows/N tall, 512 wide; its own rows only): timed through the library's driver at N = 1
bvec = prov.fill_uniform(42, -1, 1, (n, 1))


def full():
    w = prov.fill_uniform(41, -1, 1, (n, n + 1))
    x = prov.mldivide_row_partitioned(w, n, 1, rb=rb)
    prov.free(x)
    prov.free(w)


def fill_only():
    prov.free(prov.fill_uniform(41, -1, 1, (n, n + 1)))


t_full = timed(full, reps=3, warm=1) - timed(fill_only, reps=3, warm=1)
print(f"measured N=1 driver (rmhip_mldivide_row_partitioned, solve-path panels, look-ahead 1): {t_full*1e3:.1f} ms")
# model: sum over panels of  panel(p) + max(update(p), bcast(p+1))  with update and multipliers from the measured rates (piecewise linear in j)
def interp(rows, j, k):
    js = [r[0] for r in rows]
    vs = [r[k] for r in rows]
    return float(np.interp(j, js, vs))


# the owner's panel + tile-row solve: chain-latency bound (about 64-column base panels x 8 per panel); calibrated so that N = 1 reproduces the measurement
npan = n // rb
upd1 = sum(interp(comp[1], p * rb, 3) + interp(comp[1], p * rb, 4) for p in range(npan - 1))
t_panel = max(0.0, (t_full - upd1) / npan)
print(f"calibration at N=1: updates + multipliers {upd1*1e3:.1f} ms of {t_full*1e3:.1f} ms -> panel + tile solve + gather/back-substitution share {t_panel*1e3:.2f} ms per panel")
for N in (1, 2, 4, 8):
    total, bsum = 0.0, 0.0
    for p in range(npan - 1):
        j = p * rb
        upd = interp(comp[N], j, 3) + (interp(comp[N], j, 4) if N > 1 else interp(comp[1], j, 4))
        bc = rb * (n + 1 - j) * 8 / (LINK_GBS * 1e9) if N > 1 else 0.0   # the tile row goes out over the owner's links in parallel: one link's worth of time
        bsum += bc
        total += t_panel + max(upd, bc)   # depth-1 look-ahead: the next tile's broadcast runs under this update
    total += t_panel
    print(f"N={N}: predicted {total*1e3:7.1f} ms = {((2/3)*n**3+2*n*n)/total/1e12:5.1f} TFLOP/s whole job, speed-up {(t_full)/total:4.2f}x vs N=1 (broadcasts {bsum*1e3:.1f} ms in total, hidden under updates where shorter)")
prov.close()
