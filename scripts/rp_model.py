"""Developer tool (GPU box): predicted times of the sharded paths at N = 1 / 2 / 4 / 8 GPUs from components timed on ONE GPU
(profiles/r06_multi_gpu_model.txt).  No multi-GPU run has ever been available; this is the arithmetic a measured SCALE run is to be laid
against, together with the per-phase device timers of the bench line (rmhip_rp_phase_ms).

Row-partitioned solve, n = 16384, rb = 512 (csrc/sharded.cpp): panel p (first column j = 512 p) is factored by its owner among its
(n - j) / N local rows, the tile row (512 x (n + 1 - j) doubles) is broadcast, every rank forms multipliers and updates its rows.  With the
round-6 overlap the update beyond the next panel's columns runs beside the next panel's factorisation:
    iteration(p, N) = la(p, N) + max( panel(p + 1, N), update(p, N), bcast(p + 1) ) + serial(p + 1)
    T(N)            = sum_p iteration(p, N) + exchange
panel(r) = rmhip_blk_lu of an r x 512 view (measured at r = 512 ... 16384, interpolated), serial = interchanges of the other columns +
U12 + tile copy (measured against the width), la / update = rank-512 products at the measured rate of the shard's shape, bcast = tile
bytes over ONE xGMI link at 80 % of 153 GB/s (the owner sends to its N - 1 peers over separate links), exchange = the measured tail."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider

prov = HipProvider(0)
n, rb = 16384, 512
buf = prov.fill_uniform(5, -1, 1, (n, n + 1))


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    prov.synchronize()
    prov.timer_begin()
    for _ in range(reps):
        fn()
    return prov.timer_end() / reps


def fresh():  # blk_lu works in place: a fresh random block per call keeps the panels well conditioned
    return prov.fill_uniform(9, -1, 1, (n, rb))


print("# components on one MI355X (ms)")
lu_ms = {}
for r in (512, 1024, 2048, 4096, 8192, 16384):
    ts = []
    for _ in range(4):
        blk = prov.fill_uniform(9, -1, 1, (r, rb))
        prov.synchronize()
        t0 = time.perf_counter()
        ip, info = prov.blk_lu((blk, 0, 0, r, rb))
        prov.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        prov.free(ip); prov.free(blk)
    lu_ms[r] = min(ts[1:])
    print(f"panel: rmhip_blk_lu {r:6d} x {rb}: {lu_ms[r]:.3f} ms")
serial_ms = {}
for width in (16384, 8192, 2048):
    blk = prov.fill_uniform(9, -1, 1, (4096, rb))
    ip, info = prov.blk_lu((blk, 0, 0, 4096, rb))
    def serial():
        prov.blk_swap_rows((buf, 0, 0, 4096, width), ip)
        prov.blk_trsm(0, (blk, 0, 0, rb, rb), (buf, 0, 0, rb, width))
        prov.free(prov.blk_copy((buf, 0, 0, rb, width)))
    serial_ms[width] = timed(serial)
    print(f"serial: interchanges of {width} columns (4096 rows) + U12 + tile copy: {serial_ms[width]:.3f} ms")
    prov.free(ip); prov.free(blk)
rate = {}
for rows in (512, 1024, 2048, 4096, 8192, 16384):
    cols = 8192
    ms = timed(lambda: prov.blk_gemm(-1e-9, (buf, 0, 0, rows, rb), (buf, 0, rb, rb, cols), 1.0, (buf, 0, 2 * rb, rows, cols)))
    rate[rows] = 2.0 * rows * cols * rb / ms / 1e9
    print(f"update: {rows:6d} x {cols} x {rb} rank-512 product: {ms:.3f} ms  {rate[rows]:.1f} TFLOP/s")
la_rate = {}
for rows in (512, 2048, 8192, 16384):
    ms = timed(lambda: prov.blk_gemm(-1e-9, (buf, 0, 0, rows, rb), (buf, 0, rb, rb, rb), 1.0, (buf, 0, 2 * rb, rows, rb)), reps=20)
    la_rate[rows] = 2.0 * rows * rb * rb / ms / 1e9
    print(f"look-ahead: {rows:6d} x {rb} x {rb}: {ms * 1e3:.1f} us  {la_rate[rows]:.1f} TFLOP/s")


def interp(table, x):
    ks = sorted(table)
    x = min(max(x, ks[0]), ks[-1])
    for a, b in zip(ks, ks[1:]):
        if a <= x <= b:
            w = (x - a) / (b - a)
            return table[a] * (1 - w) + table[b] * w
    return table[ks[-1]]


link = 0.8 * 153e9
exchange = float(os.environ.get("RP_EXCHANGE_MS", "7.7"))  # measured: guard + gathered tail + replicated tail solve + back substitution (phase timer)
print("\n# row-partitioned x = A\\b, n = 16384, rb = 512: predicted wall clock (ms)")
print(f"# {'N':>2s} {'sum la':>8s} {'sum max(panel,update,bcast)':>28s} {'sum serial':>11s} {'exchange':>9s} {'total':>8s}  (panel-bound iterations / update-bound / bcast-bound)")
npan = n // rb - 1
t1 = None
for N in (1, 2, 4, 8):
    s_la = s_mx = s_se = 0.0
    bound = [0, 0, 0]
    for p in range(npan):
        j = p * rb
        rows_loc = max((n - j - rb) / N, 1.0)
        width = n + 1 - j - rb
        la = 2.0 * rows_loc * rb * rb / (interp(la_rate, rows_loc) * 1e9)
        upd = 2.0 * rows_loc * max(width - rb, 0) * rb / (interp(rate, rows_loc) * 1e9)
        pan = interp(lu_ms, rows_loc)
        bc = (rb * width * 8 / link * 1e3) if N > 1 else 0.0
        m = max(pan, upd, bc)
        bound[[pan, upd, bc].index(m)] += 1
        s_la += la; s_mx += m; s_se += interp(serial_ms, width)
    tot = s_la + s_mx + s_se + exchange
    t1 = t1 or tot
    print(f"  {N:2d} {s_la:8.2f} {s_mx:28.2f} {s_se:11.2f} {exchange:9.2f} {tot:8.2f}  ({bound[0]} / {bound[1]} / {bound[2]})   speed-up vs N = 1: {t1 / tot:.2f}")
print("# measured at N = 1 this round (bench.py --workload mldivide, RMHIP_BENCH_FORCE_CYCLIC=1): 96.5 ms (phases: panel 82.6 incl. its wait for the")
print("# update it overlaps, update 63.6, exchange 7.7); the single-GPU solve (rmhip_mldivide) takes 67.6 ms.")

print("\n# row-sharded C = A*B 8192^3 (rmhip_matmul_row_sharded, B replicated, no collective in the timed region)")
b = prov.fill_uniform(12, -1, 1, (8192, 8192))
base = None
for N in (1, 2, 4, 8):
    a = prov.fill_uniform(11, -1, 1, (8192 // N, 8192))
    ms = timed(lambda: prov.free(prov.matmul(a, b)), reps=8)
    tf = 2.0 * (8192 // N) * 8192 * 8192 / ms / 1e9
    base = base or ms
    print(f"  N = {N}: per-GPU product {8192 // N} x 8192 x 8192: {ms:.3f} ms  {tf:.1f} TFLOP/s per GPU -> speed-up {base / ms:.2f}, {tf * N:.0f} TFLOP/s aggregate")
    prov.free(a)
print("# a replicated C adds one all-gather of 512 MiB: (N - 1) / N x 512 MiB over 7 links x 0.8 x 153 GB/s = 0.55 ms at N = 8")
prov.close()
