"""Developer tool (GPU box): what the in-place rank-k update of the blocked LU loses to its LAYOUT - the same C <- C - A B (rmhip_blk_gemm) with
the operands where the solver has them (sub-blocks of one padded workspace) against packed copies of A (ld = m), B (ld = k) and / or C, and
against beta = 0 (no read of C); plus a deep product on the workspace's leading dimension.  TFLOP/s from HIP events, nothing else on the device.
Usage: lu_update_layouts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
N, LD = 16384, 16416
w = prov.fill_uniform(7, -1e-3, 1e-3, (LD, N))


def rate(label, a, b, beta, cc, m, n, k):
    for _ in range(2): prov.blk_gemm(-1.0, a, b, beta, cc)
    best = 1e9
    for _ in range(3):
        prov.timer_begin()
        for _ in range(4): prov.blk_gemm(-1.0, a, b, beta, cc)
        best = min(best, prov.timer_end() / 4)
    print(f"{label:58s} {m} x {n} x {k}: {best*1e3:8.1f} us  {2.0*m*n*k/best/1e9:6.1f} TFLOP/s", flush=True)


for k in (512, 256, 128):
    j = 0
    mm = N - k
    ap, bp, cp = prov.fill_uniform(8, -1e-3, 1e-3, (mm, k)), prov.fill_uniform(9, -1e-3, 1e-3, (k, mm)), prov.fill_uniform(10, -1e-3, 1e-3, (mm, mm))
    A, B, C = (w, k, 0, mm, k), (w, 0, k, k, mm), (w, k, k, mm, mm)
    AP, BP, CP = (ap, 0, 0, mm, k), (bp, 0, 0, k, mm), (cp, 0, 0, mm, mm)
    rate("in place (solver layout)", A, B, 1.0, C, mm, mm, k)
    rate("A packed", AP, B, 1.0, C, mm, mm, k)
    rate("B packed", A, BP, 1.0, C, mm, mm, k)
    rate("A and B packed", AP, BP, 1.0, C, mm, mm, k)
    rate("C separate (ld = m)", A, B, 1.0, CP, mm, mm, k)
    rate("all packed", AP, BP, 1.0, CP, mm, mm, k)
    rate("in place, beta = 0 (C not read)", A, B, 0.0, C, mm, mm, k)
    rate("all packed, beta = 0", AP, BP, 0.0, CP, mm, mm, k)
    for h in (ap, bp, cp): prov.free(h)
rate("deep product on the workspace ld", (w, 0, 0, 8192, 8192), (w, 0, 8192, 8192, 8192), 0.0, (w, 8192, 8192, 8192, 8192), 8192, 8192, 8192)
rate("deep product on the workspace ld, beta = 1", (w, 0, 0, 8192, 8192), (w, 0, 8192, 8192, 8192), 1.0, (w, 8192, 8192, 8192, 8192), 8192, 8192, 8192)
