"""Developer tool (GPU box): the trailing update of the blocked LU as the solver issues it - C <- C - A B IN PLACE on sub-blocks of one
padded workspace (rmhip_blk_gemm, alpha = -1, beta = 1) - at the panel widths and remaining orders of the n = 16384 solve, standalone
(nothing else on the device): TFLOP/s from HIP events.  Kernel knobs from the environment (RMHIP_GEMM_*).  Usage: lu_update_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
N, LD = 16384, 16416
w = prov.fill_uniform(7, -1e-3, 1e-3, (LD, N))
tag = " ".join(f"{k[6:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RMHIP_GEMM"))
for m in (16384 - 512, 12288, 8192, 4096):
    for k in (128, 256, 512, 1024):
        if k >= m:
            continue
        j = N - m - k if N - m - k >= 0 else 0
        mm = N - j - k
        a = (w, j + k, j, mm, k)          # L21: rows below the panel, the panel's columns
        b = (w, j, j + k, k, mm)          # U12: the panel's rows, the trailing columns
        cc = (w, j + k, j + k, mm, mm)    # A22
        for _ in range(2): prov.blk_gemm(-1.0, a, b, 1.0, cc)
        best = 1e9
        for _ in range(3):
            prov.timer_begin()
            for _ in range(4): prov.blk_gemm(-1.0, a, b, 1.0, cc)
            best = min(best, prov.timer_end() / 4)
        print(f"[{tag or 'default'}] update {mm} x {mm} x {k}: {best*1e3:8.1f} us  {2.0*mm*mm*k/best/1e9:6.1f} TFLOP/s", flush=True)
