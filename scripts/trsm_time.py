"""Developer tool: time the fused triangular solve base kernel (blk_trsm with a 128 x 128 triangle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from runmat_amd import HipProvider
prov = HipProvider(0)
N = 16384
buf = prov.fill_uniform(3, -1, 1, (N + 32, 8192 + 256))
for w in (32, 64, 128, 256, 512):
    for nc in (1, 128, 1024, 8192):
        vt, vb = (buf, 0, 0, w, w), (buf, 0, 256, w, nc)
        for upper in (False, True):
            for _ in range(3): prov.blk_trsm(upper, vt, vb)
            prov.timer_begin()
            for _ in range(40): prov.blk_trsm(upper, vt, vb)
            us = prov.timer_end() / 40 * 1e3
            print(f"w={w} nc={nc} {'upper' if upper else 'lower'}: {us:.1f} us", flush=True)
