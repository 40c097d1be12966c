"""Developer tool (GPU box): does a re-read of a recently streamed buffer come from the memory-side cache?  sum(x,'all') over buffers
of growing size, repeated back to back; and a copy-like pass (scalar_mul) for read+write.  Usage: mall_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 8
    x = prov.fill_uniform(1, -1, 1, (n, 1))
    for name, f, streams in (("sum", lambda: prov.reduce_sum(x), 1), ("scalar_mul", lambda: prov.scalar_mul(x, 1.5), 2)):
        for _ in range(3): prov.free(f())
        prov.synchronize(); reps = 20
        t0 = time.perf_counter()
        hs = [f() for _ in range(reps)]
        prov.synchronize(); dt = (time.perf_counter() - t0) / reps
        for h in hs: prov.free(h)
        print(f"{mb:5d} MiB {name:10s}: {dt*1e6:8.1f} us  {streams*mb*1.048576e6/dt/1e9:8.0f} GB/s", flush=True)
    prov.free(x)
