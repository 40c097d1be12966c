"""Developer tool (GPU box): the unfused implicit-expansion sequence of the reference's times builtin (times.rs:501-543) and the other
shape hooks at benchmark size, for rocprofv3 passes and HIP-event rates.
  A (8192 x 1) .* B (1 x 8192): broadcast_reps -> repmat(A, [1 8192]), repmat(B, [8192 1]) -> elem_mul -> free, `reps` times.
RMHIP_EAGER_REPMAT=1 materialises the two expansions (the three-pass form the lazy view replaces).  Usage: hooks_driver.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider

prov = HipProvider(0)
n = 8192
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
a = prov.fill_uniform(3, -1.0, 1.0, (n, 1))
b = prov.fill_uniform(4, -1.0, 1.0, (1, n))
x = prov.fill_uniform(5, -1.0, 1.0, (n, n))


def timed(label, fn, bytes_moved):
    fn()  # warm
    prov.synchronize()
    prov.timer_begin()
    for _ in range(reps):
        fn()
    ms = prov.timer_end() / reps
    print(f"{label:34s} {ms*1e3:9.1f} us  {bytes_moved/ms/1e9:7.3f} TB/s (algorithmic {bytes_moved/2**20:.0f} MiB)", flush=True)


def callers_sequence():
    le, re = prov.repmat(a, [1, n]), prov.repmat(b, [n, 1])
    h = prov.elem_mul(le, re)
    prov.free(le)
    prov.free(re)
    prov.free(h)


def materialise():
    t = prov.repmat(a, [1, n])
    prov.free(prov.reshape(t, (n * n, 1)))  # any non-elementwise consumer tiles the view


timed("repmat x2 -> elem_mul -> free", callers_sequence, 8 * n * n + 16 * n)
timed("repmat view materialised", materialise, 8 * n * n)
timed("permute [2 1] 8192^2", lambda: prov.free(prov.permute(x, [1, 0])), 16 * n * n)
x3 = prov.reshape(prov.fill_uniform(6, -1.0, 1.0, (n * n, 1)), (512, 256, 512))
timed("permute [3 1 2] 512x256x512", lambda: prov.free(prov.permute(x3, [2, 0, 1])), 16 * n * n)
timed("permute [1 3 2] 512x256x512", lambda: prov.free(prov.permute(x3, [0, 2, 1])), 16 * n * n)
timed("linspace 2^26", lambda: prov.free(prov.linspace(0.0, 1.0, n * n)), 8 * n * n)
timed("map_nan_to_zero 8192^2", lambda: prov.free(prov.map_nan_to_zero(x)), 16 * n * n)
timed("fill_like 8192^2", lambda: prov.free(prov.zeros_like(x)), 8 * n * n)
prov.synchronize()
print("ok")
