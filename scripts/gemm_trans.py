"""Developer tool: rate of the transposed-operand dgemm variants (views consumed in place) and syrk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = prov.fill_uniform(11, -1, 1, (n, n)); b = prov.fill_uniform(12, -1, 1, (n, n))
at, bt = prov.transpose(a), prov.transpose(b)
def run(tag, f, flops):
    for _ in range(2): prov.free(f())
    prov.timer_begin()
    for _ in range(8): prov.free(f())
    ms = prov.timer_end() / 8
    print(f"{tag}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s", flush=True)
fl = 2.0 * n ** 3
run("A*B  ", lambda: prov.matmul(a, b), fl)
run("A'*B ", lambda: prov.matmul(at, b), fl)
run("A*B' ", lambda: prov.matmul(a, bt), fl)
run("syrk ", lambda: prov.syrk(a), fl)
tall = prov.fill_uniform(13, -1, 1, (1 << 20, 256))
run("syrk 2^20 x 256", lambda: prov.syrk(tall), 2.0 * (1 << 20) * 256 * 256)
