"""Developer tool (GPU box): sum(x,2) / sum(x,1) rates over shapes.  Usage: red_shapes.py [rows cols]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
args = [int(a) for a in sys.argv[1:]]
shapes = list(zip(args[0::2], args[1::2])) or [(8192, 8192), (8200, 8192), (8256, 8192), (7936, 8192), (8192, 8000), (16384, 4096), (16400, 4096), (8190, 8192), (5000, 13000)]
for shape in shapes:
    a = prov.fill_uniform(1, -1, 1, shape)
    N = shape[0] * shape[1] * 8.0
    for _ in range(3): prov.free(prov.reduce_sum_dim(a, 1))
    best = 1e9
    for _ in range(4):
        prov.timer_begin()
        for _ in range(20): prov.free(prov.reduce_sum_dim(a, 1))
        best = min(best, prov.timer_end() / 20)
    best0 = 1e9
    for _ in range(4):
        prov.timer_begin()
        for _ in range(20): prov.free(prov.reduce_sum_dim(a, 0))
        best0 = min(best0, prov.timer_end() / 20)
    print(f"{shape}: sum(x,2) {best*1e3:.1f} us {N/best/1e6:.0f} GB/s | sum(x,1) {best0*1e3:.1f} us {N/best0/1e6:.0f} GB/s", flush=True)
    prov.free(a)
