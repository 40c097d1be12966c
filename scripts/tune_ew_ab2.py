"""Developer tool: interleaved A/B of grid cap / chunking / non-temporal hints for the headline fused kernel at block 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from runmat_amd import HipProvider
from planner_requests import sin_mul_add_plan
prov = HipProvider(0)
n = 8192
ins = [prov.fill_uniform(1, -np.pi, np.pi, (n, n)), prov.fill_uniform(2, -1, 1, (n, n)), prov.fill_uniform(3, -1, 1, (n, n))]
p, o = sin_mul_add_plan(); sh = p.generate_wgsl_for_output(o)
base = dict(RMHIP_EW_BLOCKS_PER_CU="16", RMHIP_EW_CHUNKED="0", RMHIP_EW_NT_LOAD="1", RMHIP_EW_NT_STORE="1")
configs = [dict(base)] + [dict(base, RMHIP_EW_BLOCKS_PER_CU=b) for b in ("2", "4", "8", "32", "64")] + \
          [dict(base, RMHIP_EW_CHUNKED="1"), dict(base, RMHIP_EW_CHUNKED="1", RMHIP_EW_BLOCKS_PER_CU="2"),
           dict(base, RMHIP_EW_NT_LOAD="0"), dict(base, RMHIP_EW_NT_STORE="0"), dict(base, RMHIP_EW_NT_LOAD="0", RMHIP_EW_NT_STORE="0")]
def run(cfg, reps=15):
    os.environ.update(cfg)
    prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    prov.timer_begin()
    for _ in range(reps): prov.free(prov.fused_elementwise(sh, ins, (n, n), n * n))
    return prov.timer_end() / reps
res = {i: [] for i in range(len(configs))}
for rnd in range(8):
    order = list(range(len(configs)))
    if rnd % 2: order.reverse()
    for i in order: res[i].append(run(configs[i]))
for i, cfg in enumerate(configs):
    v = sorted(res[i]); med = v[len(v) // 2]
    tag = " ".join(f"{k[9:].lower()}={cfg[k]}" for k in sorted(cfg))
    print(f"{32.0*n*n/med/1e6:.0f} GB/s  {tag}", flush=True)
