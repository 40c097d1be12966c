"""Developer tool: HIP-event times of the latency-sensitive paths (sum all, image_normalize) for A/B across builds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import runmat_amd._lib as L
for k in ("rmhip_set_precision", "rmhip_buffer_bits"):
    if os.environ.get("RMHIP_AB_OLD"):
        L.SIGNATURES.pop(k, None)
from runmat_amd import HipProvider
p = HipProvider(0)
N = 8192
a = p.fill_uniform(1, -3.0, 3.0, (N, N))
def timed(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    p.timer_begin()
    for _ in range(reps): fn()
    return round(p.timer_end() / reps * 1e3, 2)
f = lambda h: p.free(h)
out = {"sum_all_us": timed(lambda: f(p.reduce_sum(a))), "mean_all_us": timed(lambda: f(p.reduce_mean(a))),
       "max_all_us": timed(lambda: f(p.reduce_max(a)))}
img = p.fill_uniform(4, 0.0, 1.0, (16, 2160, 3840))
out["image_normalize_us"] = timed(lambda: f(p.image_normalize(img, 16, 2160, 3840, 1e-6, gain=1.2, bias=0.05)), reps=10, warm=2)
out["image_normalize_gamma_us"] = timed(lambda: f(p.image_normalize(img, 16, 2160, 3840, 1e-6, gain=1.2, bias=0.05, gamma=1.8)), reps=10, warm=2)
small = p.fill_uniform(5, 0.0, 1.0, (4, 256, 256))
out["image_normalize_small_us"] = timed(lambda: f(p.image_normalize(small, 4, 256, 256, 1e-6)), reps=30, warm=3)
print(json.dumps(out))
