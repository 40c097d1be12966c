"""Developer tool: image_normalize timing (4k frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from runmat_amd import HipProvider
prov = HipProvider(0)
B, H, W = 16, 2160, 3840
x = prov.fill_uniform(3, 0.0, 1.0, (B, H, W))
n = B * H * W
for tag, kw in (("gain+bias+clamp+gamma", dict(gain=1.05, bias=-0.02, gamma=1.8)), ("gain+bias+clamp", dict(gain=1.05, bias=-0.02)),
                ("plain", dict(clamp_zero=False))):
    for _ in range(3): prov.free(prov.image_normalize(x, B, H, W, 1e-6, **kw))
    prov.timer_begin()
    for _ in range(10): prov.free(prov.image_normalize(x, B, H, W, 1e-6, **kw))
    ms = prov.timer_end() / 10
    print(f"image_normalize {B} x {H} x {W} f64 [{tag}]: {ms:.3f} ms  {32.0*n/ms/1e6:.0f} GB/s on 32 B/element", flush=True)
