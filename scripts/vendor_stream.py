"""Developer tool (not used by the product): what the vendor stack streams on the same box -- torch's copy (hipMemcpyDtoD /
copy kernel), add, and the unfused sin(A)*B+C -- as a ceiling reference for the fused elementwise kernel.
Usage: vendor_stream.py [n]"""
import sys, time
import torch
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = torch.rand(n, n, dtype=torch.float64, device=dev); b = torch.rand_like(a); c = torch.rand_like(a); d = torch.empty_like(a)
def timed(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]
nb = a.numel() * 8
for name, f, streams in (("copy_ (d = a)", lambda: d.copy_(a), 2), ("torch.add(a, b, out=d)", lambda: torch.add(a, b, out=d), 3),
                         ("torch.addcmul(c, a, b, out=d)", lambda: torch.addcmul(c, a, b, out=d), 4),
                         ("unfused sin(a)*b+c (3 kernels)", lambda: torch.add(torch.mul(torch.sin(a), b), c, out=d), 8)):
    t = timed(f)
    print(f"{name}: {t*1e3:.3f} ms  {streams*nb/t/1e9:.0f} GB/s ({streams} streams of {nb>>20} MiB)", flush=True)
