"""Developer tool (GPU box): time fft_dim on resident tensors - per call, device-synchronised (the call returns after its launches are
queued; a read_scalar-free sync is the download of one small tensor) - and print the algorithmic HBM rate: bytes of the input read
once + the complex result written once."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from runmat_amd.provider import HipProvider

p = HipProvider()
sync = p.upload(np.zeros((1, 1)))


def run(label, h, length, dim, inverse=False, reps=10):
    f = p.ifft_dim if inverse else p.fft_dim
    out = f(h, length, dim); p.download(sync); p.free(out)
    t = time.perf_counter()
    for _ in range(reps):
        out = f(h, length, dim)
        p.free(out)
    p.download(sync)
    dt = (time.perf_counter() - t) / reps
    n_in = int(np.prod(h.shape)) * (16 if p.is_complex(h) else 8)
    shape = list(h.shape) + [1] * max(0, dim + 1 - len(h.shape))
    if length is not None:
        shape[dim] = length
    n_out = int(np.prod(shape)) * 16
    print(f"{label:44s} {dt*1e3:9.3f} ms   {(n_in + n_out) / dt / 1e9:8.1f} GB/s algorithmic")


m = p.fill_uniform(3, -1.0, 1.0, (8192, 8192))
run("8192x8192 real, columns (dim 0)", m, None, 0)
run("8192x8192 real, rows (dim 1)", m, None, 1)
c = p.fft_dim(m, None, 0)
run("8192x8192 complex, columns", c, None, 0)
run("8192x8192 complex, inverse columns", c, None, 0, True)
run("8192x8192 complex, rows", c, None, 1)
p.free(c)
m4 = p.fill_uniform(4, -1.0, 1.0, (4096, 16384))
run("4096x16384 real, columns (one pass)", m4, None, 0)
m1 = p.fill_uniform(4, -1.0, 1.0, (1024, 65536))
run("1024x65536 real, columns", m1, None, 0)
m6 = p.fill_uniform(4, -1.0, 1.0, (64, 1 << 20))
run("64x2^20 real, columns", m6, None, 0)
v = p.fill_uniform(6, -1.0, 1.0, (1 << 24, 1))
run("2^24 vector", v, None, 0)
v2 = p.fill_uniform(6, -1.0, 1.0, (1 << 20, 1))
run("2^20 vector", v2, None, 0)
t = p.fill_uniform(7, -1.0, 1.0, (1000, 8192))
run("1000x8192 real, columns (Bluestein)", t, None, 0)
t2 = p.fill_uniform(7, -1.0, 1.0, (8192, 1000))
run("8192x1000 real, rows (Bluestein)", t2, None, 1)
v3 = p.fill_uniform(8, -1.0, 1.0, (1000000, 1))
run("10^6 vector (Bluestein)", v3, None, 0)
