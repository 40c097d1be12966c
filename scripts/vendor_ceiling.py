"""Developer tool (not used by the product): the vendor libraries on the same box as a sanity reference --
rocBLAS dgemm through torch.matmul (SURVEY.md 8(d) config 3 asks for it) and the LU solve through torch.linalg."""
import time
import torch

dev = torch.device("cuda", 0)
def timed(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for n in (4096, 8192, 16384):
    a = torch.rand(n, n, dtype=torch.float64, device=dev) * 2 - 1
    b = torch.rand(n, n, dtype=torch.float64, device=dev) * 2 - 1
    dt = timed(lambda: torch.matmul(a, b), 5 if n <= 8192 else 2)
    print(f"rocBLAS/hipBLASLt dgemm n={n}: {dt*1e3:.2f} ms  {2.0*n**3/dt/1e12:.1f} TFLOP/s", flush=True)
    del a, b
for n in (4096, 8192, 16384):
    a = torch.rand(n, n, dtype=torch.float64, device=dev) * 2 - 1
    rhs = torch.rand(n, 1, dtype=torch.float64, device=dev)
    try:
        dt = timed(lambda: torch.linalg.solve(a, rhs), 2)
        print(f"torch.linalg.solve (vendor LU) n={n}: {dt*1e3:.1f} ms  {((2/3)*n**3+2*n*n)/dt/1e12:.2f} TFLOP/s", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"torch.linalg.solve n={n}: {type(e).__name__}: {str(e)[:120]}", flush=True)
    del a, rhs
