"""Development aid (context only, not a product path): vendor-library timings on the same box --
torch.linalg.cholesky (hipSOLVER/MAGMA), torch.matmul (hipBLASLt/rocBLAS), solve_triangular."""
import time, sys
import torch
dev = torch.device("cuda")
def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
for dt, n in ((torch.float64, 16384), (torch.float32, 32768)):
    try:
        a = torch.randn(n, 8, dtype=dt, device=dev)
        k = torch.exp(-0.5 * torch.cdist(a, a) ** 2) + 0.1 * torch.eye(n, dtype=dt, device=dev)
        t = timeit(lambda: torch.linalg.cholesky(k))
        print(f"torch.linalg.cholesky {dt} n={n}: {1e3*t:.1f} ms  {n**3/3/t/1e12:.1f} TFLOP/s", flush=True)
        l = torch.linalg.cholesky(k)
        b = torch.randn(n, 2048, dtype=dt, device=dev)
        t = timeit(lambda: torch.linalg.solve_triangular(l, b, upper=False))
        print(f"torch solve_triangular {dt} n={n} nrhs=2048: {1e3*t:.1f} ms  {n*n*2048/t/1e12:.1f} TFLOP/s", flush=True)
        del l, k
        m = 8192
        x = torch.randn(m, m, dtype=dt, device=dev); yv = torch.randn(m, m, dtype=dt, device=dev)
        t = timeit(lambda: x @ yv.T)
        print(f"torch.matmul {dt} {m}^3: {1e3*t:.2f} ms  {2*m**3/t/1e12:.1f} TFLOP/s", flush=True)
    except Exception as e:
        print("failed:", dt, repr(e)[:200], flush=True)
