"""What does the vendor DGEMM reach on this box (fp64, via torch -> rocBLAS/hipBLASLt)?
Known-good reference for the attainable fp64 matrix rate; NOT used by the product."""
import time, torch
dev = torch.device("cuda:0")
def bench(m, n, k, ta=False, reps=10):
    a = torch.randn((k, m) if ta else (m, k), dtype=torch.float64, device=dev)
    b = torch.randn((k, n), dtype=torch.float64, device=dev)
    f = (lambda: a.T @ b) if ta else (lambda: a @ b)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"DGEMM m={m} n={n} k={k} transA={ta}: {dt*1e3:.3f} ms  {2.0*m*n*k/dt*1e-12:.1f} TFLOP/s", flush=True)
bench(4096, 4096, 4096); bench(8192, 8192, 8192, reps=3)
bench(143556, 114, 114)            # X = AO . D
bench(114, 114, 143556, ta=True)   # V = B^T . AO
bench(294868, 494, 494); bench(494, 494, 294868, ta=True, reps=3)
