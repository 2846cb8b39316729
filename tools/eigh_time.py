"""Host vs device Fock diagonalisation at the BASELINE basis sizes (the SCF cycle's host part)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from scipy.linalg import eigh
from quantum_compute_dft_amd.hostinfo import blas_threads, host_cpu_share
print("cpu share", host_cpu_share(), flush=True)
def med(f, n=7):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
with blas_threads():
    for n in (114, 246, 494, 1150):
        rng = np.random.default_rng(n)
        A = rng.normal(size=(n, n)); F = 0.5 * (A + A.T)
        B = rng.normal(size=(n, n)) * 0.3 / np.sqrt(n); S = np.eye(n) + 0.5 * (B + B.T)
        s, U = np.linalg.eigh(S); X = U / np.sqrt(s)
        t_gen = med(lambda: eigh(F, S))
        t_evd = med(lambda: eigh(F, S, driver="gvd"))
        def orth():
            e, C = np.linalg.eigh(X.T @ F @ X); return e, X @ C
        t_orth = med(orth)
        Fd = torch.as_tensor(F, device="cuda"); Xd = torch.as_tensor(X, device="cuda")
        def dev():
            e, C = torch.linalg.eigh(Xd.T @ Fd @ Xd); C = Xd @ C; torch.cuda.synchronize()
        dev(); t_dev = med(dev)
        e0, _ = eigh(F, S); e1, _ = orth()
        print(f"n={n:5d}  scipy eigh(F,S) {t_gen:8.2f} ms   gvd {t_evd:8.2f} ms   X^T F X + numpy eigh {t_orth:8.2f} ms   torch/hipSOLVER on device {t_dev:8.2f} ms   (max |de| {np.abs(e0-e1).max():.1e})", flush=True)
