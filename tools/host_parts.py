"""Host-side pieces of one SCF cycle at n = 494 (Anthracene/def2-TZVP) on synthetic matrices."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from scipy.linalg import eigh
from quantum_compute_dft_amd.hostinfo import blas_threads
from quantum_compute_dft_amd.scf import CDIIS, FockDiagonaliser
def med(f, n=7):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
with blas_threads():
    for n, nocc in ((246, 47), (494, 47)):
        rng = np.random.default_rng(n)
        A = rng.normal(size=(n, n)); F = 0.5 * (A + A.T)
        B = rng.normal(size=(n, n)) * 0.3 / np.sqrt(n); S = np.eye(n) + 0.5 * (B + B.T)
        C = np.linalg.qr(rng.normal(size=(n, nocc)))[0]; dm = 2 * C @ C.T
        d = CDIIS()
        for _ in range(8): d.update(S, dm, F + 1e-3 * rng.normal(size=(n, n)))
        t_diis = med(lambda: d.update(S, dm, F))
        dd = CDIIS(device=torch.device("cuda"))
        for _ in range(10): dd.update(S, dm, F + 1e-3 * rng.normal(size=(n, n)))
        t_diis_dev = med(lambda: dd.update(S, dm, F))
        t_full = med(lambda: eigh(F, S))
        t_sub = med(lambda: eigh(F, S, subset_by_index=[0, nocc - 1]))
        fd = FockDiagonaliser(S, torch.device("cuda"), device_from=0)
        fd(F); t_dev = med(lambda: fd(F))
        t_dm = med(lambda: 2.0 * C @ C.T)
        t_en = med(lambda: (float(np.sum(dm * F)), float(np.sum(dm * S)), float(np.sum(dm * F))))
        t_fock = med(lambda: F + S + 0.5 * (F + F.T) - 0.1 * S)
        Fd = torch.as_tensor(F, device="cuda")
        t_d2h = med(lambda: (Fd.cpu().numpy(), Fd.cpu().numpy(), Fd.cpu().numpy()))
        print(f"n={n}: DIIS update {t_diis:.2f} ms (device {t_diis_dev:.2f}) | eigh host full {t_full:.2f}, host lowest-{nocc} {t_sub:.2f}, device {t_dev:.2f} ms | dm build {t_dm:.2f} | energies {t_en:.2f} | Fock sum {t_fock:.2f} | 3 D2H {t_d2h:.2f} ms", flush=True)
