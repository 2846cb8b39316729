"""Phases of the SCF tail's rotation kernel (wall-clock stamps written by the kernel) on a Benzene-sized synthetic cycle.
usage: python tools/tail_time.py [n no]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd import scf_tail
from scipy.linalg import eigh

n, no = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (114, 21)
rng = np.random.default_rng(1)
B = 0.1 * rng.standard_normal((n, n)); S = np.eye(n) + 0.5 * (B + B.T) / np.sqrt(n)
s, V = np.linalg.eigh(S); X = V / np.sqrt(s)
lev = np.concatenate([np.sort(rng.uniform(-10, -0.5, no)), np.sort(rng.uniform(0.2, 4, n - no))])
Q = np.linalg.qr(rng.standard_normal((n, n)))[0]; Xi = np.linalg.inv(X)
H = Xi.T @ (Q * lev) @ Q.T @ Xi; H = 0.5 * (H + H.T)
dev = torch.device("cuda:0")
lib = q.load_library(q.library_path())
lib.DFT_ScfTailStamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
tail = scf_tail.ScfTail(lib, H, S, no, dev)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
e0, Cp = eigh(X.T @ H @ X); U0 = X @ Cp
tail.basis.copy_(t(U0)); cocc = np.sqrt(2) * U0[:, :no]
d_dm, d_cocc = t(cocc @ cocc.T), t(cocc)
sym = lambda a: 0.5 * (a + a.T)
names = ["start", "A->LDS,K0", "gemm Q,B", "gemm R", "max+update", "loop end", "Fo,P", "chol+inv", "G", "jacobi", "basis"]
for cyc in range(6):
    scale = 0.02 * 0.3 ** cyc
    J, K, Vx = t(scale * sym(rng.standard_normal((n, n)))), t(scale * sym(rng.standard_normal((n, n)))), t(scale * rng.standard_normal((n, n)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tail.step(True, 0.2, 1e-10, J, K, Vx, d_dm, d_cocc)
    o = tail.wait(); wall = time.perf_counter() - t0
    st = (ctypes.c_longlong * 16)(); lib.DFT_ScfTailStamps(tail._h, st)
    if n > 128 or no > 32:      # the memory-resident rotation runs its fixed-point steps as launches of their own: no phase stamps
        print(f"cycle {cyc}: status {o[4]} steps {o[5]} sweeps {o[6]} wall {1e6 * wall:.0f} us (whole DFT_ScfTailStep: eleven launches + three per fixed-point step)", flush=True)
        continue
    d = [(st[k] - st[k - 1]) / 100.0 for k in range(1, 11)]
    print(f"cycle {cyc}: status {o[4]} steps {o[5]} sweeps {o[6]} wall {1e6 * wall:.0f} us; rot phases (us): " + ", ".join(f"{nm} {x:.1f}" for nm, x in zip(names[1:], d))
          + f"; core clock {(st[12] - st[11]) / max(1, st[10] - st[0]) * 100:.0f} MHz", flush=True)
