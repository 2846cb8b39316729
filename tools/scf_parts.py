"""Where one SCF cycle's wall time goes (host side included).  usage: scf_parts.py B3LYP Anthracene def2-svp [cholesky]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from scipy.linalg import eigh
from quantum_compute_dft_amd import inputs, scf
fn, molname, bname = sys.argv[1], sys.argv[2], sys.argv[3]
mode = sys.argv[4] if len(sys.argv) > 4 else "dense"
inp = inputs.build(molname, bname, 3, device="cuda", verbose=False, eri_mode=mode, chol_tol=1e-8)
be = scf.HipBackend(inp, fn)
e, C = eigh(inp.Hcore, inp.S); nocc = inp.nocc
dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T
diis = scf.CDIIS(device=be.diis_device)
import torch
from quantum_compute_dft_amd.hostinfo import blas_threads
pin = blas_threads(1 if inp.S.shape[0] < 400 else None); pin.__enter__()
for it in range(8):
    t = [time.perf_counter()]
    be.set_dm(dm); be.set_cocc(np.sqrt(2.0) * C[:, :nocc]); torch.cuda.synchronize(); t.append(time.perf_counter())
    J, K = be.jk(fn == "B3LYP"); t.append(time.perf_counter())
    exc, V, _ = be.xc(); t.append(time.perf_counter())
    F = inp.Hcore + J + 0.5 * (V + V.T) - (0.1 * K if K is not None else 0.0); t.append(time.perf_counter())
    F = diis.update(inp.S, dm, F); t.append(time.perf_counter())
    e, C = be.eigh(F); t.append(time.perf_counter())
    dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T; t.append(time.perf_counter())
    names = ["h2d", "jk+d2h", "xc+d2h", "fock", "diis", "eigh", "dm"]
    print(it, " ".join(f"{n}={1e3*(b-a):.2f}" for n, a, b in zip(names, t[:-1], t[1:])), f"total={1e3*(t[-1]-t[0]):.2f} ms", flush=True)
