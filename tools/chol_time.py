"""Pivoted Cholesky of the ERI with the integral columns from the device kernel (csrc/eri_cols.hip) against the host engine
(csrc/integrals.c through pinned memory): wall time of the whole factorisation.  usage: chol_time.py [molecule basis tol]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_compute_dft_amd import basis, inputs, cholesky
mol, bname, tol = (sys.argv[1:] + ["Anthracene", "def2-tzvp", "1e-7"])[:3] if len(sys.argv) > 1 else ("Anthracene", "def2-tzvp", "1e-7")
syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, mol + ".xyz"))
sh = basis.build_shells(syms, xyz, bname)
for devcols in (True, False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L = cholesky.cholesky_eri(sh, tol=float(tol), device="cuda:0", device_columns=devcols)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"{mol}/{bname}: nao {sh.nao}, {L.shape[0]} vectors (tol {tol}) in {tot:.2f} s with the integral columns on the {'device' if devcols else 'host'}", flush=True)
    del L; torch.cuda.empty_cache()
