"""Where the device-side pivoted Cholesky of the ERI spends its time: host integral columns, the residual-update GEMM,
the per-vector rank-1 loop (device launches + one host sync each).  Anthracene, def2-TZVP-shaped basis by default."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_compute_dft_amd import basis, inputs, integrals, cholesky

mol, bname, tol = (sys.argv[1:] + ["Anthracene", "def2-tzvp", "1e-7"])[:3] if len(sys.argv) > 1 else ("Anthracene", "def2-tzvp", "1e-7")
syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, mol + ".xyz"))
sh = basis.build_shells(syms, xyz, bname)
acc = {"cols": 0.0, "ncols": 0}
orig = integrals.EriColumns.cols
def timed(self, *a, **k):
    t0 = time.perf_counter(); r = orig(self, *a, **k); acc["cols"] += time.perf_counter() - t0; acc["ncols"] += 1; return r
integrals.EriColumns.cols = timed
od = integrals.EriColumns.diag
def tdiag(self):
    t0 = time.perf_counter(); r = od(self); acc["diag"] = time.perf_counter() - t0; return r
integrals.EriColumns.diag = tdiag
torch.cuda.synchronize(); t0 = time.perf_counter()
L = cholesky.cholesky_eri(sh, tol=float(tol), device="cuda:0")
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"{mol}/{bname}: nao {sh.nao}, {L.shape[0]} vectors in {tot:.2f} s; host integral columns {acc['cols']:.2f} s in {acc['ncols']} shell-pair blocks "
      f"(+ diagonal {acc.get('diag', 0):.2f} s); everything else (device algebra, launches, syncs) {tot - acc['cols'] - acc.get('diag', 0):.2f} s")
