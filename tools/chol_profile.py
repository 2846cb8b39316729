"""Where the device-side pivoted Cholesky of the ERI spends its time: integral columns (device kernel csrc/eri_cols.hip, or
the host engine with `host` as fourth argument), the host diagonal, and the rest (residual-update GEMM, per-vector rank-1
loop with one host sync each).  Anthracene, def2-TZVP-shaped basis by default.  The device columns are timed with a
synchronise around each call (they are asynchronous otherwise), which costs the run a little."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_compute_dft_amd import basis, inputs, integrals, cholesky

args = sys.argv[1:]
mol, bname, tol = (args + ["Anthracene", "def2-tzvp", "1e-7"])[:3] if args else ("Anthracene", "def2-tzvp", "1e-7")
host = len(args) > 3 and args[3] == "host"
syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, mol + ".xyz"))
sh = basis.build_shells(syms, xyz, bname)
acc = {"cols": 0.0, "ncols": 0, "by_l": {}}
def wrap(cls, sync):
    orig = cls.cols
    def timed(self, C, D, *a, **k):
        if sync: torch.cuda.synchronize()
        t0 = time.perf_counter(); r = orig(self, C, D, *a, **k)
        if sync: torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        acc["cols"] += dt; acc["ncols"] += 1
        key = tuple(sorted((int(sh.l[C]), int(sh.l[D]))))
        e = acc["by_l"].setdefault(key, [0, 0.0]); e[0] += 1; e[1] += dt
        return r
    cls.cols = timed
wrap(integrals.EriColumns, False); wrap(integrals.DeviceEriColumns, True)
od = integrals.EriColumns.diag
def tdiag(self):
    t0 = time.perf_counter(); r = od(self); acc["diag"] = time.perf_counter() - t0; return r
integrals.EriColumns.diag = tdiag
cholesky.cholesky_eri(basis.build_shells(*basis.parse_xyz(os.path.join(inputs.DATA_DIR, "H2O.xyz")), "def2-svp"), tol=1e-6, device="cuda:0")   # warm the libraries
acc.update(cols=0.0, ncols=0, by_l={})
torch.cuda.synchronize(); t0 = time.perf_counter()
L = cholesky.cholesky_eri(sh, tol=float(tol), device="cuda:0", device_columns=not host)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"{mol}/{bname}: nao {sh.nao}, {L.shape[0]} vectors in {tot:.2f} s; integral columns ({'host' if host else 'device'}) {acc['cols']:.2f} s in {acc['ncols']} shell-pair blocks "
      f"(+ host diagonal {acc.get('diag', 0):.2f} s); everything else (device algebra, launches, syncs) {tot - acc['cols'] - acc.get('diag', 0):.2f} s")
for k, (n, t) in sorted(acc["by_l"].items()):
    print(f"   ket (l, l') = {k}: {n:4d} blocks, {1e3 * t / n:8.2f} ms each, {t:6.2f} s")
