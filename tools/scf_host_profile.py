"""Where the host part of a Benzene-size SCF cycle goes: the driver's host loop with timers around its parts
(set_state, device work + download, DIIS, eigen-solver, density, energy sums).  Median us per cycle after the first."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantum_compute_dft_amd import inputs, scf

mol, basis_name, fn = (sys.argv[1:] + ["Benzene", "def2-svp", "GGA"])[:3] if len(sys.argv) > 1 else ("Benzene", "def2-svp", "GGA")
inp = inputs.build(mol, basis_name, 3, verbose=False, eri_mode="cholesky", chol_tol=1e-8)
be = scf.HipBackend(inp, fn)
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc.setdefault(label, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
timed(be, "set_state", "set_state (pinned upload)")
timed(be, "fock_parts", "device work + pinned download")
if be.occ_solver is not None:
    timed(be.occ_solver, "occupied", "eigen: rotation / full")
else:
    timed(be, "eigh", "eigen: full")
timed(scf.CDIIS, "update", "DIIS")
t0 = time.perf_counter()
r = scf.run_scf(inp, be, fn, log=None)
print(f"{mol} {fn}/{basis_name}: {r['cycles']} cycles, E = {r['E_tot']:.8f}, median cycle {r['iter_ms']*1e3:.0f} us")
tot = 0.0
for k, v in acc.items():
    m = float(np.median(v[1:])) * 1e6; tot += m
    print(f"  {k:34s} {m:8.1f} us   (calls {len(v)})")
print(f"  {'rest (Fock, density, sums, logging)':34s} {r['iter_ms']*1e3 - tot:8.1f} us")
