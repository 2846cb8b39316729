"""Times every hot-path row on the BASELINE shapes (synthetic data), one line each."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.hostinfo import blas_threads
_pin = blas_threads(); _pin.__enter__()   # host pools on the CPU share (hostinfo.py): no quota-throttling stalls in the timings
from quantum_compute_dft_amd import basis
from bench import synth
dev = torch.device('cuda:0')

def time_xc(name, xc, nao, ngrid, reps=10):
    dm, ao, gr, w = synth(ngrid, nao, xc != 'LDA', dev, 1)
    s = q.DFTSolverWrapper(q.library_path(), xc)
    v = torch.zeros(nao * nao, dtype=torch.float64, device=dev)
    for _ in range(2): s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    s.set_option('profile', 1); s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
    k = {n: round(ms * 1e3, 1) for n, ms in s.timings()}
    c = 1 if xc == 'LDA' else 4
    fl = 4.0 * ngrid * nao * nao
    print(f"{name:34s} nao={nao:4d} ngrid={ngrid:7d}: {dt*1e3:8.3f} ms/call  {ngrid/dt/1e6:8.1f} Mpts/s  {fl/dt/1e12:5.1f} TF  kernels(us)={k}", flush=True)
    del dm, ao, gr, w; torch.cuda.empty_cache()

time_xc("H2O LDA def2-SVP", "LDA", 24, 34310)
time_xc("H2O LDA sto-3g", "LDA", 7, 34310)
time_xc("Benzene GGA sto-3g (as shipped)", "GGA", 36, 143556)
time_xc("Benzene GGA def2-SVP", "GGA", 114, 143556)
time_xc("Benzene B3LYP def2-SVP", "B3LYP", 114, 143556)
time_xc("Anthracene B3LYP sto-3g", "B3LYP", 80, 294868)
time_xc("Anthracene B3LYP def2-SVP", "B3LYP", 246, 294868, reps=3)
time_xc("Anthracene B3LYP def2-TZVP", "B3LYP", 494, 294868, reps=3)
if "--config5" in sys.argv:
    time_xc("C33H56N7O17P3S B3LYP def2-SVP (1 GPU)", "B3LYP", 1150, 1436406, reps=2)
    sys.exit(0)

# J / K on the dense ERI
for n in (36, 80, 114):
    N2 = n * n
    eri = torch.randn((N2, N2), dtype=torch.float64, device=dev); dm = torch.randn((n, n), dtype=torch.float64, device=dev)
    J = torch.zeros_like(dm); K = torch.zeros_like(dm)
    s = q.DFTSolverWrapper(q.library_path(), 'B3LYP')
    for fn, nm in ((lambda: s.compute_coulomb(n, eri, dm, J), 'J'), (lambda: s.compute_exchange(n, eri, dm, K), 'K'), (lambda: s.compute_jk(n, eri, dm, J, K), 'J+K one pass')):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"dense ERI nao={n:3d} ({8*N2*N2/1e6:7.1f} MB) {nm:13s}: {dt*1e6:8.1f} us  {8*N2*N2/dt/1e9:7.0f} GB/s", flush=True)
    del eri; torch.cuda.empty_cache()

# AO on grid (benzene def2-SVP shells)
ring = []
import math
for i in range(6):
    a = math.pi / 3 * i
    ring.append(f"C {1.397*math.cos(a):.6f} {1.397*math.sin(a):.6f} 0.0"); ring.append(f"H {2.481*math.cos(a):.6f} {2.481*math.sin(a):.6f} 0.0")
syms, xyz = basis.parse_xyz("; ".join(ring))
for bname in ("sto-3g", "def2-svp"):
    sh = basis.build_shells(syms, xyz, bname)
    ngrid = 143556
    coords = torch.as_tensor(np.random.default_rng(0).normal(0, 3.0, (ngrid, 3)), device=dev)
    ao = torch.empty((ngrid, sh.nao), dtype=torch.float64, device=dev); gr = torch.empty((3, ngrid, sh.nao), dtype=torch.float64, device=dev)
    s = q.DFTSolverWrapper(q.library_path(), 'GGA')
    for deriv, g in ((0, None), (1, gr)):
        s.eval_ao(sh, coords, ngrid, ao, g); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): s.eval_ao(sh, coords, ngrid, ao, g)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        byts = ngrid * (8 * sh.nao * (4 if deriv else 1) + 24)
        print(f"eval_ao benzene {bname:8s} nao={sh.nao:3d} deriv={deriv}: {dt*1e6:8.1f} us  {byts/dt/1e9:7.0f} GB/s (algorithmic)", flush=True)
