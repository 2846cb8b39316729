import sys, time, math, numpy as np, torch
sys.path.insert(0, '.')
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.hostinfo import blas_threads
_pin = blas_threads(); _pin.__enter__()   # host pools on the CPU share (hostinfo.py): no quota-throttling stalls in the timings
from quantum_compute_dft_amd import basis
dev = torch.device('cuda:0')
ring = []
for i in range(6):
    a = math.pi / 3 * i
    ring.append(f"C {1.397*math.cos(a):.6f} {1.397*math.sin(a):.6f} 0.0"); ring.append(f"H {2.481*math.cos(a):.6f} {2.481*math.sin(a):.6f} 0.0")
syms, xyz = basis.parse_xyz("; ".join(ring))
from quantum_compute_dft_amd import grid_gen
g = grid_gen.Grids(syms, xyz, 3, device='cuda')
for bname in ("sto-3g", "def2-svp"):
    sh = basis.build_shells(syms, xyz, bname)
    ngrid = g.size
    coords = torch.as_tensor(g.coords, device=dev)
    ao = torch.empty((ngrid, sh.nao), dtype=torch.float64, device=dev); gr = torch.empty((3, ngrid, sh.nao), dtype=torch.float64, device=dev)
    for pt in (0, 16, 8):
      s = q.DFTSolverWrapper(sys.argv[1] if len(sys.argv) > 1 else q.library_path(), 'GGA'); s.set_option('ao_pt', pt)
      for deriv, gg in ((0, None), (1, gr)):
        t_spin = time.perf_counter() + 0.08   # GPU clock ramp: ~40 ms of sustained load before the clocks settle
        while time.perf_counter() < t_spin: s.eval_ao(sh, coords, ngrid, ao, gg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): s.eval_ao(sh, coords, ngrid, ao, gg)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        byts = ngrid * (8 * sh.nao * (4 if deriv else 1) + 24)
        print(f"eval_ao benzene (real Becke grid {ngrid}) {bname:8s} nao={sh.nao:3d} deriv={deriv} points/WG={pt:2d}: {dt*1e6:8.1f} us  {byts/dt/1e9:7.0f} GB/s", flush=True)
