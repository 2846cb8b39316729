"""DFT_ComputeXCDirect against resident planes: Benzene/def2-SVP (real shells and grid) over chunk sizes, and the
config-5 shape (synthetic shells nao 1150, 1.44 M points) where the resident planes are 53 GB."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd import basis, grid_gen, inputs

dev = torch.device("cuda:0")
f64 = torch.float64


def run(label, sh, coords, weights, fn, chunks, reps=10):
    ngrid, nao = coords.shape[0], sh.nao
    s = q.DFTSolverWrapper(q.library_path(), fn)
    g = torch.Generator(device=dev); g.manual_seed(1)
    C = 0.3 * torch.randn((nao, max(1, nao // 5)), dtype=f64, device=dev, generator=g)
    dm = (2.0 * C @ C.T).contiguous()
    d_c, d_w = torch.as_tensor(coords, dtype=f64, device=dev), torch.as_tensor(weights, dtype=f64, device=dev)
    d_v, d_e = torch.zeros((nao, nao), dtype=f64, device=dev), torch.zeros(1, dtype=f64, device=dev)
    for ch in chunks:
        for _ in range(2):
            s.compute_xc_direct(sh, ngrid, d_c, d_w, dm, d_v, d_e, ch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            s.compute_xc_direct(sh, ngrid, d_c, d_w, dm, d_v, d_e, ch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"{label}: chunk {ch if ch else 'auto':>8}: {dt*1e3:9.3f} ms per call, {ngrid/dt/1e6:8.1f} M points/s, Exc {float(d_e.item()):.10f}, "
              f"peak HBM in use {torch.cuda.max_memory_allocated()/1e9:.2f} GB (torch side)")


syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, "Benzene.xyz"))
sh = basis.build_shells(syms, xyz, "def2-svp")
gr = grid_gen.Grids(syms, xyz, level=3)
run("Benzene GGA/def2-SVP", sh, gr.coords, gr.weights, "GGA", [0, 16384, 32768, 65536, gr.size])
if len(sys.argv) > 1 and sys.argv[1] == "c33":
    syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, "C33H56N7O17P3S.xyz"))
    sh5 = basis.synthetic_shells(syms, xyz, basis.DEF2_SVP_PATTERN)          # the contraction pattern of def2-SVP, seeded exponents
    g5 = grid_gen.Grids(syms, xyz, level=3, device=dev)                       # the real level-3 grid of the molecule
    run(f"config 5 (C33H56N7O17P3S B3LYP, def2-SVP-shaped shells, nao {sh5.nao})", sh5, g5.coords, g5.weights, "B3LYP", [0, 131072], reps=3)
