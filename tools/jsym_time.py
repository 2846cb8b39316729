"""DFT_ComputeCoulomb on a symmetric dense ERI: the full pass against the upper-triangle pass (option eri_symmetric)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantum_compute_dft_amd as q
dev = torch.device("cuda:0")
for n in (114, 80):
    N2 = n * n
    g = torch.Generator(device=dev); g.manual_seed(1)
    A = torch.randn((N2, N2), dtype=torch.float64, device=dev, generator=g)
    E = (A + A.T).contiguous(); del A
    d = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g); d = (d + d.T).contiguous()
    J = torch.zeros((n, n), dtype=torch.float64, device=dev)
    s = q.DFTSolverWrapper(q.library_path(), "GGA")
    for opt in (0, 1, 0, 1):
        s.set_option("eri_symmetric", opt)
        for _ in range(20): s.compute_coulomb(n, E, d, J)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): s.compute_coulomb(n, E, d, J)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        gb = 8.0 * N2 * N2 * (0.5 if opt else 1.0) / 1e9
        print(f"nao {n}: eri_symmetric={opt}: {1e6 * dt:.1f} us per J = {gb / dt / 1e3:.2f} TB/s of the bytes it needs", flush=True)
