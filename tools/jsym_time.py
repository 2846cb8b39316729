"""DFT_ComputeCoulomb on a symmetric dense ERI: the full pass against the upper-triangle pass (option eri_symmetric)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantum_compute_dft_amd as q
dev = torch.device("cuda:0")
for n in (114, 80):
    N2 = n * n
    g = torch.Generator(device=dev); g.manual_seed(1)
    npk = n * (n + 1) // 2
    G = torch.randn((npk, npk), dtype=torch.float64, device=dev, generator=g); G = G + G.T
    ii, jj = torch.meshgrid(torch.arange(n, device=dev), torch.arange(n, device=dev), indexing="ij")
    a, c = torch.maximum(ii, jj), torch.minimum(ii, jj)
    P = (a * (a + 1) // 2 + c).reshape(-1)
    E = G[P][:, P].contiguous(); del G                            # eight-fold symmetric
    d = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g); d = (d + d.T).contiguous()
    J = torch.zeros((n, n), dtype=torch.float64, device=dev)
    s = q.DFTSolverWrapper(q.library_path(), "GGA")
    for opt in (0, 1, 2, 0, 1, 2):
        s.set_option("eri_symmetric", opt)
        for _ in range(20): s.compute_coulomb(n, E, d, J)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): s.compute_coulomb(n, E, d, J)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        gb = 8.0 * N2 * N2 * (1.0, 0.5, 0.125)[opt] / 1e9
        print(f"nao {n}: eri_symmetric={opt}: {1e6 * dt:.1f} us per J = {gb / dt / 1e3:.2f} TB/s of the bytes it needs", flush=True)
