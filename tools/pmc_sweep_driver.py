"""The sweep at the Benzene/def2-SVP shape (nao 114) and the Anthracene/def2-TZVP shape (nao 494), through DFT_ComputeXC (dm)
and DFT_ComputeXCOcc (occupied orbitals), a few calls each: the program tools/pmc_sweep.sh runs under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_compute_dft_amd as q

dev = torch.device("cuda:0")
for xc, ngrid, nao, nocc in (("GGA", 143556, 114, 21), ("B3LYP", 294868, 494, 47)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = 0.7 * np.sqrt(2.0) * torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    s = q.DFTSolverWrapper(q.library_path(), xc)
    v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    for _ in range(8):
        s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
    for _ in range(8):
        s.compute_xc_occ(ngrid, nao, nocc, c, ao, w, v, gr, dm)
    torch.cuda.synchronize()
    del ao, gr, s
    torch.cuda.empty_cache()
