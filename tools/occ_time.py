"""Per-kernel times (HIP events of the library) of DFT_ComputeXC (dm) against DFT_ComputeXCOcc (occupied orbitals) at the
BASELINE shapes; synthetic planes (SURVEY 8(d) recipe).  usage: python tools/occ_time.py [benzene|anthracene|c33 ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_compute_dft_amd as q

SHAPES = {"benzene": ("GGA", 143556, 114, 21), "anthracene": ("B3LYP", 294868, 494, 47), "anthracene_svp": ("B3LYP", 294868, 246, 47),
          "c33": ("B3LYP", 400000, 1150, 250), "h2o": ("LDA", 34310, 24, 5), "benzene_b3lyp": ("B3LYP", 143556, 114, 21)}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["benzene", "anthracene", "c33"]):
    xc, ngrid, nao, nocc = SHAPES[name]
    g = torch.Generator(device=dev); g.manual_seed(1)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g) if xc != "LDA" else None
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = 0.7 * np.sqrt(2.0) * torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    s = q.DFTSolverWrapper(q.library_path(), xc)
    v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    for label, call in (("dm ", lambda: s.compute_xc(ngrid, nao, dm, ao, w, v, gr)),
                        ("occ", lambda: s.compute_xc_occ(ngrid, nao, nocc, c, ao, w, v, gr, dm))):
        s.set_option("profile", 0)
        for _ in range(30 if ngrid < 200000 else 5):
            e = call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 100 if ngrid < 200000 else 10
        for _ in range(n):
            e = call()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n
        s.set_option("profile", 1)
        acc = {}
        for i in range(20 if ngrid < 200000 else 5):
            call()
            for k, ms in s.timings():
                acc.setdefault(k, []).append(ms)
        print(f"{name:15s} {label} wall {1e3 * wall:8.4f} ms  exc {e:.12f}  " + "  ".join(f"{k} {1e3 * np.median(x):.1f}us" for k, x in acc.items()), flush=True)
    del ao, gr, s
    torch.cuda.empty_cache()
