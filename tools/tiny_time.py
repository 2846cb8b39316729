"""Synchronous DFT_ComputeXC at small bases: the one-pass kernel (option tiny, csrc/xc_tiny_kernels.hpp) against the
four-launch path, as plain launches and as a recorded graph; wall time per call and the per-kernel event times.
usage: python tools/tiny_time.py [h2o h2o_gga h2o_b3lyp nh3 big_grid ...]
       python tools/tiny_time.py scan      (default graph option, a grid of sizes: where the one-pass kernel pays)
       python tools/tiny_time.py band      (the same, 17-32 functions around one sub-tile per wave slot)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_compute_dft_amd as q

SHAPES = {"h2o": ("LDA", 34310, 24), "h2o_gga": ("GGA", 34310, 24), "h2o_b3lyp": ("B3LYP", 34310, 24),
          "h2_sto3g": ("GGA", 22000, 2), "ch4_sto3g": ("GGA", 56000, 9), "nh3_gga": ("GGA", 45000, 29),
          "big_grid": ("GGA", 600000, 24), "big_grid_b3lyp": ("B3LYP", 600000, 32), "big_grid_lda16": ("LDA", 600000, 16)}
dev = torch.device("cuda:0")
if sys.argv[1:] in (["scan"], ["band"]):
    band = sys.argv[1] == "band"      # the sizes right above one sub-tile per wave slot, two column tiles only
    for xc in ("LDA", "GGA", "B3LYP"):
        for nao in ((24, 32) if band else (8, 16, 24, 32)):
            for ngrid in ((30000, 32768, 33000, 36000, 40000, 44000, 48000, 56000, 64000, 80000, 120000) if band else
                          (20000, 34310, 50000, 70000, 100000, 150000, 300000)):
                g = torch.Generator(device=dev); g.manual_seed(1)
                ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
                gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g) if xc != "LDA" else None
                w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
                c = 0.7 * np.sqrt(2.0) * torch.randn((nao, max(1, nao // 4)), dtype=torch.float64, device=dev, generator=g)
                dm = (c @ c.T).contiguous()
                res = []
                for tiny in (0, 1):
                    s = q.DFTSolverWrapper(q.library_path(), xc)
                    s.set_option("tiny", tiny)
                    v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
                    for _ in range(30):
                        s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
                    ts = []
                    for rep in range(5):
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        for _ in range(200):
                            s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
                        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200)
                    res.append(1e6 * float(np.median(ts)))
                print(f"scan {xc:6s} nao {nao:3d} ngrid {ngrid:7d}  four launches {res[0]:8.2f} us  one pass {res[1]:8.2f} us  ratio {res[1] / res[0]:.2f}", flush=True)
    sys.exit(0)
for name in (sys.argv[1:] or ["h2o", "h2o_gga", "h2o_b3lyp", "h2_sto3g", "ch4_sto3g", "nh3_gga", "big_grid", "big_grid_b3lyp", "big_grid_lda16"]):
    xc, ngrid, nao = SHAPES[name]
    g = torch.Generator(device=dev); g.manual_seed(1)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g) if xc != "LDA" else None
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = 0.7 * np.sqrt(2.0) * torch.randn((nao, max(1, nao // 4)), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    ref = None
    for tiny in (0, 1):
        for graph in (0, 1):
            s = q.DFTSolverWrapper(q.library_path(), xc)
            s.set_option("tiny", tiny); s.set_option("graph", graph)
            v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            n = 300
            for _ in range(30):
                e = s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
            ts = []
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    e = s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
            parts = ""
            if graph == 0:
                s.set_option("profile", 1)
                acc = {}
                for _ in range(20):
                    s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
                    for k, ms in s.timings():
                        acc.setdefault(k, []).append(ms)
                parts = "  " + " ".join(f"{k} {1e3 * np.median(x):.1f}" for k, x in acc.items())
            if ref is None:
                ref = (e, v.clone())
            dv = float((v - ref[1]).abs().max() / ref[1].abs().max())
            print(f"{name:15s} tiny={tiny} graph={graph} wall min {1e6 * min(ts):8.2f} us median {1e6 * np.median(ts):8.2f} us  "
                  f"exc {e:.10f} (rel {abs(e - ref[0]) / abs(ref[0]):.1e}, dV {dv:.1e}){parts}", flush=True)
    del ao, gr
    torch.cuda.empty_cache()
