"""Synchronous DFT_ComputeXC as plain launches against the recorded HIP graph (option "graph"), wall time per call.
usage: python tools/graph_time.py [h2o h2o_gga benzene ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantum_compute_dft_amd as q

SHAPES = {"h2o": ("LDA", 34310, 24, 5), "h2o_gga": ("GGA", 34310, 24, 5), "h2o_b3lyp": ("B3LYP", 34310, 24, 5),
          "benzene": ("GGA", 143556, 114, 21), "benzene_sto3g": ("GGA", 143556, 36, 21), "anthracene": ("B3LYP", 294868, 494, 47)}
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["h2o", "h2o_gga", "benzene_sto3g", "benzene"]):
    xc, ngrid, nao, nocc = SHAPES[name]
    g = torch.Generator(device=dev); g.manual_seed(1)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g) if xc != "LDA" else None
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = 0.7 * np.sqrt(2.0) * torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    s = q.DFTSolverWrapper(q.library_path(), xc)
    out = {}
    for label, call in (("dm ", lambda v: s.compute_xc(ngrid, nao, dm, ao, w, v, gr)),
                        ("occ", lambda v: s.compute_xc_occ(ngrid, nao, nocc, c, ao, w, v, gr, dm))):
        for graph in (0, 1, 0, 1):
            s.set_option("graph", graph)
            v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            n = 300 if ngrid < 200000 else 10
            for _ in range(30):
                e = call(v)
            ts = []
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    e = call(v)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
            out[(label, graph)] = (e, v.clone())
            print(f"{name:15s} {label} graph={graph} wall min {1e6 * min(ts):8.2f} us median {1e6 * np.median(ts):8.2f} us  exc {e:.12f}", flush=True)
        e0, v0 = out[(label, 0)]; e1, v1 = out[(label, 1)]
        print(f"{name:15s} {label} replay identical to launches: exc {e0 == e1}, vxc {bool(torch.equal(v0, v1))}", flush=True)
    del ao, gr, s
    torch.cuda.empty_cache()
