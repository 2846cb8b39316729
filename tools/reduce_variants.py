import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import quantum_compute_dft_amd as q
import bench
xc, nao, ngrid = bench.WORKLOADS["benzene_gga_def2svp"]
dev = torch.device("cuda:0")
dm, ao, gr, w = bench.synth(ngrid, nao, True, dev, bench.SEED)
d_v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
s = q.DFTSolverWrapper(q.library_path(), xc)
for ff, rv in ((1, 0), (0, 0), (1, 0), (0, 0)):
    s.set_option("fuse_finish", ff); s.set_option("profile", 1)
    acc = {}
    for r in range(6):
        for _ in range(30): s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        if r:
            for n, ms in s.timings(): acc.setdefault(n, []).append(ms * 1e3)
    s.set_option("profile", 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200 * 1e6
    print(f"fuse_finish={ff} reduce_vec={rv}: wall {wall:.1f} us", {n: round(float(np.median(x)), 1) for n, x in acc.items()})
