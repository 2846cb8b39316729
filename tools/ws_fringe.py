"""What the fringe tile costs: the wave-specialised kernels at nao 96 / 112 (whole tiles) against 114 (Benzene:
7 tiles + 2 columns, padded to 8) and 128, same grid.  Per-kernel HIP-event medians, interleaved rounds."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_compute_dft_amd as q
import bench

ngrid = 143556
naos = [int(a) for a in sys.argv[1:]] or [96, 112, 114, 128]
dev = torch.device("cuda:0")
cases = {}
for nao in naos:
    dm, ao, gr, w = bench.synth(ngrid, nao, True, dev, bench.SEED)
    cases[nao] = (dm, ao, gr, w, torch.zeros((nao, nao), dtype=torch.float64, device=dev), q.DFTSolverWrapper(q.library_path(), "GGA"))
res = {n: {} for n in naos}
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:
    dm, ao, gr, w, d_v, s = cases[naos[0]]
    s.compute_xc(ngrid, naos[0], dm, ao, w, d_v, gr)
for r in range(5):
    for nao in naos:
        dm, ao, gr, w, d_v, s = cases[nao]
        for _ in range(30):
            s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        s.set_option("profile", 1)
        for _ in range(10):
            s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        for n, ms in s.timings():
            res[nao].setdefault(n, []).append(ms)
        s.set_option("profile", 0)
for nao in naos:
    print(nao, {n: round(float(np.median(v)) * 1e3, 1) for n, v in res[nao].items()})
