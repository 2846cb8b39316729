"""A/B of the eight- and sixteen-wave wave-specialised kernels on one workload, interleaved rounds in ONE
process (cdna_hip_programming.md rule 24): per-kernel HIP-event times and the wall time per call."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_compute_dft_amd as q
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "benzene_gga_def2svp"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
xc, nao, ngrid = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
dm, ao, gr, w = bench.synth(ngrid, nao, xc != "LDA", dev, bench.SEED)
d_v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
sol = {k: q.DFTSolverWrapper(q.library_path(), xc) for k in (8, 16)}
for k, s in sol.items():
    s.set_option("ws_waves", k)
res = {k: {"wall": [], "kern": {}} for k in sol}
exc = {}
t_end = time.perf_counter() + 0.1
while time.perf_counter() < t_end:            # clock spin-up
    sol[8].compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
for r in range(rounds):
    for k, s in sol.items():
        for _ in range(20):
            s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            exc[k] = s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        torch.cuda.synchronize(); res[k]["wall"].append((time.perf_counter() - t0) / 100 * 1e3)
        s.set_option("profile", 1)
        for _ in range(10):
            s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
        for n, ms in s.timings():
            res[k]["kern"].setdefault(n, []).append(ms)
        s.set_option("profile", 0)
for k in sol:
    kk = {n: round(float(np.median(v)) * 1e3, 1) for n, v in res[k]["kern"].items()}
    print(f"{name} ws_waves={k}: wall median {np.median(res[k]['wall'])*1e3:.1f} us  min {min(res[k]['wall'])*1e3:.1f} us  kernels(us) {kk}  exc {exc[k]!r}")
print("rel diff of Exc between the two paths:", abs(exc[8] - exc[16]) / abs(exc[8]))
