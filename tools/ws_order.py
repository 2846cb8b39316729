"""Effect of the sub-tile walking order of the two contraction kernels (option sweep_order: bit 0 = rho
backwards, bit 1 = Vxc backwards) on Benzene GGA: a pass that starts where the previous one stopped
re-reads the tail of the planes from the Infinity Cache."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantum_compute_dft_amd as q
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "benzene_gga_def2svp"
xc, nao, ngrid = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
dm, ao, gr, w = bench.synth(ngrid, nao, xc != "LDA", dev, bench.SEED)
d_v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
s = q.DFTSolverWrapper(q.library_path(), xc)
s.set_option("ws_waves", 16)
ref = None
for rnd in range(2):
    for order, hot in ((0, 8), (2, 8), (0, 0), (0, 2), (0, 3), (0, 4), (2, 3), (0, 5)):
        s.set_option("sweep_order", order); s.set_option("hot8", hot); s.set_option("profile", 1)
        acc = {}; walls = []
        for r in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                exc = s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
            walls.append((time.perf_counter() - t0) / 50 * 1e6)
            if r:
                for n, ms in s.timings(): acc.setdefault(n, []).append(ms * 1e3)
        v = d_v.cpu().numpy()
        if ref is None: ref = (exc, v)
        print(f"{name} sweep_order={order} hot8={hot}: wall {np.median(walls[1:]):.1f} us", {n: round(float(np.median(x)), 1) for n, x in acc.items()},
              f"dExc {abs(exc-ref[0])/abs(ref[0]):.1e} dV {np.abs(v-ref[1]).max()/np.abs(ref[1]).max():.1e}")
