"""Ablations of the sixteen-wave kernels (option dbg: 1 = plane loads dropped by the range check,
2 = MFMAs skipped): which role sets the pace.  Results are wrong by construction; only times matter."""
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
for waves in (16, 8):
    s.set_option("ws_waves", waves)
    for dbg in ((0, 1, 32, 8, 9, 64, 65, 72, 73) if waves == 16 else (0,)):
        s.set_option("dbg", dbg); s.set_option("profile", 1)
        acc = {}
        for r in range(6):
            for _ in range(30):
                try: s.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)
                except RuntimeError: pass
            if r:
                for n, ms in s.timings(): acc.setdefault(n, []).append(ms * 1e3)
        print(f"{name} waves={waves} dbg={dbg}:", {n: round(float(np.median(v)), 1) for n, v in acc.items()})
