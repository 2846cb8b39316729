"""Half-transform time vs contraction length (stages per workgroup): fit T_round = a + b*nst."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.hostinfo import blas_threads
_pin = blas_threads(); _pin.__enter__()   # host pools on the CPU share (hostinfo.py): no quota-throttling stalls in the timings
dev = torch.device('cuda:0')
nocc = 47
for nao in (128, 256, 384, 512, 768, 1024):
    naux = max(256, int(3.0e9 / (8 * nao * nao)))
    L = torch.randn((naux, nao, nao), dtype=torch.float64, device=dev) * 0.1
    c = torch.randn((nao, nocc), dtype=torch.float64, device=dev); dm = c @ c.T
    J = torch.zeros((nao, nao), dtype=torch.float64, device=dev); K = torch.zeros_like(J)
    s = q.DFTSolverWrapper(q.library_path(), 'B3LYP'); s.set_option("profile", 1)
    out = []
    for with_j in (True, False):
        acc = {}
        for it in range(6):
            s.compute_jk_factorized(nao, naux, nocc, L, dm if with_j else None, c, J if with_j else None, K)
            torch.cuda.synchronize()
            if it >= 2:
                for k, v in s.timings(): acc[k] = acc.get(k, 0.0) + v / 4
        out.append(acc["cd_half"])
    wgs = naux * ((nao + 127) // 128); rounds = wgs / 512.0; nst = (nao + 15) // 16
    print(f"nao={nao:5d} naux={naux:5d} stages/WG={nst:3d} WGs={wgs:6d}: half+dot {out[0]*1e3:8.1f} us ({out[0]*1e3/rounds:6.2f} us/round)  half only {out[1]*1e3:8.1f} us ({out[1]*1e3/rounds:6.2f} us/round)  ideal {nst*1.44:6.1f} us/round", flush=True)
    del L, s; torch.cuda.empty_cache()
