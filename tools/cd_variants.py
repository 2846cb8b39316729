"""Half-transform variants at the Anthracene/def2-TZVP shape (nao 494, nocc 47, 3000 vectors)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import quantum_compute_dft_amd as q
dev = torch.device('cuda:0')
nao, nocc, naux = 494, 47, 3000
g = torch.Generator(device=dev); g.manual_seed(1)
L = torch.randn((naux, nao, nao), dtype=torch.float64, device=dev, generator=g) * 0.1
c = torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
dm = c @ c.T
J = torch.zeros((nao, nao), dtype=torch.float64, device=dev); K = torch.zeros_like(J)
s = q.DFTSolverWrapper(q.library_path(), 'B3LYP'); s.set_option("profile", 1)
def run(tag, wantJ, **opts):
    for k, v in opts.items(): s.set_option(k, v)
    acc = {}
    for it in range(7):
        s.compute_jk_factorized(nao, naux, nocc, L, dm, c, J if wantJ else None, K); torch.cuda.synchronize()
        if it >= 2:
            for k, v in s.timings(): acc[k] = acc.get(k, 0.0) + v / 5
    print(tag, {k: round(v, 3) for k, v in acc.items()}, flush=True)
run("J+K", True)
run("K only", False)
