"""Factorised J/K (DFT_ComputeJKFactorized) timings on synthetic Cholesky vectors of the BASELINE shapes."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.hostinfo import blas_threads
_pin = blas_threads(); _pin.__enter__()   # host pools on the CPU share (hostinfo.py): no quota-throttling stalls in the timings
dev = torch.device('cuda:0')
cases = [("benzene def2-SVP", 114, 21, 900), ("anthracene def2-SVP", 246, 47, 1900), ("anthracene def2-TZVP", 494, 47, 3000),
         ("C33 def2-SVP (1/8 of the vectors)", 1150, 250, 1000)]
if len(sys.argv) > 1:
    cases = [c for c in cases if sys.argv[1] in c[0]]
for name, nao, nocc, naux in cases:
    g = torch.Generator(device=dev); g.manual_seed(1)
    L = torch.randn((naux, nao, nao), dtype=torch.float64, device=dev, generator=g) * 0.1
    L[:64] = 0.5 * (L[:64] + L[:64].transpose(1, 2))   # the checked slice is symmetric like real vectors
    c = torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = c @ c.T
    J = torch.zeros((nao, nao), dtype=torch.float64, device=dev); K = torch.zeros_like(J)
    s = q.DFTSolverWrapper(q.library_path(), 'B3LYP')
    s.set_option("profile", 1)
    for _ in range(2): s.compute_jk_factorized(nao, naux, nocc, L, dm, c, J, K)
    torch.cuda.synchronize()
    acc = {}
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        s.compute_jk_factorized(nao, naux, nocc, L, dm, c, J, K)
        torch.cuda.synchronize()
        for k, v in s.timings(): acc[k] = acc.get(k, 0.0) + v / R
    wall = (time.perf_counter() - t0) / R
    fl = 2.0 * naux * nao * nao * nocc
    lb = 8.0 * naux * nao * nao
    print(f"{name}: nao={nao} nocc={nocc} naux={naux} L={lb/1e9:.2f} GB  wall {wall*1e3:.3f} ms", flush=True)
    print(f"   J pass (v_P L_P; L:D rides in the half transform): {acc['cd_j']:.3f} ms = {lb/acc['cd_j']/1e6:.0f} GB/s", flush=True)
    print(f"   half transform     : {acc['cd_half']:.3f} ms = {fl/acc['cd_half']/1e9:.1f} TFLOP/s (useful), L read {lb/acc['cd_half']/1e6:.0f} GB/s", flush=True)
    print(f"   K = Yt^T Yt        : {acc['cd_k']:.3f} ms = {fl/acc['cd_k']/1e9:.1f} TFLOP/s counted as the full square ({0.5*fl*(nao+1)/nao/acc['cd_k']/1e9:.1f} as a symmetric rank-k update)", flush=True)
    # reference check on a slice of vectors
    Y = torch.matmul(L[:64], c)
    Kr = torch.einsum('pmi,pni->mn', Y, Y)
    s.compute_jk_factorized(nao, 64, nocc, L, dm, c, None, K); torch.cuda.synchronize()
    print(f"   check vs torch (64 vectors): max rel err {float((K-Kr).abs().max()/Kr.abs().max()):.2e}", flush=True)
    del L, s
    torch.cuda.empty_cache()
