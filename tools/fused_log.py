import sys; sys.path.insert(0,'/root/repo')
import torch
from quantum_compute_dft_amd import inputs, scf
dev=torch.device('cuda:0')
for fn,eri in (("GGA","dense"),("B3LYP","cholesky")):
    inp=inputs.build("Benzene","def2-svp",3,device=dev,verbose=False,eri_mode=eri,chol_tol=1e-8)
    be=scf.HipBackend(inp,fn,device=dev)
    r=scf.run_scf(inp,be,fn,log=None)
    print(fn,eri,r["E_tot"],r["cycles"],"iter_ms",round(r["iter_ms"],4),"xc",round(r["xc_ms"],4),"jk",round(r["jk_ms"],4)); print(r["tail_log"]); print(r["cycle_ms"])
