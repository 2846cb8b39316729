"""Two ranks on one GPU (gloo): the Anthracene/def2-SVP leg of bench.py outside the bench harness."""
import faulthandler, os, sys, time
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def main(rank, world, port):
    faulthandler.enable(all_threads=True)
    import torch.distributed as dist
    from quantum_compute_dft_amd import inputs, scf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    print(rank, "build", flush=True)
    inp = inputs.build("Anthracene", "def2-svp", 3, device=dev, verbose=False, eri_mode="cholesky", chol_tol=1e-8, rank=rank, world=world)
    print(rank, "backend", flush=True)
    be = scf.HipBackend(inp, "B3LYP", rank=rank, world=world, device=dev)
    print(rank, "scf", be.fused, be.tail is not None, flush=True)
    r = scf.run_scf(inp, be, "B3LYP", log=None)
    print(rank, r["E_tot"], r["cycles"], r["iter_ms"], flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.spawn(main, args=(2, 29533), nprocs=2, join=True)
