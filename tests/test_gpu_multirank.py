"""Two ranks of the SCF driver on ONE GPU (gloo rendezvous on 127.0.0.1): grid block + Cholesky-vector
slice (or dense-ERI row block, DFT_ComputeJKRows) per rank, one all-reduce of [Vxc | J | K | Exc] per cycle,
rank 0 authoritative for DIIS + eigh (one broadcast of [dm | cocc | scalars]) -- same energies as the
single-rank run, host and device-resident forms of the loop."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, fn, eri_mode, out_dir, device_resident=False, eigensolver="auto"):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from quantum_compute_dft_amd import inputs, scf
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # Cholesky vectors: factorised on rank 0 alone, slices sent to the ranks (inputs.build(world > 1))
        inp = inputs.build("H2O", "def2-svp", 3, device="cuda:0" if eri_mode == "cholesky" else "cpu", verbose=False, eri_mode=eri_mode,
                           chol_tol=1e-10, rank=rank, world=world)
        be = scf.HipBackend(inp, fn, rank=rank, world=world, device="cuda:0", device_resident=device_resident, eigensolver=eigensolver)
        res = scf.run_scf(inp, be, fn, log=None, conv_e=1e-11, conv_dm=1e-9)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), e=res["E_tot"], exc=res["E_xc"], ex=res["E_ex_hf"],
                 dm=res["dm"], conv=res["converged"], ngrid=be.ngrid)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fn,eri_mode,device_resident,eigensolver", [
    ("B3LYP", "cholesky", False, "auto"), ("GGA", "dense", False, "auto"), ("B3LYP", "dense", False, "auto"),
    ("B3LYP", "cholesky", True, "auto"), ("GGA", "cholesky", True, "rotate"), ("B3LYP", "dense", False, "rotate"),
    # rank 0 runs the SCF tail kernels (scf._run_scf_fused with replicas), the others receive [dm | cocc | scalars]
    ("B3LYP", "cholesky", None, "rotate"), ("GGA", "dense", None, "rotate")])
def test_two_ranks_on_one_gpu_match_the_single_rank_scf(tmp_path, fn, eri_mode, device_resident, eigensolver):
    import torch.multiprocessing as mp
    from quantum_compute_dft_amd import inputs, scf
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), fn, eri_mode, str(tmp_path), device_resident, eigensolver), nprocs=world, join=True)
    inp = inputs.build("H2O", "def2-svp", 3, verbose=False, eri_mode=eri_mode, chol_tol=1e-10)
    ref = scf.run_scf(inp, scf.HipBackend(inp, fn), fn, log=None, conv_e=1e-11, conv_dm=1e-9)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert bool(r0["conv"]) and bool(r1["conv"]) and ref["converged"]
    assert int(r0["ngrid"]) + int(r1["ngrid"]) == inp.grids.size and int(r1["ngrid"]) > 0
    assert float(r0["e"]) == float(r1["e"]) and np.array_equal(r0["dm"], r1["dm"])     # replicas agree bitwise
    assert float(r0["e"]) == pytest.approx(ref["E_tot"], abs=1e-9)
    assert float(r0["exc"]) == pytest.approx(ref["E_xc"], abs=1e-9)
    assert float(r0["ex"]) == pytest.approx(ref["E_ex_hf"], abs=1e-9)
