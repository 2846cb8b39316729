"""Input builder: counterpart of the reference's grid.py (`build`, `get_ao_grad`, grid.py:23-67)
without PySCF: molecule + basis -> shell table, level-3 Becke/Lebedev grid, S, T, V, dense ERI,
E_nuc, electron count.  AO values / gradients are NOT built here: the driver evaluates them on the
device with DFT_EvalAO (grid.py:30,38 did it on the CPU and dft.py:155,172 uploaded them)."""
import os
from dataclasses import dataclass

import numpy as np

from . import basis, grid_gen, integrals

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@dataclass
class SCFInputs:
    symbols: list
    atom_xyz: np.ndarray   # bohr
    shells: basis.ShellTable
    grids: grid_gen.Grids
    S: np.ndarray
    T: np.ndarray
    V: np.ndarray
    Hcore: np.ndarray
    eri: np.ndarray        # (nao, nao, nao, nao), or None when only Cholesky vectors were built
    E_nuc: float
    nocc: int
    nelec: int
    chol: np.ndarray = None  # (naux, nao, nao) Cholesky vectors of the ERI (eri_mode='cholesky')
    chol_range: tuple = None  # (lo, hi, naux): `chol` is only THIS rank's slice of the naux vectors (build(..., world > 1))


def build(atom_path, basis_name="sto-3g", grid_level=3, device="cpu", verbose=True, eri_mode="dense",
          chol_tol=1e-9, rank=0, world=1, group=None):
    """grid.py:42-67.  `atom_path`: an .xyz file (or a molecule name resolved in data/).
    eri_mode "dense": the (nao^4) tensor of grid.py:65; "cholesky": pivoted Cholesky vectors only.
    world > 1 (torch.distributed initialised): the Cholesky factorisation -- the one expensive step, host integral columns
    + device algebra -- runs on rank 0 ALONE, on the whole node's CPU allowance while the other ranks wait, and every
    rank receives only its slice of the vectors (grid_shard.scatter_vectors); `chol_range` records the slice."""
    if not os.path.exists(atom_path):
        cand = os.path.join(DATA_DIR, atom_path if atom_path.endswith(".xyz") else atom_path + ".xyz")
        if os.path.exists(cand):
            atom_path = cand
    symbols, xyz = basis.parse_xyz(atom_path)
    shells = basis.build_shells(symbols, xyz, basis_name)
    nelec = sum(basis.atomic_number(s) for s in symbols)
    if nelec % 2:
        raise ValueError("closed-shell (RKS) only: odd electron count")
    nocc = nelec // 2
    if verbose:  # grid.py:54-56,60
        print(f"Number of basis functions: {shells.nao}")
        print(f"Number of electrons: {nelec}")
        print(f"Number of occupied orbitals: {nocc}")
    grids = grid_gen.Grids(symbols, xyz, level=grid_level, device=device)
    if verbose:
        print(f"Number of grid points for integration: {grids.size}")
    S, T, V = integrals.int1e(shells, symbols, xyz)
    eri = chol = chol_range = None
    if eri_mode == "dense":
        eri = integrals.int2e(shells)
    elif eri_mode == "cholesky":
        from .cholesky import cholesky_eri
        import time
        t0 = time.time()
        if world > 1:
            import torch.distributed as dist
            from .grid_shard import scatter_vectors, vector_bounds
            from .hostinfo import host_cpu_share
            full = None
            if rank == 0:
                integrals.set_threads(host_cpu_share(whole_node=True))    # the other ranks are waiting in the scatter below
                try:
                    full = cholesky_eri(shells, tol=chol_tol, device=device)
                finally:
                    integrals.set_threads(host_cpu_share())
            chol, naux = scatter_vectors(full, shells.nao, device, world, rank, group)
            del full
            chol_range = (*vector_bounds(naux, world, rank), naux)
            if not str(device).startswith("cuda"):
                chol = chol.numpy()
        else:
            chol = cholesky_eri(shells, tol=chol_tol, device=device)   # on a GPU: the factorisation's algebra and the vectors stay there
        if verbose:
            where = "integral columns (DFT_EriColumns), algebra and vectors on the device" if str(device).startswith("cuda") else "on the host"
            print(f"Cholesky vectors of the ERI: {chol.shape[0]} (threshold {chol_tol:g}, {time.time() - t0:.1f} s; {where})")
    else:
        raise ValueError(f"eri_mode {eri_mode!r}: expected 'dense' or 'cholesky'")
    return SCFInputs(symbols, xyz, shells, grids, S, T, V, T + V, eri,
                     integrals.energy_nuc(symbols, xyz), nocc, nelec, chol, chol_range)
