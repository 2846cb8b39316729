"""`python -m quantum_compute_dft_amd.dft <LDA|GGA|B3LYP> <Molecule>` -- the reference's driver
surface (dft.py:101-297: same positionals, same printed lines) on the MI355X engine.
Extra flags (defaults = what the reference hard-codes): --basis sto-3g, --grid-level 3, --quirks 1."""
import argparse
import importlib.util
import os
import sys
import time

from . import inputs, scf


def main(argv=None):
    p = argparse.ArgumentParser(description="Run DFT (LDA/GGA/B3LYP) using the MI355X HIP backend.")
    p.add_argument("functional", type=str, choices=["LDA", "GGA", "B3LYP"], help="Functional type")
    p.add_argument("xyzfile", type=str, help="Molecule name (e.g., H2O)")
    p.add_argument("--basis", default="sto-3g")            # grid.py:45 hard-codes sto-3g
    p.add_argument("--basis-file", default=None, help="NWChem / Gaussian94 basis file (Basis Set Exchange export) to register "
                                                     "under the --basis name: tables not shipped here (def2-SVP P, S; def2-TZVP N, O ...)")
    p.add_argument("--grid-level", type=int, default=3)   # grid.py:59
    p.add_argument("--quirks", type=int, default=1, help="1: reference formulas as shipped; 0: corrected VWN5/PBE-c derivatives")
    p.add_argument("--lib", default=None, help="path of libdft.so")
    p.add_argument("--eri", default="auto", choices=["auto", "dense", "cholesky"],
                   help="dense: the nao^4 tensor of grid.py:65; cholesky: factorised J/K; auto: dense while it stays below 8 GB (nao <= 178)")
    p.add_argument("--chol-tol", type=float, default=1e-9)
    p.add_argument("--device-resident", type=int, default=-1,
                   help="1: Fock build, DIIS, eigh and the density stay in HBM (only scalars cross PCIe per cycle); "
                        "0: host LAPACK for the eigenproblem; -1 (default): device from 400 basis functions")
    p.add_argument("--dist-backend", default="nccl", help="torch.distributed backend when launched with WORLD_SIZE > 1 (nccl = RCCL)")
    args = p.parse_args(argv)

    # one process per GPU: `python -m torch.distributed.run --nproc-per-node N -m quantum_compute_dft_amd.dft ...`
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = "cuda"
    if world > 1:
        import torch
        import torch.distributed as dist
        device = f"cuda:{int(os.environ.get('LOCAL_RANK', rank)) % max(1, torch.cuda.device_count())}"
        torch.cuda.set_device(device)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    if rank:  # rank 0 speaks
        sys.stdout = open(os.devnull, "w")

    atom_file = args.xyzfile if args.xyzfile.lower().endswith(".xyz") else args.xyzfile + ".xyz"
    atom_path = atom_file if os.path.exists(atom_file) else os.path.join(inputs.DATA_DIR, atom_file)
    if not os.path.exists(atom_path):
        print(f"Error: {atom_path} not found.")
        sys.exit(1)
    from .hostinfo import blas_threads
    _pool_pin = blas_threads()   # host BLAS/OpenMP pools on the CPU share from the first numpy call on (hostinfo.py)
    _pool_pin.__enter__()
    if args.basis_file:
        from . import basis as _b
        _b.load_basis_file(args.basis_file, args.basis)
        for _sym, _sh in _b._BASIS_SETS[args.basis.lower().replace("_", "-")].items():
            for _msg in _b.check_table(_sym, _sh):
                print("basis file check:", _msg)
    print(f"=== DFT Solver: {args.functional} | Molecule: {atom_file} ===")
    print("Building CPU data...")
    if args.eri == "auto":
        from . import basis as _basis
        _nao = _basis.build_shells(*_basis.parse_xyz(atom_path), args.basis).nao
        args.eri = "dense" if 8.0 * _nao ** 4 <= 8.0e9 else "cholesky"
    inp = inputs.build(atom_path, args.basis, args.grid_level, device=device, eri_mode=args.eri, chol_tol=args.chol_tol)
    print(f"System Info: NAO={inp.shells.nao}, Grid={inp.grids.size}, Occupied={inp.nocc}")
    print(f"Calculating AO Gradients ({args.functional} mode)..." if args.functional != "LDA" else "Skipping AO Gradients (LDA mode).")
    print("Moving data to GPU...")
    try:
        backend = scf.HipBackend(inp, args.functional, args.lib, quirks=bool(args.quirks), rank=rank, world=world, device=device,
                                 device_resident=None if args.device_resident < 0 else bool(args.device_resident))
    except Exception as e:  # dft.py:149-153
        print(e)
        sys.exit(1)
    print(f"GPU Init Time: {backend.init_time:.4f}s" + (f"  ({world} ranks: grid block + Cholesky-vector slice per GPU)" if world > 1 else ""))
    res = scf.run_scf(inp, backend, args.functional)
    if res["converged"]:
        print("-" * 80); print("Converged!")
        print(f"Total Energy: {res['E_tot']:.8f} Ha"); print(f"E_one       : {res['E_one']:.8f} Ha")
        print(f"E_coul      : {res['E_coul']:.8f} Ha"); print(f"E_nuc       : {inp.E_nuc:.8f} Ha")
        print(f"E_xc_dft    : {res['E_xc']:.8f} Ha")
        if args.functional == "B3LYP":
            print(f"E_ex_hf     : {res['E_ex_hf']:.8f} Ha")
        print(f"Total Time  : {res['total_time']:.4f} s"); print("-" * 80)
        print("Kernel Statistics (Avg per iter):"); print(f"XC(Exc+Vxc) Time: {res['xc_ms_avg']:.4f} ms")
        print(f"Median per cycle after the first: XC {res['xc_ms']:.4f} ms, J/K {res['jk_ms']:.4f} ms ({args.eri} ERI), "
              f"whole SCF iteration {res['iter_ms']:.4f} ms ({res['cycles']} cycles)")
        print("Host part of the cycle: " + ("device-resident (Fock build, DIIS, hipSOLVER eigh in HBM)" if backend.device_resident
                                            else "host LAPACK eigh; [dm|cocc] up and [J|K|Vxc] down in one pinned transfer each"))
        print("-" * 80)
    else:
        print("SCF Unconverged.")

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if importlib.util.find_spec("pyscf") is None or rank:   # dft.py:272-297 needs PySCF
        print("\nPySCF not importable here: reference cross-check skipped.")
        return res
    print("\nRunning PySCF reference calculation...")
    from pyscf import dft as pdft, gto
    mol = gto.Mole(); mol.atom = "".join(open(atom_path).readlines()[2:]); mol.basis = args.basis; mol.verbose = 0; mol.build()
    mf = pdft.RKS(mol); mf.grids.level = args.grid_level
    mf.xc = {"LDA": "slater,vwn5", "GGA": "PBE,PBE", "B3LYP": "b3lyp"}[args.functional]
    t0 = time.time(); mf.kernel(); el = time.time() - t0
    print(f"PySCF ({mf.xc}) Energy : {mf.e_tot:.8f} Hartree")
    print(f"Difference             : {abs(mf.e_tot - res['E_tot']):.2e} Hartree")
    print(f"PySCF Time             : {el:.4f} s")
    return res


if __name__ == "__main__":
    main()
