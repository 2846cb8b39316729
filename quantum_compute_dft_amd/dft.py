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
    p.add_argument("--eigensolver", default="auto", choices=["auto", "rotate", "exact"],
                   help="exact: eigh(F, S) every cycle as dft.py:227; rotate: occupied-subspace rotation from the previous cycle's "
                        "orbitals, full solver as first cycle and fallback; auto (default): rotate from 80 basis functions, exact below")
    p.add_argument("--device-resident", type=int, default=-1,
                   help="1: Fock build, DIIS, eigh and the density stay in HBM (only scalars cross PCIe per cycle); "
                        "0: host loop (one pinned transfer each way per cycle); -1 (default): device from 200 basis functions")
    p.add_argument("--fused-tail", type=int, default=-1,
                   help="1: the host part of the cycle as kernels of libdft.so (DFT_ScfTailStep; one rank, nao <= 512, nocc <= 64), "
                        "0: the host / torch loops, -1 (default): fused where it applies")
    p.add_argument("--ao", default="resident", choices=["resident", "direct"],
                   help="resident: AO values and gradients of the whole grid stay in HBM (the reference's layout, dft.py:155,172); "
                        "direct: they are re-evaluated chunk by chunk inside every XC call (DFT_ComputeXCDirect), memory ~100 MB")
    p.add_argument("--xc-occ", type=int, default=1, choices=[0, 1],
                   help="1 (default): the XC sweep's density step through the occupied orbitals (DFT_ComputeXCOcc, 4 nao nocc flops "
                        "per grid point); 0: the reference's call with the full density matrix (DFT_ComputeXC, dft.py:206)")
    p.add_argument("--both-quirks", action="store_true",
                   help="LDA/GGA: run the SCF twice, with the reference's formulas as shipped (its CUDA path) and with the "
                        "corrected VWN5 / PBE-c derivatives (what PySCF's slater,vwn5 / PBE,PBE compute), and report both energies")
    p.add_argument("--json", default=None, help="also append the run's one-line JSON record to this file")
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
    inp = inputs.build(atom_path, args.basis, args.grid_level, device=device, eri_mode=args.eri, chol_tol=args.chol_tol, rank=rank, world=world)
    print(f"System Info: NAO={inp.shells.nao}, Grid={inp.grids.size}, Occupied={inp.nocc}")
    print(f"Calculating AO Gradients ({args.functional} mode)..." if args.functional != "LDA" else "Skipping AO Gradients (LDA mode).")
    print("Moving data to GPU...")
    try:
        backend = scf.HipBackend(inp, args.functional, args.lib, quirks=bool(args.quirks), rank=rank, world=world, device=device,
                                 device_resident=None if args.device_resident < 0 else bool(args.device_resident),
                                 fused_tail=None if args.fused_tail < 0 else bool(args.fused_tail),
                                 eigensolver=args.eigensolver, ao_mode=args.ao, xc_occ=bool(args.xc_occ))
    except Exception as e:  # dft.py:149-153
        print(e)
        sys.exit(1)
    print(f"GPU Init Time: {backend.init_time:.4f}s" + (f"  ({world} ranks: grid block + Cholesky-vector slice per GPU)" if world > 1 else ""))
    res = scf.run_scf(inp, backend, args.functional)
    eig_stats = dict(backend.occ_solver.stats) if backend.occ_solver is not None else None   # of THIS run (the second one below adds to the counters)
    other = None
    if args.both_quirks and args.functional != "B3LYP":   # B3LYP's four components are derivative-correct: one answer
        backend.solver.set_option("quirks", 0 if args.quirks else 1)
        if backend.occ_solver is not None:
            backend.occ_solver.reset()            # fresh full solve: the second SCF does not start from the first one's rotation
        other = scf.run_scf(inp, backend, args.functional, log=None)
        backend.solver.set_option("quirks", 1 if args.quirks else 0)
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
        if backend.occ_solver is not None:
            st = eig_stats
            print(f"Eigensolver: {st['rotated']} cycles by occupied-subspace rotation ({st['inner_steps']} fixed-point steps), {st['exact']} by full diagonalisation")
        eig_dev = "occupied-subspace rotation, hipSOLVER eigh as fallback," if backend.occ_solver is not None else "hipSOLVER eigh"
        print("Host part of the cycle: " + ("device-resident: Fock build, DIIS, occupied-subspace rotation, density and energy traces as kernels of "
                                            "libdft.so (DFT_ScfTailStep); full diagonalisations by " + ("hipSOLVER" if inp.shells.nao >= 400 else "host LAPACK") if getattr(backend, "tail", None) is not None else
                                            f"device-resident (Fock build, DIIS, {eig_dev} in HBM)" if backend.device_resident
                                            else "host LAPACK eigh; [dm|cocc] up and [J|K|Vxc] down in one pinned transfer each"))
        print("-" * 80)
    else:
        print("SCF Unconverged.")
    if other is not None:
        a, b = ("reference formulas as shipped", "corrected derivatives") if args.quirks else ("corrected derivatives", "reference formulas as shipped")
        print(f"Total Energy, {a:32s}: {res['E_tot']:.8f} Ha   (this run; --quirks {args.quirks})")
        print(f"Total Energy, {b:32s}: {other['E_tot']:.8f} Ha   (difference {res['E_tot'] - other['E_tot']:+.2e} Ha)")
    import json
    record = {"functional": args.functional, "molecule": os.path.splitext(atom_file)[0], "basis": args.basis, "grid_level": args.grid_level,
              "nao": int(inp.shells.nao), "ngrid": int(inp.grids.size), "nocc": int(inp.nocc), "n_gpus": world, "eri": args.eri,
              "quirks": int(args.quirks), "converged": bool(res["converged"]), "cycles": int(res.get("cycles", 0)),
              "E_tot": res.get("E_tot"), "E_one": res.get("E_one"), "E_coul": res.get("E_coul"), "E_xc": res.get("E_xc"),
              "E_ex_hf": res.get("E_ex_hf"), "E_nuc": float(inp.E_nuc), "total_time_s": res.get("total_time"),
              "xc_ms_avg": res.get("xc_ms_avg"), "xc_ms": res.get("xc_ms"), "jk_ms": res.get("jk_ms"), "iter_ms": res.get("iter_ms"),
              "cycle_ms": res.get("cycle_ms"), "gpu_init_s": backend.init_time, "device_resident": bool(backend.device_resident), "loop": res.get("loop", "device" if backend.device_resident else "host"), "ao": args.ao, "eigensolver": args.eigensolver,
              "eigensolver_stats": eig_stats, "xc_occ": int(backend.xc_occ)}
    if other is not None:
        record["E_tot_other_quirks"] = other.get("E_tot"); record["other_quirks"] = 0 if args.quirks else 1
    line = json.dumps(record)
    print(line)                                  # one JSON line per run for a harness
    if args.json and not rank:
        with open(args.json, "a") as fh:
            fh.write(line + "\n")

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
