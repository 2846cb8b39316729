#!/usr/bin/env python3
"""Benchmark of the XC sweep (DFT_ComputeXC) on MI355X.

python bench.py --gpus N --steps K --warmup W

N > 1 started as a plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) launches its own
N rank processes -- a `python -m torch.distributed.run` child on 127.0.0.1, started BEFORE anything in
this process touches the GPU -- and exits with the child's code; started by torch.distributed.run itself
it is one of the ranks.  One process per GPU, RCCL ("nccl") over xGMI.

Step   = one DFT_ComputeXC call on one batch of synthetic AO/grid data resident in HBM
         (the bracket the reference times at dft.py:205-208: call + device sync), followed for
         N>1 by the RCCL all-reduce of [Vxc | Exc] over the grid shards.
Work   = BASELINE.json's metric config: Benzene GGA(PBE) def2-SVP shape (nao 114, ngrid 143 556)
         per GPU (weak scaling: every rank owns a full-size grid shard).  `--scaling strong` shards ONE
         molecule's grid over the ranks instead (default there: BASELINE config 5, C33H56N7O17P3S
         B3LYP/def2-SVP, nao 1150, 1 436 406 points / N); the default run carries the same measurement
         as the `strong_config5` object so one driver invocation per N yields both curves.
Output = ONE JSON line on rank 0 (metric grid-points/s).  `roofline` is the WHOLE call against SURVEY
         8(d)'s algorithmic bytes/flops (its `dominant_kernel` member is the single-kernel figure, HIP
         events on the solver's stream); `cpu_baseline` is the prebuilt OpenMP CPU oracle timed on a
         bounded slice of the same inputs (N=1, rank 0 only); `ao_sweep` the AO-on-grid kernel on
         Benzene/def2-SVP's real shells and level-3 grid; `scf_iteration`, `scf_iteration_factorised_j`,
         `scf_iteration_anthracene` = ms per SCF cycle of the driver's own loop on the real Benzene PBE/def2-SVP and
         Anthracene B3LYP/def2-TZVP-shaped basis (converged energies as checksums); `scf_iteration*_synthetic` the same
         loop body on synthetic operands with eigh(F, S) every cycle (the reference's loop; round 1's legs);
         `scf_benzene_real` / `scf_anthracene_def2svp_real` the driver's whole SCF on the real molecules (energy as checksum; the
         second one sharded over the N ranks); `k_build` the factorised exact exchange on the fp64 matrix cores; `small_basis` us per
         synchronous call at bases of at most 32 functions (H2O/def2-SVP ... shapes): default options against the one-pass kernel
         forced on / off (option `tiny`).
The headline is measured first and its line is complete before any extra leg starts; the legs run under a watchdog
         (`--legs-seconds`) and a leg that fails or hangs only costs its own member (`legs_note` says so).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import quantum_compute_dft_amd as q  # noqa: E402

WORKLOADS = {
    # name: (functional, nao, ngrid)   sizes from SURVEY.md section 8
    "benzene_gga_def2svp": ("GGA", 114, 143556),
    "h2o_lda_def2svp": ("LDA", 24, 34310),
    "anthracene_b3lyp_def2tzvp": ("B3LYP", 494, 294868),
    "anthracene_b3lyp_sto3g": ("B3LYP", 80, 294868),
    "anthracene_b3lyp_def2svp": ("B3LYP", 246, 294868),
    "c33_b3lyp_def2svp": ("B3LYP", 1150, 1436406),      # BASELINE config 5 on ONE GPU: 52.9 GB of planes + 73 GB of Cholesky vectors
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # AMD datasheet fp64 matrix peak (the local guide lists no fp64 figure); measured here:
                             # 77.9 TF on constant operands at 2.39 GHz, ~59 TF on real data (clock drops to ~1.8 GHz),
                             # profiles/r01_mfma_f64_probe2.txt and DESIGN.md
SEED = 20260128
NOCC = {114: 21, 24: 5, 494: 47, 80: 47, 246: 47, 1150: 250}   # closed-shell occupied orbitals: Benzene 21, H2O 5, Anthracene 47, C33H56N7O17P3S 250


def synth(ngrid, nao, need_grad, dev, seed):
    """SURVEY 8(d) recipe, generated on the device: ao=0.4 N, grad=0.3 N, w=0.05 U, dm=2CC^T."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g) if need_grad else None
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    nocc = NOCC.get(nao, -(-nao // 5))     # occupied orbitals of the named molecules; SURVEY's ceil(nao/5) elsewhere
    C = 0.7 * torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = 2.0 * C @ C.T
    return dm.contiguous(), ao, gr, w, (float(np.sqrt(2.0)) * C).contiguous()      # dm = cocc cocc^T


def kernel_model(name, xc, ngrid, nao):
    """Algorithmic bytes and flops of one launch (B = ngrid*(8*nao*c+8)+16*nao^2,
    F = 4*ngrid*nao^2 + (2c+8[c=4])*ngrid*nao for the whole sweep; each of the two contraction
    kernels streams the c AO planes once and does half of the GEMM flops)."""
    c = 1 if xc == "LDA" else 4
    planes = 8.0 * ngrid * nao * c
    if name == "rho":
        return planes + 8.0 * nao * nao + 8.0 * ngrid * c, 2.0 * ngrid * nao * nao + 2.0 * c * ngrid * nao
    if name == "vxc":
        return planes + 8.0 * ngrid * c + 8.0 * nao * nao, 2.0 * ngrid * nao * nao + 2.0 * c * ngrid * nao
    if name == "xc_sweep":  # fused single kernel
        return planes + 8.0 * ngrid + 16.0 * nao * nao, 4.0 * ngrid * nao * nao + (2 * c + (8 if c == 4 else 0)) * ngrid * nao
    if name == "xc_points":
        return 8.0 * ngrid * (2 + 2 * c), 0.0
    return 0.0, 0.0


from quantum_compute_dft_amd.hostinfo import blas_threads, host_cpu_share  # noqa: E402


def cpu_baseline(xc, dm, ao, gr, w, target_seconds):
    """The OpenMP build of the CPU oracle (a port of the reference's loop structure,
    dft_solver.cu:294-432 + B^T.AO) timed on a bounded slice of the same inputs."""
    import oracle                                   # test infrastructure: the checker, timed as the baseline
    threads = host_cpu_share()
    os.environ["OMP_NUM_THREADS"] = str(threads)    # the prebuilt OpenMP checker is loaded, never compiled here
    t = {"LDA": 0, "GGA": 1, "B3LYP": 2}[xc]
    ngrid = ao.shape[0]

    def run(n):
        h = lambda a: None if a is None else np.ascontiguousarray(a.cpu().numpy())
        dm_h, ao_h, w_h = h(dm), h(ao[:n]), h(w[:n])
        gr_h = None if gr is None else h(gr[:, :n])
        t0 = time.perf_counter()
        oracle.compute_xc(t, dm_h, ao_h, w_h, gr_h, omp=True)
        return time.perf_counter() - t0

    run(256)                                        # warm (library load, thread pool)
    probe = min(4096, ngrid)
    rate = probe / run(probe)
    sample = int(min(ngrid, max(probe, rate * target_seconds)))
    dt = run(sample)
    return {"value": sample / dt, "unit": "grid-points/s", "cores": threads, "kind": "port",
            "seconds": dt,
            "sample": f"first {sample} of {ngrid} grid points of the same inputs, OpenMP build of "
                      f"oracle/xc_oracle.c (reference loop structure, dft_solver.cu:294-432), "
                      f"{threads} threads"}


def scf_iteration_ms(solver, xc, nao, ngrid, dm, ao, gr, w, dev, iters=9, eri="auto"):
    """One SCF iteration as dft.py:199-236 does it -- eigh(F, S) EVERY cycle, the reference's loop -- on the same
    synthetic shapes (a synthetic Fock sequence says nothing about scf.OccupiedRotation, which the driver uses from 80
    functions: see scf_real_leg for the real molecule).  Below 400 functions (host LAPACK eigh): ONE pinned upload
    [dm | cocc], J (+K for B3LYP), XC sweep, ONE pinned download [J | K | Vxc], Fock build + eigh on the host.  From
    400 functions: device-resident -- Fock build, hipSOLVER eigh and dm = cocc cocc^T in HBM, one 4-double download.
    `eri`: "dense" = one pass over a synthetic dense ERI (the reference's formulation, nao <= 200), "cholesky" =
    6 nao synthetic Cholesky vectors (DFT_ComputeJKFactorized), "auto" = dense up to 200 functions."""
    from quantum_compute_dft_amd.scf import FockDiagonaliser
    f64 = torch.float64
    n2 = nao * nao
    nocc = {114: 21, 24: 5, 494: 47, 80: 47, 246: 47, 1150: 250}.get(nao, max(1, nao // 5))  # occupied orbitals of the named molecules
    dense = (nao <= 200) if eri == "auto" else eri == "dense"
    if dense:
        d_eri = torch.randn((n2, n2), dtype=f64, device=dev) * 1e-3
        store = 8.0 * n2 * n2
    else:
        naux = int(min(6 * nao, 80e9 / (8.0 * n2)))
        chol = torch.randn((naux, nao, nao), dtype=f64, device=dev) * 1e-2
        store = 8.0 * naux * n2
    resident = nao >= 400
    S = np.eye(nao); H = np.diag(np.linspace(-1.0, 1.0, nao))
    solve = FockDiagonaliser(S, dev, device_from=0 if resident else 400)
    up = torch.zeros(n2 + nao * nocc, dtype=f64, device=dev)                       # [dm | cocc]
    d_dm, d_c = up[:n2].view(nao, nao), up[n2:].view(nao, nocc)
    down = torch.zeros(3 * n2, dtype=f64, device=dev)                               # [J | K | Vxc]
    d_J, d_K, d_v = (down[k * n2:(k + 1) * n2].view(nao, nao) for k in range(3))
    pin_up, pin_down = torch.empty(up.shape, dtype=f64).pin_memory(), torch.empty(down.shape, dtype=f64).pin_memory()
    C = np.linalg.qr(np.random.default_rng(SEED).normal(size=(nao, nocc)))[0]
    cocc_h = np.ascontiguousarray(np.sqrt(2.0) * C); dm_h = cocc_h @ cocc_h.T
    want_k = xc == "B3LYP"
    if resident:
        Hd = torch.as_tensor(H, dtype=f64, device=dev); X = solve.X
        cocc = torch.as_tensor(cocc_h, dtype=f64, device=dev); dmd = cocc @ cocc.T
    rows = []
    pin = blas_threads(1 if nao < 400 else None); pin.__enter__()   # as scf.run_scf pins the host pools

    def device_parts():
        if not dense:
            solver.compute_jk_factorized(nao, naux, nocc, chol, d_dm, d_c if want_k else None, d_J, d_K if want_k else None)
        elif want_k:
            solver.compute_jk(nao, d_eri, d_dm, d_J, d_K)
        else:
            solver.compute_coulomb(nao, d_eri, d_dm, d_J)
        torch.cuda.synchronize(); t_jk = time.perf_counter()
        solver.compute_xc(ngrid, nao, d_dm, ao, w, d_v, gr)
        return t_jk, time.perf_counter()

    for it in range(iters + 1):
        t0 = time.perf_counter()
        if resident:
            d_dm.copy_(dmd); d_c.copy_(cocc)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            t2, t3 = device_parts()
            t4 = t3
            F = Hd + 1e-3 * (d_J + 0.5 * (d_v + d_v.T) - (0.1 * d_K if want_k else 0.0))
            e, Cp = torch.linalg.eigh(X.T @ F @ X)
            cocc = (X @ Cp[:, :nocc]) * float(np.sqrt(2.0)); dm_new = cocc @ cocc.T
            scal = torch.stack([(dm_new * Hd).sum(), (dm_new * d_J).sum(), (dm_new * d_K).sum(), torch.linalg.norm(dm_new - dmd)]).tolist()
            dmd = dm_new
        else:
            pin_up[:n2].copy_(torch.as_tensor(dm_h).reshape(-1)); pin_up[n2:].copy_(torch.as_tensor(cocc_h).reshape(-1))
            up.copy_(pin_up, non_blocking=True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            t2, t3 = device_parts()
            pin_down.copy_(down, non_blocking=True); torch.cuda.synchronize()
            hd = pin_down.numpy()
            J, K, V = (hd[k * n2:(k + 1) * n2].reshape(nao, nao) for k in range(3))
            t4 = time.perf_counter()
            F = H + 1e-3 * (J + 0.5 * (V + V.T) - (0.1 * K if want_k else 0.0))
            e, Cf = solve(F)
            cocc_h = np.ascontiguousarray(np.sqrt(2.0) * Cf[:, :nocc]); dm_h = cocc_h @ cocc_h.T
        t5 = time.perf_counter()
        if it:  # first iteration warms allocations
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0))
    # medians: a 16-core share of a 256-core host stalls a cycle for tens of ms now and then (seen: one
    # 24 ms host part among 15 cycles of 0.7 ms); the worst cycle is reported next to them
    med = 1e3 * np.median(np.array(rows), axis=0)
    parts = dict(zip(("upload", "jk", "xc", "download", "fock_eigh_density"), (float(v) for v in med[:5])))
    t_all, t_worst = float(med[5]), 1e3 * max(r[5] for r in rows)
    pin.__exit__(None, None, None)
    if dense:
        del d_eri
    else:
        del chol
    torch.cuda.empty_cache()
    how = ("one pass over a synthetic dense ERI" if dense else f"{naux} synthetic Cholesky vectors (factorised)")
    return {"ms": t_all, "parts_ms": parts, "statistic": f"median of {iters} cycles", "worst_cycle_ms": t_worst, "eri_bytes": store,
            "form": "device-resident: Fock build, eigh (hipSOLVER syevd EVERY cycle: the synthetic Fock sequence says nothing about an "
                    "SCF trajectory; the driver's occupied-subspace rotation takes the real Anthracene/def2-TZVP cycle from 23.5 to "
                    "15.7 ms, profiles/r02_eigensolver.txt), density in HBM; scalars only cross PCIe" if resident
                    else "host LAPACK eigh; [dm|cocc] up and [J|K|Vxc] down in one pinned transfer each",
            "note": "synthetic dm; J" + ("+K" if want_k else "") + f" from {how}, XC, Fock build + eigh as in dft.py:199-236"}


def k_build_mfma(lib_path, dev, nao=494, nocc=47, naux=3000, reps=5):
    """Exact-exchange build on the fp64 matrix cores (north_star (d)): DFT_ComputeJKFactorized on synthetic
    Cholesky vectors of the Anthracene/def2-TZVP shape (dense ERI there: 476 GB).  Algorithmic flops:
    2 naux nao^2 nocc for the half transform + naux nocc nao (nao+1) for K = Yt^T Yt as a symmetric
    rank-k update (the kernel mirrors the lower tiles); kernel times from HIP events inside the library."""
    g = torch.Generator(device=dev); g.manual_seed(SEED)
    L = torch.randn((naux, nao, nao), dtype=torch.float64, device=dev, generator=g) * 0.1
    c = torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = c @ c.T
    J = torch.zeros((nao, nao), dtype=torch.float64, device=dev); K = torch.zeros_like(J)
    s = q.DFTSolverWrapper(lib_path, "B3LYP")
    s.set_option("profile", 1)
    acc = {}
    for it in range(reps + 2):
        s.compute_jk_factorized(nao, naux, nocc, L, dm, c, J, K)
        torch.cuda.synchronize()
        if it >= 2:
            for k, v in s.timings():
                acc[k] = acc.get(k, 0.0) + v / reps
    fl_half = 2.0 * naux * nao * nao * nocc                 # Yt_P = Cocc^T L_P
    fl_syrk = 1.0 * naux * nocc * nao * (nao + 1)           # K = Yt^T Yt counted as a symmetric rank-k update
    t_k = acc["cd_half"] + acc["cd_k"]
    del L, s
    torch.cuda.empty_cache()
    return {"workload": f"anthracene_b3lyp_def2tzvp shape: nao={nao} nocc={nocc} naux={naux} synthetic Cholesky vectors "
                        f"({8.0 * naux * nao * nao / 1e9:.2f} GB resident)",
            "kernels_ms": acc, "k_ms": t_k, "flops": fl_half + fl_syrk,
            "achieved": (fl_half + fl_syrk) / t_k / 1e9, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": (fl_half + fl_syrk) / t_k / 1e9 / F64_MFMA_PEAK_TF,
            "full_square_equivalent_tflops": 2 * fl_half / t_k / 1e9,
            # the J pass (k_cd_axpy) reads the upper triangle of every symmetric vector only: 4 naux nao (nao + 1) bytes
            "j_pass_gbs": 4.0 * naux * nao * (nao + 1) / acc["cd_j"] / 1e6,
            "j_pass_note": "bytes the kernel reads (upper triangles), not the full-square 8 naux nao^2"}


def pmc_traffic(workload, kernels):
    """HBM bytes per launch of the kernels of one call from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE): the contraction kernels from the newest profiles/r*_pmc_sweep.json
    (by role: rho, vxc, rho_occ; Benzene and Anthracene/def2-TZVP shapes), the small kernels (xc_points, reduce_vxc) from the
    newest r*_pmc_traffic.json of the same workload.  Also the MFMA-pipe utilisation the sweep file holds.  None when nothing
    matches; `covered` lists the kernels the sum includes."""
    per, mfma, src = {}, {}, []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload:
            got = {k: v["hbm_bytes"] for k, v in d.get("kernels", {}).items() if k in kernels}
            if got:
                per.update(got); src.append(os.path.basename(f))
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sweep.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for e in d.get("workloads", {}).get(workload, {}).values():
            if e.get("role") in kernels:
                if "hbm_bytes" in e:
                    per[e["role"]] = e["hbm_bytes"]
                if "mfma_utilisation" in e:
                    mfma[e["role"]] = e["mfma_utilisation"]
                src.append(os.path.basename(f))
    if not per:
        return None
    return {"per_kernel": per, "mfma_utilisation": mfma, "covered": sorted(per), "source": ", ".join(sorted(set(src)))}


def small_basis_leg(lib_path, dev, reps=200, rounds=5):
    """Synchronous DFT_ComputeXC at small bases (BASELINE configs[0]'s class of molecule: H2O/def2-SVP has 24 functions and
    34 310 grid points): us per call of the default options against the one-pass kernel forced on / off (option `tiny`,
    csrc/xc_tiny_kernels.hpp) on synthetic planes of the named shapes.  Median of `rounds` timed bursts of `reps` calls."""
    out = {"unit": "us per synchronous DFT_ComputeXC call", "statistic": f"median of {rounds} bursts of {reps} calls"}
    for name, xc, ngrid, nao in (("h2o_def2svp_lda", "LDA", 34310, 24), ("h2o_def2svp_gga", "GGA", 34310, 24),
                                 ("h2o_sto3g_gga", "GGA", 34310, 7), ("ch4_sto3g_gga", "GGA", 56000, 9),
                                 ("nh3_def2svp_b3lyp", "B3LYP", 45000, 29), ("h2o2_sto3g_b3lyp", "B3LYP", 46000, 12),
                                 ("small_basis_large_grid_gga", "GGA", 300000, 32)):
        dm, ao, gr, w, _ = synth(ngrid, nao, xc != "LDA", dev, SEED)
        row = {"ngrid": ngrid, "nao": nao, "functional": xc}
        for label, tiny in (("default", None), ("four_launches", 0), ("one_pass", 1)):
            s = q.DFTSolverWrapper(lib_path, xc)
            if tiny is not None:
                s.set_option("tiny", tiny)
            v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            for _ in range(30):
                e = s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
            ts = []
            for _ in range(rounds):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    e = s.compute_xc(ngrid, nao, dm, ao, w, v, gr)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / reps)
            row[label] = 1e6 * float(np.median(ts))
            row["exc_" + label] = e
        out[name] = row
        del ao, gr
    torch.cuda.empty_cache()
    return out


def ao_sweep_leg(lib_path, dev, reps=6, burst=10):
    """North-star kernel (a): DFT_EvalAO (values + gradients, grid.py:30-31,38) on Benzene/def2-SVP's real
    shell table and its real level-3 grid.  Algorithmic bytes ngrid*(8*nao*4 + 32): the
    kernel is HBM-write bound.  HIP events recorded by the library around the launch; also the chained
    AO -> rho -> Vxc rate (DFT_EvalAO + DFT_ComputeXC per step, wall clock)."""
    from quantum_compute_dft_amd import basis, grid_gen, inputs
    syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, "Benzene.xyz"))
    sh = basis.build_shells(syms, xyz, "def2-svp")
    grids = grid_gen.Grids(syms, xyz, level=3, device=dev)
    ngrid, nao = grids.size, sh.nao
    d_c = torch.as_tensor(grids.coords, device=dev)
    d_w = torch.as_tensor(grids.weights, device=dev)
    d_ao = torch.empty((ngrid, nao), dtype=torch.float64, device=dev)
    d_gr = torch.empty((3, ngrid, nao), dtype=torch.float64, device=dev)
    s = q.DFTSolverWrapper(lib_path, "GGA")
    s.set_option("profile", 1)
    ms = []
    for r in range(reps + 1):
        for _ in range(burst):
            s.eval_ao(sh, d_c, ngrid, d_ao, d_gr)
        t = dict(s.timings())["eval_ao"]                 # the burst's last launch
        if r:
            ms.append(t)
    t_ao = float(np.mean(ms))
    s.set_option("profile", 0)
    g = torch.Generator(device=dev); g.manual_seed(SEED)
    C = 0.3 * torch.randn((nao, 21), dtype=torch.float64, device=dev, generator=g)
    dm = (2.0 * C @ C.T).contiguous()
    d_v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
    for _ in range(5):
        s.eval_ao(sh, d_c, ngrid, d_ao, d_gr); s.compute_xc(ngrid, nao, dm, d_ao, d_w, d_v, d_gr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        s.eval_ao(sh, d_c, ngrid, d_ao, d_gr)
        exc = s.compute_xc(ngrid, nao, dm, d_ao, d_w, d_v, d_gr)
    torch.cuda.synchronize(); t_chain = (time.perf_counter() - t0) / n
    b_alg = ngrid * (8.0 * nao * 4 + 32)
    nelec = float((d_w * ((d_ao @ dm) * d_ao).sum(1)).sum())
    # the same step without resident planes: DFT_ComputeXCDirect (chunks of the grid through a workspace)
    d_e = torch.zeros(1, dtype=torch.float64, device=dev)
    direct = {}
    for label, chunk in (("auto", 0), ("one_chunk", ngrid)):
        for _ in range(5):
            s.compute_xc_direct(sh, ngrid, d_c, d_w, dm, d_v, d_e, chunk)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            s.compute_xc_direct(sh, ngrid, d_c, d_w, dm, d_v, d_e, chunk)
            exc_d = float(d_e.item())
        direct[label] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / n, "exc": exc_d}
    direct["note"] = ("DFT_ComputeXCDirect: AO values and gradients re-evaluated per call into a chunk workspace (auto: ~96 MB of planes), "
                      "never resident for the whole grid; memory 8*chunk*nao*4 B instead of 8*ngrid*nao*4 B")
    return {"workload": f"DFT_EvalAO deriv 1, Benzene/def2-SVP real shells ({len(sh.l)} shells, nao {nao}) on its level-3 grid ({ngrid} points)",
            "kernel_ms": t_ao, "alg_bytes": b_alg, "bound": "hbm", "achieved": b_alg / t_ao / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b_alg / t_ao / 1e6 / HBM_PEAK_GBS,
            "chained_ao_rho_vxc": {"ms_per_step": 1e3 * t_chain, "grid_points_per_sec": ngrid / t_chain,
                                   "exc": exc, "integral_rho": nelec,
                                   "note": "DFT_EvalAO + DFT_ComputeXC(GGA) per step on the real AO values, synthetic PSD density matrix"},
            "direct_ao_rho_vxc": direct}


def scf_real_leg(lib_path, dev, molecule="Benzene", functional="GGA", basis_name="def2-svp"):
    """The BASELINE metric "ms/SCF-iter (Benzene GGA)" on the REAL molecule: the driver's own loop (scf.run_scf, the
    reference's thresholds |dE| < 1e-8, |d dm| < 1e-6, dft.py:243) on Benzene PBE/def2-SVP -- real shells, integrals,
    level-3 grid -- once with the dense ERI (the reference's formulation, dft.py:166,203) and once with the
    factorised J (Cholesky, 1e-8); eigensolver "auto" (occupied-subspace rotation at this size) and eigh(F, S) every
    cycle (dft.py:227) side by side.  The converged energy is the checksum (rounds 1 and 2: -231.77070180 Ha)."""
    import dataclasses
    from quantum_compute_dft_amd import inputs, scf
    from quantum_compute_dft_amd.cholesky import cholesky_eri
    t0 = time.perf_counter()
    inp = inputs.build(molecule, basis_name, 3, device=dev, verbose=False, eri_mode="dense")
    t_build = time.perf_counter() - t0
    inp_cd = dataclasses.replace(inp, eri=None, chol=cholesky_eri(inp.shells, tol=1e-8, device=dev))
    out = {"workload": f"{molecule} {functional}/{basis_name}: nao {inp.shells.nao}, {inp.grids.size} grid points, {inp.nocc} occupied; real shells, "
                       f"integrals and level-3 grid (host build {t_build:.1f} s, not timed)",
           "statistic": "median per cycle after the first; thresholds of dft.py:243"}
    for name, ii in (("dense_eri", inp), ("factorised_j", inp_cd)):
        # auto / exact: the XC sweep through DFT_ComputeXCOcc (the loop holds cocc); abi_xc: eigensolver auto with the
        # reference ABI's DFT_ComputeXC (full density matrix) beside it
        # host_loop: rounds 2-3's loop (Fock assembly, DIIS, rotation in numpy on the host, one pinned transfer each way)
        for eig, occ in (("auto", True), ("exact", True), ("abi_xc", False), ("host_loop", True)):
            be = scf.HipBackend(ii, functional, lib_path, device=dev, eigensolver="exact" if eig == "exact" else "auto", xc_occ=occ,
                                device_resident=False if eig == "host_loop" else None)
            r = scf.run_scf(ii, be, functional, log=None)
            out[f"{name}_{eig}"] = {"ms_per_cycle": r["iter_ms"], "xc_ms": r["xc_ms"], "jk_ms": r["jk_ms"], "cycles": r["cycles"],
                                    "converged": bool(r["converged"]), "E_tot": r["E_tot"], "total_ms": 1e3 * r["total_time"],
                                    "xc_entry_point": "DFT_ComputeXCOcc" if occ else "DFT_ComputeXC",
                                    "loop": r.get("loop", "host"),
                                    "eigensolver": dict(be.occ_solver.stats) if be.occ_solver is not None else "eigh(F, S) every cycle"}
            del be
    del inp, inp_cd
    torch.cuda.empty_cache()
    return out


def scf_real_sharded_leg(lib_path, dev, world, rank, molecule="Anthracene", functional="B3LYP", basis_name="def2-svp", tol=1e-8):
    """The BASELINE metric "ms/SCF-iter (Anthracene B3LYP)" through the driver's own path on N GPUs: real shells,
    level-3 grid and Cholesky vectors of the ERI (every rank builds the inputs, keeps its grid block and its slice of
    the vectors: scf.HipBackend), the loop of scf.run_scf with rank 0 authoritative -- ONE all-reduce of
    [Vxc | J | K | Exc] and ONE broadcast of [dm | cocc | scalars] per cycle.  def2-SVP (nao 246): the def2-TZVP-shaped
    basis of config 3 needs 12 s of host integrals per rank, too long for a bench leg (its cycle: profiles/r02_scf_*).
    The converged energy is the checksum (-539.14207342 Ha on one GPU)."""
    from quantum_compute_dft_amd import inputs, scf
    t0 = time.perf_counter()
    inp = inputs.build(molecule, basis_name, 3, device=dev, verbose=False, eri_mode="cholesky", chol_tol=tol, rank=rank, world=world)
    t_build = time.perf_counter() - t0
    abi = torch_loop = None
    if world == 1:   # the same SCF with the reference ABI's DFT_ComputeXC (full density matrix) in the sweep, beside the default
        be = scf.HipBackend(inp, functional, lib_path, device=dev, xc_occ=False)
        ra = scf.run_scf(inp, be, functional, log=None)
        abi = {"xc_entry_point": "DFT_ComputeXC", "ms_per_cycle": ra["iter_ms"], "xc_ms": ra["xc_ms"], "jk_ms": ra["jk_ms"], "cycles": ra["cycles"], "E_tot": ra["E_tot"],
               "loop": ra.get("loop", "device")}
        del be
        # ... and with rounds 2-3's device-resident loop (Fock assembly, DIIS and the rotation as torch operations) instead of the tail kernels
        be = scf.HipBackend(inp, functional, lib_path, device=dev, fused_tail=False)
        rt = scf.run_scf(inp, be, functional, log=None)
        torch_loop = {"ms_per_cycle": rt["iter_ms"], "xc_ms": rt["xc_ms"], "jk_ms": rt["jk_ms"], "cycles": rt["cycles"], "E_tot": rt["E_tot"],
                      "eigensolver": dict(be.occ_solver.stats) if be.occ_solver is not None else None}
        del be
    be = scf.HipBackend(inp, functional, lib_path, rank=rank, world=world, device=dev)
    r = scf.run_scf(inp, be, functional, log=None)
    out = {"xc_entry_point": "DFT_ComputeXCOcc (occupied orbitals; scf.HipBackend default)", "abi_xc_entry": abi, "torch_loop": torch_loop,
           "loop": r.get("loop", "device-resident torch loop" if be.device_resident else "host"),
           "workload": f"{molecule} {functional}/{basis_name}: nao {inp.shells.nao}, {inp.grids.size} grid points, {(inp.chol_range[2] if inp.chol_range else inp.chol.shape[0])} Cholesky vectors "
                       f"({tol:g}), sharded over {world} GPU(s); inputs built in {t_build:.1f} s (not timed; the Cholesky factorisation on rank 0 alone, "
                       f"vector slices sent to the ranks)",
           "scaling": "strong", "ms_per_cycle": r["iter_ms"], "xc_ms": r["xc_ms"], "jk_ms": r["jk_ms"], "cycles": r["cycles"],
           "converged": bool(r["converged"]), "E_tot": r["E_tot"], "total_ms": 1e3 * r["total_time"],
           "device_resident": bool(be.device_resident),
           "eigensolver": dict(be.occ_solver.stats) if be.occ_solver is not None else "eigh(F, S) every cycle",
           "statistic": "rank 0's median per cycle after the first (every cycle ends in a collective, so all ranks keep its pace)"}
    del be, inp
    torch.cuda.empty_cache()
    return out


def strong_leg(lib_path, dev, dist, backend, world, rank, workload, steps=6, warmup=2):
    """ONE molecule's grid sharded over the ranks (BASELINE config 5): rank r keeps block
    grid_shard.shard_bounds(ngrid, world, r) resident, a step is the local sweep + ONE all-reduce of
    [Vxc | Exc] (nao^2+1 doubles).  Per step the sweep and the collective are bracketed separately
    (device sync between them) so the payload's time is reported next to its size."""
    from quantum_compute_dft_amd.grid_shard import shard_bounds
    xc, nao, ngrid = WORKLOADS[workload]
    lo, hi = shard_bounds(ngrid, world, rank)
    n_loc = hi - lo
    dm, ao, gr, w, cocc = synth(max(n_loc, 16), nao, xc != "LDA", dev, SEED + 7919 * (rank + 1))
    if world > 1:
        for t_ in (dm, cocc):
            if backend == "nccl":
                dist.broadcast(t_, 0)
            else:
                h = t_.cpu(); dist.broadcast(h, 0); t_.copy_(h)
    solver = q.DFTSolverWrapper(lib_path, xc)
    nocc = cocc.shape[1]
    out = torch.zeros(nao * nao + 1, dtype=torch.float64, device=dev)
    d_v, d_e = out[: nao * nao], out[nao * nao:]
    def measure(entry):
        rows = []
        for it in range(warmup + steps):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if not n_loc:
                out.zero_()
            elif entry == "occ":
                solver.compute_xc_occ_async(n_loc, nao, nocc, cocc, ao, w, d_v, d_e, gr, dm)
            else:
                solver.compute_xc_async(n_loc, nao, dm, ao, w, d_v, d_e, gr)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if world > 1:
                if backend == "nccl":
                    dist.all_reduce(out)
                else:
                    h = out.cpu(); dist.all_reduce(h); out.copy_(h)
            exc = float(d_e.item()); t2 = time.perf_counter()
            if it >= warmup:
                rows.append((t1 - t0, t2 - t1, t2 - t0))
        t = torch.tensor(rows, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # slowest rank per step
        med = t.median(dim=0).values.tolist()
        return {"ms_per_step": 1e3 * med[2], "sweep_ms": 1e3 * med[0], "allreduce_ms": 1e3 * med[1], "grid_points_per_sec": ngrid / med[2],
                "exc": exc, "hbm_frac": b_alg / med[2] / 1e9 / HBM_PEAK_GBS, "mfma_frac": f_alg / med[2] / 1e12 / F64_MFMA_PEAK_TF}

    b_alg, f_alg = kernel_model("xc_sweep", xc, ngrid, nao)
    abi = measure("dm")
    occ = measure("occ")
    del ao, gr, w, solver
    torch.cuda.empty_cache()
    res = {"workload": f"{workload}: ONE grid of {ngrid} points sharded over {world} GPU(s) ({n_loc} on rank 0), nao {nao}, nocc {nocc}, {xc}",
           "scaling": "strong", "entry_point": "DFT_ComputeXCOccAsync (occupied orbitals, dm = cocc cocc^T: the extension a driver that holds C calls)",
           "allreduce_payload_bytes": 8 * (nao * nao + 1), "steps": steps, "statistic": "median over steps of the max over ranks",
           "fractions_note": "hbm_frac / mfma_frac price the time against SURVEY 8(d)'s algorithmic bytes and flops of the dm form (4 ngrid nao^2 + ...), "
                             "whichever entry point ran: the occupied form executes fewer flops for the same result",
           "abi_dm_entry": dict(abi, entry_point="DFT_ComputeXCAsync (full density matrix, the reference ABI's contraction)")}
    res.update(occ)
    return res


def scf_sharded_leg(lib_path, dev, dist, backend, world, rank, workload="anthracene_b3lyp_def2tzvp", cycles=6, warmup=2):
    """One SCF cycle of the Anthracene B3LYP/def2-TZVP shape with the device work sharded over the ranks the way
    scf.HipBackend does it under torch.distributed: rank r keeps grid block shard_bounds(ngrid, N, r) and Cholesky
    vectors vector_bounds(naux, N, r) resident; per cycle local J/K + local XC sweep, ONE all-reduce of
    [Vxc | J | K | Exc] (3 nao^2 + 1 doubles), then rank 0 alone builds F, solves the eigenproblem (hipSOLVER syevd:
    a synthetic Fock sequence says nothing about the rotation solver) and forms dm, and ONE broadcast of [dm | cocc]
    brings the replicas back in step.  Synthetic planes and vectors (6 nao of them), device-resident."""
    from quantum_compute_dft_amd.grid_shard import shard_bounds, vector_bounds
    f64 = torch.float64
    xc, nao, ngrid = WORKLOADS[workload]
    n2, nocc = nao * nao, 47
    naux = 6 * nao
    lo, hi = shard_bounds(ngrid, world, rank)
    plo, phi = vector_bounds(naux, world, rank)
    n_loc, nv_loc = hi - lo, phi - plo
    dm, ao, gr, w, _ = synth(max(n_loc, 16), nao, True, dev, SEED + 104729 * (rank + 1))
    g = torch.Generator(device=dev); g.manual_seed(SEED + 31 * (rank + 1))
    chol = torch.randn((max(nv_loc, 1), nao, nao), dtype=f64, device=dev, generator=g) * 1e-2
    solver = q.DFTSolverWrapper(lib_path, xc)
    buf = torch.zeros(3 * n2 + 1, dtype=f64, device=dev)                            # [Vxc | J | K | Exc]
    d_v, d_J, d_K = (buf[k * n2:(k + 1) * n2].view(nao, nao) for k in range(3))
    d_e = buf[3 * n2:]
    state = torch.zeros(n2 + nao * nocc, dtype=f64, device=dev)                     # [dm | cocc]
    d_dm, d_c = state[:n2].view(nao, nao), state[n2:].view(nao, nocc)
    C = torch.linalg.qr(torch.randn((nao, nocc), dtype=f64, device=dev, generator=g))[0]
    d_c.copy_(C * float(np.sqrt(2.0))); d_dm.copy_(d_c @ d_c.T)
    H = torch.diag(torch.linspace(-1.0, 1.0, nao, dtype=f64, device=dev))

    def coll(fn, t):
        if world == 1:
            return
        if backend == "nccl":
            fn(t)
        else:
            h = t.cpu(); fn(h); t.copy_(h)

    coll(lambda t: dist.broadcast(t, 0), state)
    rows = []
    for it in range(warmup + cycles):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if nv_loc:
            solver.compute_jk_factorized(nao, nv_loc, nocc, chol, d_dm, d_c, d_J, d_K)
        else:
            d_J.zero_(); d_K.zero_()
        if n_loc:
            solver.compute_xc_async(n_loc, nao, d_dm, ao, w, d_v, d_e, gr)
        else:
            d_v.zero_(); d_e.zero_()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        coll(lambda t: dist.all_reduce(t), buf)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if rank == 0:
            F = H + 1e-3 * (d_J + 0.5 * (d_v + d_v.T) - 0.1 * d_K)
            e, Cf = torch.linalg.eigh(F)
            d_c.copy_(Cf[:, :nocc] * float(np.sqrt(2.0))); d_dm.copy_(d_c @ d_c.T)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        coll(lambda t: dist.broadcast(t, 0), state)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        if it >= warmup:
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    t = torch.tensor(rows, dtype=f64, device=dev if backend == "nccl" else "cpu")
    own = (1e3 * t.median(dim=0).values).tolist()       # this rank's parts (rank 0's are reported: a waiting rank books its wait as "broadcast")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    med = (1e3 * t.median(dim=0).values).tolist()
    del ao, gr, w, chol, solver
    torch.cuda.empty_cache()
    return {"workload": f"{workload} shape: nao {nao}, {ngrid} grid points and {naux} synthetic Cholesky vectors sharded over {world} GPU(s)",
            "scaling": "strong", "ms_per_cycle": med[4], "slowest_rank_device_jk_xc_ms": med[0],
            "parts_ms_rank0": {"device_jk_xc": own[0], "allreduce": own[1], "fock_eigh_density": own[2], "broadcast": own[3]},
            "allreduce_payload_bytes": 8 * (3 * n2 + 1), "broadcast_payload_bytes": 8 * (n2 + nao * nocc),
            "statistic": f"median over {cycles} cycles of the max over ranks"}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child
    (fresh processes; this one has not touched the GPU and never does) and leave with its exit code."""
    import socket
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL, shared tensors)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=80.0, help="untimed sustained load before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a full-size grid (default, the BASELINE metric); strong: one grid sharded over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-k-build", action="store_true", help="skip the factorised exact-exchange (fp64 MFMA) measurement")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip ao_sweep / scf_iteration / strong_config5 (kernel iteration runs)")
    ap.add_argument("--legs-seconds", type=float, default=420.0,
                    help="watchdog for everything after the headline measurement: the JSON line is printed with the legs finished by then")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo only to rehearse N>1 on a single card")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # Host BLAS/OpenMP pools on the CPU share from the first numpy call on: left at one thread per
    # visible core (256), the workers of a single BLAS call spin long enough after it to exhaust the
    # cgroup's CPU quota, and the whole process is throttled for tens of ms somewhere later (seen as
    # one 24-90 ms cycle among 0.7 ms ones).
    _pool_pin = blas_threads()
    _pool_pin.__enter__()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under torch.distributed.run with N ranks")
    lib_path = q.library_path()                          # never compiles: __graft_entry__.build() did
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    dev_index = local if args.backend == "nccl" else local % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    coll_name = "RCCL" if args.backend == "nccl" else "gloo (host; rehearsal only)"
    strong = args.scaling == "strong"
    workload = args.workload or ("c33_b3lyp_def2svp" if strong else "benzene_gga_def2svp")
    xc, nao, ngrid_all = WORKLOADS[workload]
    if strong:
        from quantum_compute_dft_amd.grid_shard import shard_bounds
        lo, hi = shard_bounds(ngrid_all, world, rank)
        ngrid = hi - lo
        total_points = ngrid_all
    else:
        ngrid, total_points = ngrid_all, world * ngrid_all
    dm, ao, gr, w, cocc = synth(max(ngrid, 16), nao, xc != "LDA", dev, SEED + rank)   # dm identical on all ranks
    if world > 1:
        for t_ in (dm, cocc):
            if args.backend == "nccl":
                dist.broadcast(t_, 0)
            else:
                h = t_.cpu(); dist.broadcast(h, 0); t_.copy_(h)
    solver = q.DFTSolverWrapper(lib_path, xc)
    out = torch.zeros(nao * nao + 1, dtype=torch.float64, device=dev)   # [Vxc | Exc]
    d_v, d_e = out[: nao * nao], out[nao * nao:]

    def step():
        if world == 1:
            return solver.compute_xc(ngrid, nao, dm, ao, w, d_v, gr)    # synchronous, returns Exc
        if ngrid:
            solver.compute_xc_async(ngrid, nao, dm, ao, w, d_v, d_e, gr)
        else:
            out.zero_()
        if args.backend == "nccl":
            dist.all_reduce(out)                                        # sum of the shard partials
        else:                                                           # rehearsal: gloo reduces on the host
            h = out.cpu(); dist.all_reduce(h); out.copy_(h)
        return float(d_e.item())                                        # device sync, like the ABI call

    nocc = cocc.shape[1]

    def step_occ():   # the same step through the extension entry point DFT_ComputeXCOcc (occupied orbitals, dm = cocc cocc^T)
        if world == 1:
            return solver.compute_xc_occ(ngrid, nao, nocc, cocc, ao, w, d_v, gr, dm)
        if ngrid:
            solver.compute_xc_occ_async(ngrid, nao, nocc, cocc, ao, w, d_v, d_e, gr, dm)
        else:
            out.zero_()
        if args.backend == "nccl":
            dist.all_reduce(out)
        else:
            h = out.cpu(); dist.all_reduce(h); out.copy_(h)
        return float(d_e.item())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock spin-up: the GPU raises its clocks over the first ~40 ms of sustained load (measured: 290-300 us
    # per call for the first 10 ms after idle, 243-248 us from 40 ms on), so an idle-start measurement of a
    # few ms quotes the ramp, not the engine.  Untimed, like the warm-up steps that follow.
    t_spin = time.perf_counter() + args.spinup_ms * 1e-3
    while True:
        for _ in range(8):
            exc = step()
        go = 1.0 if time.perf_counter() < t_spin else 0.0
        if world > 1:   # every rank must run the same number of (collective) steps: the decision is collective too
            flag = torch.tensor([go], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = float(flag.item())
        if go == 0.0:
            break
    for _ in range(args.warmup):
        exc = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        exc = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the same K steps through DFT_ComputeXCOcc (same inputs, same barrier / synchronise bracket, max over ranks)
    for _ in range(args.warmup):
        exc_occ = step_occ()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        exc_occ = step_occ()
    fence()
    dt_occ = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt_occ], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_occ = float(tt.item())

    # per-kernel durations: HIP events recorded by the library on its own stream, in a loop of its own (profile option on)
    def kernel_times(call):
        solver.set_option("profile", 1)
        acc = {}
        for i in range(max(args.steps, 10)):
            call()
            if i % 10 == 9:   # events are recorded on every call; read every tenth so the calls stay back to back
                for name, ms in solver.timings():
                    acc.setdefault(name, []).append(ms)
        solver.set_option("profile", 0)
        return {k: float(np.mean(v)) for k, v in acc.items()}

    kern, kern_occ = {}, {}
    if ngrid:
        kern = kernel_times(lambda: solver.compute_xc(ngrid, nao, dm, ao, w, d_v, gr))
        kern_occ = kernel_times(lambda: solver.compute_xc_occ(ngrid, nao, nocc, cocc, ao, w, d_v, gr, dm))

    # ---- the headline line is complete here; everything below only adds members to it ---------------------------
    line = None
    if rank == 0:
        t_step = dt / args.steps
        # whole call against SURVEY 8(d): B = ngrid(8 nao c + 8) + 16 nao^2, F = 4 ngrid nao^2 + (2c + 8[c=4]) ngrid nao
        b_all, f_all = kernel_model("xc_sweep", xc, ngrid, nao)
        hbm_frac, mfma_frac = b_all / t_step / 1e9 / HBM_PEAK_GBS, f_all / t_step / 1e12 / F64_MFMA_PEAK_TF
        if b_all / (HBM_PEAK_GBS * 1e9) >= f_all / (F64_MFMA_PEAK_TF * 1e12):
            roof = {"bound": "hbm", "achieved": b_all / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac}
        else:
            roof = {"bound": "mfma", "achieved": f_all / t_step / 1e12, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": mfma_frac}
        roof.update({"scope": "whole DFT_ComputeXC call (all its kernels + the host's wait), algorithmic bytes/flops of SURVEY 8(d) / ms_per_step",
                     "hbm_frac": hbm_frac, "mfma_frac": mfma_frac, "alg_bytes": b_all, "alg_flops": f_all,
                     "kernel_sum_ms_profiled_loop": float(sum(kern.values())),
                     "kernel_sum_note": "HIP events of a separate loop run with the library's profile option on (an event pair per kernel): "
                                        "not the timed loop, so it may exceed ms_per_step by a few us"})
        tr = pmc_traffic(workload, list(kern))
        roof["traffic"] = float(sum(tr["per_kernel"].values())) if tr else None
        if tr:
            roof["traffic_source"] = "profiles/: " + tr["source"]
            roof["traffic_kernels_covered"] = tr["covered"]
        if kern:
            dom = max(kern, key=kern.get)
            b_alg, f_alg = kernel_model(dom, xc, ngrid, nao)
            t_dom = kern[dom] * 1e-3
            roof["dominant_kernel"] = {"kernel": dom, "kernel_ms": kern[dom], "alg_bytes": b_alg, "alg_flops": f_alg,
                                       "hbm_frac": b_alg / t_dom / 1e9 / HBM_PEAK_GBS, "mfma_frac": f_alg / t_dom / 1e12 / F64_MFMA_PEAK_TF,
                                       "traffic": tr["per_kernel"].get(dom) if tr else None,
                                       "mfma_pipe_busy_pmc": tr["mfma_utilisation"].get(dom) if tr else None,
                                       "mfma_pipe_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES over all SIMD cycles, rocprofv3 counter pass (profiles/r03_pmc_sweep.json)"}
        line = {
            "metric": "grid_points_per_sec", "value": total_points * args.steps / dt, "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_step * 1e3, "spinup_ms": args.spinup_ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload}: DFT_ComputeXC ({xc}) nao={nao} ngrid={ngrid_all}" + (" per GPU" if not strong else f" sharded over {world} GPU(s)") +
                                   ", synthetic AO/grid (SURVEY 8(d) recipe), inputs resident in HBM",
                       "functional": xc, "nao": nao, "nocc": nocc, "ngrid_per_gpu": ngrid,
                       "entry_point": "DFT_ComputeXC (the reference ABI: full density matrix)" if world == 1 else "DFT_ComputeXCAsync (full density matrix) + all-reduce",
                       "sharding": "grid points" + ("" if world == 1 else f" x{world}, {coll_name} all-reduce of Vxc|Exc ({8 * (nao * nao + 1)} B)")},
            "backend": args.backend if world > 1 else None,
            "world_size_seen": (dist.get_world_size() if world > 1 else 1), "device_count": ndev,
            "roofline": roof, "kernels_ms": kern, "exc": exc,
            # the same steps through the extension that takes the occupied orbitals (what scf.HipBackend calls): same
            # results, the density step does 4 nao nocc instead of 2 nao^2 flops per grid point
            "occ_entry": {"entry_point": "DFT_ComputeXCOcc (occupied orbitals cocc (nao, nocc), dm = cocc cocc^T)" + ("" if world == 1 else " async + all-reduce"),
                          "ms_per_step": 1e3 * dt_occ / args.steps, "value": total_points * args.steps / dt_occ, "unit": "grid-points/s",
                          "exc": exc_occ, "exc_rel_diff_to_abi": abs(exc_occ - exc) / max(abs(exc), 1e-300), "kernels_ms": kern_occ,
                          "hbm_frac": b_all / (dt_occ / args.steps) / 1e9 / HBM_PEAK_GBS,
                          "note": "hbm_frac against the same algorithmic bytes as the ABI call; the occupied form reads the AO plane once"},
        }

    # The extra legs must never cost the headline: a watchdog thread ends the process after `args.legs_seconds` -- rank 0
    # printing the line with whatever legs have finished -- and a leg that raises is recorded and ends the extra legs
    # (with N > 1 the ranks' collectives would be out of step after it).  The thread runs while the main thread waits
    # inside HIP / RCCL (torch drops the GIL there).
    import threading
    printed = threading.Lock()
    line_lock = threading.Lock()       # members are added under it; the dump takes a copy under it

    def put(key, value):
        if os.environ.get("QCDFT_BENCH_TRACE"):
            print(f"[bench rank {rank}] leg done: {key}", file=sys.stderr, flush=True)
        if line is not None:
            with line_lock:
                line[key] = value

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return
        if line is not None:
            with line_lock:
                snap = dict(line)
            if note:
                snap["legs_note"] = note
            print(json.dumps(snap), flush=True)

    def expire():
        # a leg that hangs is a failure the harness must see: the headline line is printed, then the process ends NON-ZERO
        emit(f"extra legs cut off after {args.legs_seconds:.0f} s (exit code 3): members present are complete, the others are missing")
        os._exit(3)

    dog = threading.Timer(args.legs_seconds, expire)
    dog.daemon = True
    dog.start()
    note = None
    try:
        if not strong and not args.no_extra_legs:            # every rank takes part (collectives inside)
            del ao, gr
            ao = gr = None
            torch.cuda.empty_cache()
            strong5 = strong_leg(lib_path, dev, dist, args.backend, world, rank, "c33_b3lyp_def2svp")
            put("strong_config5", strong5)
            if world > 1:
                scf_sh = scf_sharded_leg(lib_path, dev, dist, args.backend, world, rank)
                put("scf_iteration_anthracene_sharded", scf_sh)
            scf_an = scf_real_sharded_leg(lib_path, dev, world, rank)
            put("scf_anthracene_def2svp_real", scf_an)
            if world == 1:
                dm, ao, gr, w, cocc = synth(ngrid, nao, xc != "LDA", dev, SEED + rank)   # the same inputs again for the legs below
                put("ao_sweep", ao_sweep_leg(lib_path, dev))
                put("small_basis", small_basis_leg(lib_path, dev))
                # ms/SCF-iter of the BASELINE metric = the driver's own loop on the real molecules (energies as checksums)
                real = scf_real_leg(lib_path, dev)
                put("scf_benzene_real", real)
                for key, src, what in (("scf_iteration", "dense_eri_auto", "dense ERI (the reference's formulation, dft.py:166,203)"),
                                       ("scf_iteration_factorised_j", "factorised_j_auto", "factorised J (Cholesky vectors, 1e-8)")):
                    r = real[src]
                    put(key, {"ms": r["ms_per_cycle"], "workload": real["workload"] + "; " + what, "statistic": real["statistic"],
                                 "loop": ("fused: Fock assembly, DIIS, occupied-subspace rotation, density and energy traces as kernels of libdft.so behind the "
                                          "cycle's J/K and sweep (DFT_ScfTailStep), one host wait per cycle; xc / jk are device-side durations (events)"
                                          if r.get("loop") == "fused" else r.get("loop")),
                                 "parts_ms": {"xc": r["xc_ms"], "jk": r["jk_ms"], "fock_diis_eigen_density_and_waits": r["ms_per_cycle"] - r["xc_ms"] - r["jk_ms"]},
                                 "cycles": r["cycles"], "converged": r["converged"], "E_tot": r["E_tot"], "eigensolver": r["eigensolver"],
                                 "eigh_every_cycle_ms": real[src.replace("_auto", "_exact")]["ms_per_cycle"],
                                 "host_loop": real.get(src.replace("_auto", "_host_loop")),
                                 "xc_entry_point": "DFT_ComputeXCOcc (the driver holds cocc; scf.HipBackend default)",
                                 "abi_xc_entry": real.get(src.replace("_auto", "_abi_xc"))})
                # the same loop body with synthetic operands and eigh(F, S) EVERY cycle (round 1's legs, kept for comparison)
                put("scf_iteration_synthetic", scf_iteration_ms(solver, xc, nao, ngrid, dm, ao, gr, w, dev))
                put("scf_iteration_synthetic_factorised_j", scf_iteration_ms(solver, xc, nao, ngrid, dm, ao, gr, w, dev, eri="cholesky"))
                del ao, gr
                torch.cuda.empty_cache()
                xa, na, ga = WORKLOADS["anthracene_b3lyp_def2tzvp"]
                dm_a, ao_a, gr_a, w_a, _ = synth(ga, na, True, dev, SEED)
                put("scf_iteration_anthracene_synthetic", scf_iteration_ms(q.DFTSolverWrapper(lib_path, xa), xa, na, ga, dm_a, ao_a, gr_a, w_a, dev, iters=7))
                del ao_a, gr_a
                torch.cuda.empty_cache()
                # BASELINE config 3 through the driver: Anthracene B3LYP in the def2-TZVP-shaped basis (nao 494; ~12 s of host
                # integrals for its Cholesky vectors, hence last among the SCF legs)
                an = scf_real_sharded_leg(lib_path, dev, 1, 0, basis_name="def2-tzvp", tol=1e-7)
                put("scf_iteration_anthracene", dict(an, ms=an["ms_per_cycle"]))
                dm, ao, gr, w, cocc = synth(ngrid, nao, xc != "LDA", dev, SEED + rank)
        if world == 1 and not args.no_k_build:
            put("k_build", k_build_mfma(lib_path, dev))
            put("k_build_nocc60", k_build_mfma(lib_path, dev, nocc=60))   # the 49-64 orbital tile of the half transform (MI = 4: two workgroups per CU since round 3)
        if world == 1 and not args.no_cpu_baseline:
            if ao is None:
                dm, ao, gr, w, cocc = synth(ngrid, nao, xc != "LDA", dev, SEED + rank)
            put("cpu_baseline", cpu_baseline(xc, dm, ao, gr, w, args.cpu_seconds))
    except Exception as e:   # noqa: BLE001 -- whatever a leg raises, the measured headline is still printed
        note = f"an extra leg failed ({type(e).__name__}: {e}); members present are complete"
        import traceback
        print(f"[bench rank {rank}] {note}", file=sys.stderr, flush=True)   # every rank says what it saw (only rank 0 owns the line)
        traceback.print_exc(file=sys.stderr)
        sys.stderr.flush()
    dog.cancel()
    emit(note)
    if note is not None:
        # a leg raised: the headline line is out, the failure is visible to the harness through the exit code
        # (with N > 1 the ranks may be out of step: no further collective, no clean shutdown)
        sys.stdout.flush()
        os._exit(4)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
