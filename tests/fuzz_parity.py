"""Randomised parity sweep on the GPU: DFT_ComputeXC / DFT_ComputeXCOcc (all functionals, all kernel paths), dense and factorised
J/K, AO evaluation -- each against the oracle on random sizes.  usage: python tests/fuzz_parity.py [seconds] [seed]  (a checker like the tests next to it: the only places the oracle is used from)"""
import sys, time, numpy as np, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import oracle
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.hostinfo import blas_threads
_pin = blas_threads(); _pin.__enter__()   # host pools on the CPU share (hostinfo.py): no quota-throttling stalls in the timings
from quantum_compute_dft_amd import basis
from helpers import synth_inputs
dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
t_end = time.time() + budget
n_xc = n_jk = n_cd = n_ao = n_direct = n_occ = 0
worst = {"xc_e": 0.0, "xc_v": 0.0, "jk": 0.0, "cd": 0.0, "ao": 0.0}
names = ["LDA", "GGA", "B3LYP"]
while time.time() < t_end:
    kind = rng.integers(0, 10)
    if kind < 6:   # XC sweep
        xc = int(rng.integers(0, 3)); nao = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 24, 31, 32, 33, 64, 80, 113, 114, 127, 128, 129, 130, 160, 200, 257]))
        ngrid = int(rng.choice([1, 5, 15, 16, 17, 31, 100, 255, 256, 257, 1000, 2049, 4097, 9000]))
        if nao * ngrid > 1.2e6: ngrid = max(1, int(1.2e6 // nao))
        path = int(rng.choice([0, 0, 0, 1, 2])); quirks = int(rng.integers(0, 2))
        dm, ao, gr, w = synth_inputs(ngrid, nao, seed=int(rng.integers(1 << 30)))
        occ_call = path == 0 and rng.random() < 0.5      # DFT_ComputeXCOcc: the same sweep through the occupied orbitals
        if occ_call:
            nocc = int(min(nao, rng.choice([1, 2, 5, 15, 16, 17, 21, 33, 47, 64, 65, 100, 129])))
            cocc = np.sqrt(2.0) * 0.7 * np.random.default_rng(int(rng.integers(1 << 30))).standard_normal((nao, nocc))
            dm = cocc @ cocc.T
        if rng.random() < 0.3: w[rng.integers(0, ngrid, size=max(1, ngrid // 7))] = 0.0
        e_ref, v_ref = oracle.compute_xc(xc, dm, ao, w, gr if xc else None, quirks=bool(quirks))
        s = q.DFTSolverWrapper(q.library_path(), names[xc]); s.set_option("path", path); s.set_option("quirks", quirks)
        if rng.random() < 0.5: s.set_option("rho_rows", 128)
        if rng.random() < 0.3: s.set_option("ws_waves", 16)
        if nao <= 32 and rng.random() < 0.4: s.set_option("tiny", int(rng.choice([0, 1])))   # the one-pass kernel forced on / off (default: auto)
        s.set_option("sweep_order", int(rng.integers(0, 4)))
        d_v = torch.full((nao, nao), 3.0, dtype=torch.float64, device=dev)
        if occ_call:
            s.set_option("occ", int(rng.choice([0, 1, 1])))
            e = s.compute_xc_occ(ngrid, nao, nocc, t(cocc), t(ao), t(w), d_v, t(gr) if xc else None, t(dm) if rng.random() < 0.5 else None)
            n_occ += 1
        else:
            e = s.compute_xc(ngrid, nao, t(dm), t(ao), t(w), d_v, t(gr) if xc else None)
        ee = abs(e - e_ref) / max(1e-300, abs(e_ref)) if e_ref else abs(e)
        ve = np.abs(d_v.cpu().numpy() - v_ref).max() / max(1e-300, np.abs(v_ref).max())
        worst["xc_e"] = max(worst["xc_e"], ee); worst["xc_v"] = max(worst["xc_v"], ve); n_xc += 1
        # the synthetic grids contain unphysical low-density / high-gradient points where LYP and PBE are
        # ill-conditioned in rho: for the worst inputs met, scaling dm by (1 + 1e-15) moves the ORACLE's V
        # by 7e-8 relative, and all three kernel paths then differ from the oracle by that same amount.
        # A kernel bug would differ between paths and sit orders of magnitude above this bound.
        if ve > 1e-11: print(f"  note: XC {names[xc]} nao={nao} ngrid={ngrid} path={path} quirks={quirks}: dE {ee:.1e} dV {ve:.1e}", flush=True)
        if ve > 2e-9:   # look closer: the same inputs on every path, and how sensitive the oracle itself is
            for p2 in (0, 1, 2):
                s2 = q.DFTSolverWrapper(q.library_path(), names[xc]); s2.set_option("path", p2); s2.set_option("quirks", quirks)
                v2 = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
                e2 = s2.compute_xc(ngrid, nao, t(dm), t(ao), t(w), v2, t(gr) if xc else None)
                print(f"    path {p2}: dV {np.abs(v2.cpu().numpy() - v_ref).max() / np.abs(v_ref).max():.1e}", flush=True)
            _, v3 = oracle.compute_xc(xc, dm * (1 + 1e-15), ao, w, gr if xc else None, quirks=bool(quirks))
            _, v4 = oracle.compute_xc(xc, dm, ao * (1 + 1e-15), w, gr if xc else None, quirks=bool(quirks))
            print(f"    oracle sensitivity: dm*(1+1e-15) -> dV {np.abs(v3 - v_ref).max() / np.abs(v_ref).max():.1e}; ao*(1+1e-15) -> {np.abs(v4 - v_ref).max() / np.abs(v_ref).max():.1e}", flush=True)
            import os; os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fuzz_case.npz", dm=dm, ao=ao, gr=gr, w=w, xc=xc, quirks=quirks, path=path)
        assert ee < 1e-9 and ve < 1e-6, ("XC", names[xc], nao, ngrid, path, quirks, ee, ve)
    elif kind < 7:   # dense J/K
        n = int(rng.choice([1, 2, 5, 7, 12, 24, 36, 41])); n2 = n * n
        eri = rng.normal(size=(n2, n2)); dm = rng.normal(size=(n, n))
        s = q.DFTSolverWrapper(q.library_path(), "B3LYP")
        d_J = torch.zeros((n, n), dtype=torch.float64, device=dev); d_K = torch.zeros_like(d_J)
        s.compute_jk(n, t(eri), t(dm), d_J, d_K); torch.cuda.synchronize()
        J_ref, K_ref = oracle.coulomb(eri, dm), oracle.exchange(eri, dm)
        err = max(np.abs(d_J.cpu().numpy() - J_ref).max() / np.abs(J_ref).max(), np.abs(d_K.cpu().numpy() - K_ref).max() / np.abs(K_ref).max())
        worst["jk"] = max(worst["jk"], err); n_jk += 1
        assert err < 1e-11, ("JK", n, err)
    elif kind < 9:   # factorised J/K
        nao = int(rng.choice([1, 3, 16, 17, 31, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300])); naux = int(rng.choice([1, 2, 7, 33, 100]))
        nocc = int(rng.choice([1, 2, 15, 16, 17, 33, 48, 49, 64, 65, 90])); nocc = min(nocc, nao)
        A = rng.normal(0, 0.3, (naux, nao, nao)); chol = 0.5 * (A + A.transpose(0, 2, 1))
        cocc = rng.normal(0, 0.7, (nao, nocc)); dm = cocc @ cocc.T
        J_ref, K_ref = oracle.jk_from_factors(chol, dm)
        s = q.DFTSolverWrapper(q.library_path(), "B3LYP")
        d_J = torch.zeros((nao, nao), dtype=torch.float64, device=dev); d_K = torch.zeros_like(d_J)
        s.compute_jk_factorized(nao, naux, nocc, t(chol), t(dm), t(cocc), d_J, d_K); torch.cuda.synchronize()
        err = max(np.abs(d_J.cpu().numpy() - J_ref).max() / np.abs(J_ref).max(), np.abs(d_K.cpu().numpy() - K_ref).max() / np.abs(K_ref).max())
        worst["cd"] = max(worst["cd"], err); n_cd += 1
        assert err < 1e-11, ("CD", nao, naux, nocc, err)
    else:   # AO evaluation
        natm = int(rng.integers(1, 12)); syms = list(rng.choice(["H", "C", "N", "O"], size=natm))
        xyz = rng.uniform(-4, 4, (natm, 3)); bname = str(rng.choice(["sto-3g", "def2-svp", "def2-svp"]))
        if bname == "def2-svp" and rng.random() < 0.3 and all(sy in ("H", "C") for sy in syms): bname = "def2-tzvp"
        sh = basis.build_shells(syms, xyz, bname)
        ngrid = int(rng.choice([1, 7, 8, 9, 16, 17, 100, 1000, 2049])); deriv = int(rng.integers(0, 2))
        coords = rng.uniform(-6, 6, (ngrid, 3)); coords[0] = xyz[0]
        ref = oracle.eval_ao(sh, coords, deriv=deriv)
        s = q.DFTSolverWrapper(q.library_path(), "GGA"); s.set_option("ao_pt", int(rng.choice([0, 8, 16])))
        d_ao = torch.full((ngrid, sh.nao), 9.0, dtype=torch.float64, device=dev)
        d_gr = torch.full((3, ngrid, sh.nao), 9.0, dtype=torch.float64, device=dev) if deriv else None
        assert s.eval_ao(sh, t(coords), ngrid, d_ao, d_gr) == 0; torch.cuda.synchronize()
        a_ref = ref[0] if deriv else ref
        err = np.abs(d_ao.cpu().numpy() - a_ref).max() / max(1.0, np.abs(a_ref).max())
        if deriv: err = max(err, np.abs(d_gr.cpu().numpy() - ref[1]).max() / max(1.0, np.abs(ref[1]).max()))
        worst["ao"] = max(worst["ao"], err); n_ao += 1
        assert err < 1e-12, ("AO", bname, natm, ngrid, deriv, sh.nao, err)
        if deriv:   # the direct sweep (planes re-evaluated chunk by chunk) against the resident-plane call on the same shells
            xc = int(rng.integers(0, 3)); nao = sh.nao
            C = rng.normal(0, 0.4, (nao, max(1, nao // 4))); dm = 2.0 * C @ C.T
            w = np.abs(rng.normal(0.1, 0.05, ngrid))
            sx = q.DFTSolverWrapper(q.library_path(), names[xc])
            d_v, d_v2, d_e = (torch.full(sz, 5.0, dtype=torch.float64, device=dev) for sz in ((nao, nao), (nao, nao), (1,)))
            e0 = sx.compute_xc(ngrid, nao, t(dm), d_ao, t(w), d_v, d_gr if xc else None)
            sx.compute_xc_direct(sh, ngrid, t(coords), t(w), t(dm), d_v2, d_e, int(rng.choice([0, 1, 16, 256, 1024, 5000])))
            torch.cuda.synchronize()
            ed = abs(float(d_e.item()) - e0) / max(1e-300, abs(e0))
            vd = float((d_v2 - d_v).abs().max()) / max(1e-300, float(d_v.abs().max()))
            worst["direct"] = max(worst.get("direct", 0.0), ed, vd); n_direct += 1
            assert ed < 1e-11 and vd < 1e-10, ("direct", bname, natm, ngrid, xc, ed, vd)
print(f"fuzz ok: {n_xc} XC sweeps ({n_occ} of them through DFT_ComputeXCOcc), {n_jk} dense J/K, {n_cd} factorised J/K, {n_ao} AO evaluations, {n_direct} direct sweeps; worst relative errors {worst}")
