"""The SCF loop of the reference driver (dft.py:181-266) with the device work behind a backend.

Loop contract kept exactly: core guess `eigh(Hcore, S)`; per cycle J (compute_coulomb), XC
(compute_xc), `Vxc = (V + V^T)/2` (dft.py:212), for B3LYP K and `F = H + J + Vxc - 0.5*0.2*K`
(dft.py:217-221), DIIS on (S, dm, F), `eigh(F, S)`, energies from the NEW density with J/K/Exc of
the old one (dft.py:230-236), convergence |dE| < 1e-8 and ||d dm||_F < 1e-6 (dft.py:243),
200 cycles.  Differences: J and K come from ONE pass over the ERI (DFT_ComputeJK), AO values
are evaluated on the device, nothing but dm / J / K / Vxc (nao^2 each) crosses PCIe per cycle.
"""
import time

import numpy as np
from scipy.linalg import eigh


class CDIIS:
    """Pulay DIIS on the commutator SDF - FDS (PySCF scf.diis.CDIIS as used at dft.py:184,225)."""

    def __init__(self, space=8):
        self.space, self.F, self.e = space, [], []

    def update(self, S, dm, F):
        sdf = S @ dm @ F
        self.F.append(F.copy()); self.e.append((sdf.T - sdf).ravel())
        if len(self.F) > self.space:
            self.F.pop(0); self.e.pop(0)
        n = len(self.F)
        if n < 2:
            return F
        B = np.zeros((n + 1, n + 1)); B[0, 1:] = B[1:, 0] = 1.0
        for i in range(n):
            for j in range(i + 1):
                B[i + 1, j + 1] = B[j + 1, i + 1] = self.e[i] @ self.e[j]
        rhs = np.zeros(n + 1); rhs[0] = 1.0
        try:
            c = np.linalg.solve(B, rhs)[1:]
        except np.linalg.LinAlgError:
            c = np.linalg.lstsq(B, rhs, rcond=None)[0][1:]
        return sum(ci * Fi for ci, Fi in zip(c, self.F))


class FockDiagonaliser:
    """F C = S C e, the one dense eigenproblem of an SCF cycle (dft.py:181,227 `eigh(F, S)` on the host).
    Small matrices stay on the host (LAPACK through scipy, as the reference does); from `device_from`
    basis functions on, F is orthogonalised with X = U s^-1/2 (once per S) and diagonalised on the GPU
    by hipSOLVER through torch.linalg.eigh -- the only library call on the device path (measured on
    MI355X + 16 host cores: n=246 host 4.4 ms / device 6.1 ms, n=494 16.6 / 11.4 ms, n=1150 84 / 27 ms,
    tools/eigh_time.py)."""

    def __init__(self, S, device=None, device_from=400):
        self.S, self.n = S, S.shape[0]
        self.on_device = device is not None and self.n >= device_from
        if self.on_device:
            import torch
            self.torch = torch
            s, U = np.linalg.eigh(S)
            self.X = torch.as_tensor(U / np.sqrt(s), dtype=torch.float64, device=device)

    def __call__(self, F):
        if not self.on_device:
            return eigh(F, self.S)
        t = self.torch
        Fd = t.as_tensor(F, dtype=t.float64, device=self.X.device)
        e, Cp = t.linalg.eigh(self.X.T @ Fd @ self.X)
        return e.cpu().numpy(), (self.X @ Cp).cpu().numpy()


class HipBackend:
    """Device side of the loop: libdft.so through DFTSolverWrapper, torch tensors as buffers."""

    def __init__(self, inp, functional, lib_path=None, quirks=True):
        import torch
        from .solver import DFTSolverWrapper
        assert torch.cuda.is_available(), "the SCF driver needs a GPU (there is no CPU fallback)"
        self.torch, self.dev = torch, torch.device("cuda")
        self.functional = functional.upper()
        self.solver = DFTSolverWrapper(lib_path, self.functional)
        self.solver.set_option("quirks", 1 if quirks else 0)
        t0 = time.time()
        nao, ngrid = inp.shells.nao, inp.grids.size
        self.nao, self.ngrid = nao, ngrid
        f64 = torch.float64
        d_coords = torch.as_tensor(inp.grids.coords, dtype=f64, device=self.dev)
        self.d_w = torch.as_tensor(inp.grids.weights, dtype=f64, device=self.dev)
        self.d_ao = torch.empty((ngrid, nao), dtype=f64, device=self.dev)
        self.d_gr = torch.empty((3, ngrid, nao), dtype=f64, device=self.dev) if self.functional != "LDA" else None
        self.solver.eval_ao(inp.shells, d_coords, ngrid, self.d_ao, self.d_gr)      # grid.py:30,38 on the device
        self.d_eri = self.d_chol = self.d_cocc = None
        if inp.eri is not None:
            self.d_eri = torch.as_tensor(inp.eri.reshape(nao * nao, nao * nao), dtype=f64, device=self.dev)  # dft.py:166
        else:  # factorised J/K (DFT_ComputeJKFactorized): Cholesky vectors stay resident instead of the ERI
            self.d_chol = torch.as_tensor(inp.chol, dtype=f64, device=self.dev)
            self.d_cocc = torch.zeros((nao, inp.nocc), dtype=f64, device=self.dev)
        self.nocc = inp.nocc
        self.d_dm = torch.zeros((nao, nao), dtype=f64, device=self.dev)
        self.d_J = torch.zeros_like(self.d_dm); self.d_K = torch.zeros_like(self.d_dm); self.d_v = torch.zeros_like(self.d_dm)
        self.eigh = FockDiagonaliser(inp.S, self.dev)
        torch.cuda.synchronize()
        self.init_time = time.time() - t0

    def set_dm(self, dm):
        self.d_dm.copy_(self.torch.as_tensor(dm, dtype=self.torch.float64))       # dft.py:200

    def set_cocc(self, cocc):
        """cocc (nao, nocc) with dm = cocc cocc^T; only the factorised exchange needs it."""
        if self.d_cocc is not None:
            self.d_cocc.copy_(self.torch.as_tensor(np.ascontiguousarray(cocc), dtype=self.torch.float64))

    def jk(self, want_k):
        if self.d_chol is not None:
            self.solver.compute_jk_factorized(self.nao, self.d_chol.shape[0], self.nocc, self.d_chol, self.d_dm,
                                              self.d_cocc if want_k else None, self.d_J, self.d_K if want_k else None)
            return self.d_J.cpu().numpy(), (self.d_K.cpu().numpy() if want_k else None)
        if want_k:
            self.solver.compute_jk(self.nao, self.d_eri, self.d_dm, self.d_J, self.d_K)
            return self.d_J.cpu().numpy(), self.d_K.cpu().numpy()
        self.solver.compute_coulomb(self.nao, self.d_eri, self.d_dm, self.d_J)        # dft.py:203
        return self.d_J.cpu().numpy(), None

    def xc(self):
        t0 = time.time()
        exc = self.solver.compute_xc(self.ngrid, self.nao, self.d_dm, self.d_ao, self.d_w, self.d_v, self.d_gr)
        self.torch.cuda.synchronize()                                                # dft.py:205-208
        return exc, self.d_v.cpu().numpy(), time.time() - t0


def run_scf(inp, backend, functional, max_cycle=200, conv_e=1e-8, conv_dm=1e-6, log=print):
    from .hostinfo import blas_threads
    with blas_threads():   # host eigh / DIIS on the CPU share, not on every visible core
        return _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log)


def _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log):
    functional = functional.upper()
    c_hf = 0.2 if functional == "B3LYP" else 0.0                                       # dft.py:197
    Hcore, S, nocc = inp.Hcore, inp.S, inp.nocc
    solve = getattr(backend, "eigh", None) or (lambda F: eigh(F, S))
    e, C = solve(Hcore)                                                                # dft.py:181
    dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T
    set_cocc = getattr(backend, "set_cocc", None)
    diis = CDIIS()
    if log:
        log("\nSCF started!"); log("-" * 80)
        log(f"{'epoch':>4} {'tot energy':>15} {'Δenergy':>12} {'Δdensity':>12} {'HF_Ex':>12}"); log("-" * 80)
    E_old, xc_times, jk_times, it_times, t_start = 0.0, [], [], [], time.time()
    res = {"converged": False}
    for cycle in range(max_cycle):
        t_it = time.time()
        backend.set_dm(dm)
        if set_cocc:
            set_cocc(np.sqrt(2.0) * C[:, :nocc])
        J, K = backend.jk(functional == "B3LYP")
        jk_times.append(time.time() - t_it)
        E_xc, Vraw, t_xc = backend.xc()
        xc_times.append(t_xc)
        Vxc = 0.5 * (Vraw + Vraw.T)                                                    # dft.py:212
        F = Hcore + J + Vxc - (c_hf * 0.5 * K if K is not None else 0.0)               # dft.py:221,223
        F = diis.update(S, dm, F)
        e, C = solve(F)
        dm_new = 2.0 * C[:, :nocc] @ C[:, :nocc].T
        E_one = float(np.sum(dm_new * Hcore)); E_coul = 0.5 * float(np.sum(dm_new * J))
        E_ex = -0.25 * c_hf * float(np.sum(dm_new * K)) if K is not None else 0.0
        E_tot = E_one + E_coul + E_xc + E_ex + inp.E_nuc                               # dft.py:236
        dE, ddm = E_tot - E_old, float(np.linalg.norm(dm_new - dm))
        it_times.append(time.time() - t_it)
        if log:
            log(f"{cycle + 1:4d} {E_tot:18.8f} {dE:15.6e} {ddm:15.6e} {E_ex:12.6f}")
        res.update(E_tot=E_tot, E_one=E_one, E_coul=E_coul, E_xc=E_xc, E_ex_hf=E_ex, cycles=cycle + 1,
                   dm=dm_new, mo_energy=e)
        if abs(dE) < conv_e and ddm < conv_dm:                                         # dft.py:243
            res["converged"] = True
            break
        dm, E_old = dm_new, E_tot
    res["total_time"] = time.time() - t_start
    res["xc_ms_avg"] = 1e3 * sum(xc_times) / max(1, len(xc_times))     # dft.py:259 (includes the first call's allocations)
    steady = lambda ts: 1e3 * float(np.median(ts[1:] if len(ts) > 1 else ts))
    res["xc_ms"], res["jk_ms"], res["iter_ms"] = steady(xc_times), steady(jk_times), steady(it_times)  # medians past cycle 1
    res["nelec_grid"] = None
    return res
