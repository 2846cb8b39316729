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
    """Pulay DIIS on the commutator SDF - FDS (PySCF scf.diis.CDIIS as used at dft.py:184,225).
    With `device` the n^3 products and the history live on the GPU (rocBLAS through torch; only the
    (space+1)^2 system is solved on the host): n = 494 costs 2.3 ms per cycle on 16 host cores."""

    def __init__(self, space=8, device=None):
        self.space, self.F, self.e, self.dev = space, [], [], device
        if device is not None:
            import torch
            self.torch = torch
            self._S = None

    def _error(self, S, dm, F):
        if self.dev is None:
            sdf = S @ dm @ F
            return F.copy(), (sdf.T - sdf).ravel()
        t = self.torch
        if self._S is None:
            self._S = t.as_tensor(S, dtype=t.float64, device=self.dev)
        Fd = t.as_tensor(F, dtype=t.float64, device=self.dev)
        sdf = self._S @ t.as_tensor(dm, dtype=t.float64, device=self.dev) @ Fd
        return Fd, (sdf.T - sdf).reshape(-1)

    def update(self, S, dm, F):
        Fk, ek = self._error(S, dm, F)
        self.F.append(Fk); self.e.append(ek)
        if len(self.F) > self.space:
            self.F.pop(0); self.e.pop(0)
            if self.dev is not None:
                self._G = self._G[1:, 1:]
        n = len(self.F)
        if self.dev is not None:   # Gram matrix of the error vectors: only the new row (one GEMV on the GPU)
            row = (self.torch.stack(self.e) @ ek).cpu().numpy()
            G = np.zeros((n, n))
            G[:n - 1, :n - 1] = getattr(self, "_G", np.zeros((0, 0)))[:n - 1, :n - 1]
            G[n - 1, :] = G[:, n - 1] = row
            self._G = G
        if n < 2:
            return F
        B = np.zeros((n + 1, n + 1)); B[0, 1:] = B[1:, 0] = 1.0
        if self.dev is None:
            for i in range(n):
                for j in range(i + 1):
                    B[i + 1, j + 1] = B[j + 1, i + 1] = self.e[i] @ self.e[j]
        else:
            B[1:, 1:] = self._G
        rhs = np.zeros(n + 1); rhs[0] = 1.0
        try:
            c = np.linalg.solve(B, rhs)[1:]
        except np.linalg.LinAlgError:
            c = np.linalg.lstsq(B, rhs, rcond=None)[0][1:]
        if self.dev is None:
            return sum(ci * Fi for ci, Fi in zip(c, self.F))
        out = self.torch.zeros_like(self.F[0])
        for ci, Fi in zip(c, self.F):
            out.add_(Fi, alpha=float(ci))
        return out.cpu().numpy()


class FockDiagonaliser:
    """F C = S C e, the one dense eigenproblem of an SCF cycle (dft.py:181,227 `eigh(F, S)` on the host).
    F is orthogonalised with X = U s^-1/2 (once per S).  Small matrices stay on the host: LAPACK dsyevd
    on ONE thread (measured on the MI355X box, tools/eigh_threads.py: n = 114 0.67 ms on one thread
    against 1.0 ms for scipy's generalised driver on 16; n = 246 3.2 against 4.9 ms).  From
    `device_from` basis functions on the problem goes to the GPU: hipSOLVER through
    torch.linalg.eigh, the only library call on the device path (n = 494: 11.4 ms against 16.5 ms on
    16 host cores, n = 1150: 27 against 84 ms, tools/eigh_time.py)."""

    def __init__(self, S, device=None, device_from=400):
        self.S, self.n = S, S.shape[0]
        self.on_device = device is not None and self.n >= device_from
        s, U = np.linalg.eigh(S)
        self.Xh = U / np.sqrt(s)
        if self.on_device:
            import torch
            self.torch = torch
            self.X = torch.as_tensor(self.Xh, dtype=torch.float64, device=device)

    def __call__(self, F):
        if not self.on_device:
            e, Cp = eigh(self.Xh.T @ F @ self.Xh, driver="evd")   # run_scf pins the BLAS pool to one thread at this size
            return e, self.Xh @ Cp
        t = self.torch
        Fd = t.as_tensor(F, dtype=t.float64, device=self.X.device)
        e, Cp = t.linalg.eigh(self.X.T @ Fd @ self.X)
        return e.cpu().numpy(), (self.X @ Cp).cpu().numpy()


class RefinedDiagonaliser:
    """F C = S C e by iterative refinement of the previous cycle's eigenvectors (Ogita & Aishima 2018):
    with X the approximate eigenvectors of F' = Xo^T F Xo,

        R = I - X^T X,   S = X^T F' X,   lam_i = S_ii / (1 - R_ii),
        E_ij = (S_ij + lam_j R_ij) / (lam_j - lam_i)   where that stays small,   R_ij / 2 otherwise,
        X <- X + X E

    -- four n^3 GEMMs and a few elementwise passes per step (rocBLAS through torch), quadratically
    convergent once the Fock matrix moves little between cycles, which is most of an SCF run.  The
    dense eigenproblem is otherwise the largest item of a cycle (n = 494: 11.4 ms on hipSOLVER, 16.5 ms
    on 16 host cores, against 6.2 ms for the XC sweep); a refinement step costs ~0.15 ms there.

    Pairs whose first-order rotation would exceed 1/3 (near-degenerate orbitals, e.g. Benzene's e pairs)
    get their orthonormality part and an exact 2x2 rotation instead; their mutual rotation is irrelevant to
    the density matrix as long as both lie on one side of the gap, and that is checked -- an
    occupied-virtual pair left in that state,
    a first step above 0.25, a step that does not contract by 0.3, or `max_it` steps without reaching `tol`
    all hand the matrix to the full solver (`exact`), which also does the first cycle.  Returns all n
    orbitals sorted by energy, like eigh.

    OPT-IN (`--eigensolver refine`).  It reproduces the exact loop to 1e-12 Ha with the same cycle count
    (tests/test_scf_cpu.py), but as it stands it does not pay on MI355X: only the later half of the cycles
    moves little enough to converge (Benzene/def2-SVP 8 of 16 cycles; Anthracene/def2-SVP, whose virtual
    spectrum is dense, 4 of 24), each step is ~15 small launches and two host syncs
    (~0.1 ms at n = 114), and the failed attempts are paid on top of the full solve: 2.04 against 1.42 ms
    per cycle (Benzene), 7.6 against 7.0 ms (Anthracene/def2-SVP).  See DESIGN.md section 8."""

    def __init__(self, S, nocc, exact, device=None, tol=1e-10, max_it=6):
        import torch
        self.torch, self.n, self.nocc, self.exact = torch, S.shape[0], nocc, exact
        self.tol, self.max_it = tol, max_it
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        s, U = np.linalg.eigh(S)
        self.Xo = torch.as_tensor(U / np.sqrt(s), dtype=torch.float64, device=self.dev)
        self.Xo_inv = torch.as_tensor((U * np.sqrt(s)).T, dtype=torch.float64, device=self.dev)  # Xo^-1 = s^1/2 U^T
        self.eye = torch.eye(self.n, dtype=torch.float64, device=self.dev)
        self.X = None
        self.stats = {"exact": 0, "refined": 0, "steps": 0}

    def _refine(self, Fp, X):
        t = self.torch
        prev = None
        for _ in range(self.max_it):
            R = self.eye - X.T @ X
            S = X.T @ (Fp @ X)
            lam = t.diagonal(S) / (1.0 - t.diagonal(R))
            diff = lam[None, :] - lam[:, None]                  # lam_j - lam_i
            num = S + lam[None, :] * R
            far = diff.abs() > 3.0 * num.abs() + 1e-10          # first-order rotation stays below 1/3
            far.fill_diagonal_(False)
            # the other pairs (near-degenerate orbitals): orthonormality part, plus -- where they are really
            # coupled -- the exact 2x2 rotation that zeroes S_ij (tan 2 theta = 2 S_ij / (S_jj - S_ii), small
            # branch); left coupled they make the convergence of every pair they touch linear
            coupled = (~far) & (S.abs() > 1e-12 * float(lam.abs().max()))
            coupled.fill_diagonal_(False)
            theta = 0.5 * t.atan2(2.0 * S, diff)
            theta = t.where(theta > np.pi / 4, theta - np.pi / 2, theta)
            theta = t.where(theta < -np.pi / 4, theta + np.pi / 2, theta)
            EJ = t.where(coupled, t.sin(theta), t.zeros_like(S))
            E = t.where(far, num / t.where(far, diff, t.ones_like(diff)), 0.5 * R + EJ)
            # occupied-virtual pairs must all be resolved: a coupled pair left "near" across the gap is a failure
            occ = t.zeros(self.n, dtype=t.bool, device=self.dev)
            occ[t.argsort(lam)[:self.nocc]] = True
            across = occ[:, None] != occ[None, :]
            stuck = bool(((~far) & across & (num.abs() > 1e-9)).any())
            emax = float(t.where(coupled, t.zeros_like(E), E).abs().max())   # rotations inside a cluster are free
            self.stats["steps"] += 1
            if stuck or emax > 0.25 or (prev is not None and emax > 0.3 * prev and emax > 1e-9):
                return None, None
            X = X + X @ E
            if emax < self.tol:
                return X, lam
            prev = emax
        return None, None

    def __call__(self, F):
        t = self.torch
        Fd = t.as_tensor(F, dtype=t.float64, device=self.dev)
        Fp = self.Xo.T @ Fd @ self.Xo
        X = lam = None
        if self.X is not None:
            X, lam = self._refine(Fp, self.X)
        if X is None:
            e, C = self.exact(F)                                   # host LAPACK or hipSOLVER (FockDiagonaliser)
            self.X = self.Xo_inv @ t.as_tensor(C, dtype=t.float64, device=self.dev)
            self.stats["exact"] += 1
            return e, C
        self.stats["refined"] += 1
        order = t.argsort(lam)
        self.X = X[:, order]
        return lam[order].cpu().numpy(), (self.Xo @ self.X).cpu().numpy()


class SubspaceDiagonaliser:
    """Occupied orbitals by Chebyshev-filtered subspace iteration, warm-started from the previous cycle.

    `eigh(F, S)` (dft.py:181,227) is the largest single item of an SCF cycle once the XC sweep and
    J/K run on the GPU (n = 494: 11.4 ms on hipSOLVER, 16.5 ms on 16 host cores, against 6.5 ms for
    the XC sweep), yet the loop only consumes the nocc lowest orbitals and the Fock matrix changes
    little between cycles.  Here the orthogonalised F' = X^T F X (X = U s^-1/2, once per S) acts on a
    block V of m = nocc + buffer vectors: a degree-`degree` Chebyshev polynomial that damps
    [theta_m, ||F'||_1] and grows below it (GEMMs, rocBLAS through torch.addmm), orthonormalisation
    and Rayleigh-Ritz through m x m matrices on the host.  Passes repeat until the occupied residual
    max_i ||F' v_i - theta_i v_i|| < tol; the first cycle, any cycle whose Fock matrix moved too far
    for the old block (residual of the unfiltered block > `exact_above`), and any that does not reach
    tol in `max_pass` passes fall back to the full diagonalisation.  Returns the m lowest orbitals
    (energies, coefficients); their span agrees with the exact one to ~tol.

    OPT-IN (`--eigensolver subspace`), not the default: it reproduces the exact loop's energies to
    1e-9 Ha (tests/test_scf_cpu.py) but did not pay on MI355X -- Anthracene/def2-TZVP (n = 494): the
    spectrum is ~10^2 Ha wide against a 0.1 Ha gap, degree 16 gains only ~3x per pass, 17 of 26
    cycles fell back to the full solve and the cycle took 35 ms instead of 30."""

    def __init__(self, S, nocc, device=None, degree=16, tol=1e-9, max_pass=4, nbuf=None, exact_above=0.5):
        import torch
        self.torch, self.n, self.nocc = torch, S.shape[0], nocc
        self.m = min(self.n, nocc + (nbuf if nbuf is not None else max(10, nocc // 4)))
        self.degree, self.tol, self.max_pass, self.exact_above = degree, tol, max_pass, exact_above
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        s, U = np.linalg.eigh(S)
        self.X = torch.as_tensor(U / np.sqrt(s), dtype=torch.float64, device=self.dev)
        self.V = self.theta = None
        self.stats = {"exact": 0, "subspace": 0, "passes": 0}

    def _exact(self, Fp):
        t = self.torch
        if self.dev.type == "cpu" or self.n < 400:   # small problems: LAPACK on the host is faster than hipSOLVER
            th, V = np.linalg.eigh(Fp.cpu().numpy())
            th, V = t.as_tensor(th[:self.m], device=self.dev), t.as_tensor(np.ascontiguousarray(V[:, :self.m]), device=self.dev)
        else:
            th, V = t.linalg.eigh(Fp)
            th, V = th[:self.m].clone(), V[:, :self.m].clone()
        self.stats["exact"] += 1
        return th, V

    def _small_eigh(self, M):
        w, Q = np.linalg.eigh(M.cpu().numpy())
        return w, Q

    def _rayleigh_ritz(self, Fp, Y):
        """Orthonormalise Y through the eigen-decomposition of its Gram matrix (twice), then Ritz pairs."""
        t = self.torch
        for _ in range(2):
            w, Q = self._small_eigh(Y.T @ Y)
            w = np.maximum(w, w.max() * 1e-28)
            Y = Y @ t.as_tensor(Q / np.sqrt(w), device=self.dev)
        FY = Fp @ Y
        th, Q = self._small_eigh(Y.T @ FY)
        Qd = t.as_tensor(Q, device=self.dev)
        V, FV = Y @ Qd, FY @ Qd
        thd = t.as_tensor(th, device=self.dev)
        resid = float(t.linalg.norm(FV[:, :self.nocc] - V[:, :self.nocc] * thd[:self.nocc], dim=0).max())
        return thd, V, resid

    def _filter(self, Fp, V, lam_lo, lam_cut, lam_up):
        t = self.torch
        e, c = 0.5 * (lam_up - lam_cut), 0.5 * (lam_up + lam_cut)
        Fs = Fp - c * t.eye(self.n, dtype=Fp.dtype, device=self.dev)
        sigma = e / (lam_lo - c)
        sigma1 = sigma
        Y = (Fs @ V) * (sigma1 / e)
        for _ in range(2, self.degree + 1):
            sigma2 = 1.0 / (2.0 / sigma1 - sigma)
            V, Y = Y, t.addmm(V, Fs, Y, beta=-sigma * sigma2, alpha=2.0 * sigma2 / e)
            sigma = sigma2
        return Y

    def __call__(self, F):
        t = self.torch
        Fd = t.as_tensor(F, dtype=t.float64, device=self.dev)
        Fp = self.X.T @ Fd @ self.X
        done = False
        if self.V is not None:
            th, V, resid = self._rayleigh_ritz(Fp, self.V)        # how far did the Fock matrix move?
            if resid < self.exact_above:
                lam_up = float(t.linalg.matrix_norm(Fp, ord=1))   # a guaranteed bound of the spectrum
                for _ in range(self.max_pass):
                    if resid < self.tol:
                        break
                    lam_lo, lam_cut = float(th[0]), float(th[self.m - 1])
                    if not (lam_lo < lam_cut < lam_up):
                        break
                    th, V, resid = self._rayleigh_ritz(Fp, self._filter(Fp, V, lam_lo, lam_cut, lam_up))
                    self.stats["passes"] += 1
                done = resid < self.tol
        if done:
            self.stats["subspace"] += 1
        else:
            th, V = self._exact(Fp)
        self.V, self.theta = V, th
        return th.cpu().numpy(), (self.X @ V).cpu().numpy()


class HipBackend:
    """Device side of the loop: libdft.so through DFTSolverWrapper, torch tensors as buffers.

    One process per GPU.  With world > 1 (torch.distributed initialised by the caller) this rank
    keeps grid block shard_bounds(ngrid, world, rank) -- AO values are only ever evaluated for it --
    and Cholesky-vector slice vector_bounds(naux, world, rank) resident; a cycle is the local XC
    sweep + local J/K followed by ONE all-reduce of [Vxc | J | K | Exc] (grid_shard.ShardedFock),
    after which every rank holds identical matrices and repeats the small host part.  A dense ERI
    is not sharded (rank 0 contracts it): large jobs use the factorised form."""

    def __init__(self, inp, functional, lib_path=None, quirks=True, rank=0, world=1, device=None, group=None,
                 eigensolver="auto"):
        import torch
        from .grid_shard import ShardedFock, shard_bounds, vector_bounds
        from .solver import DFTSolverWrapper
        assert torch.cuda.is_available(), "the SCF driver needs a GPU (there is no CPU fallback)"
        self.torch, self.dev = torch, torch.device(device if device is not None else "cuda")
        if self.dev.index is not None:
            torch.cuda.set_device(self.dev)
        self.functional = functional.upper()
        self.solver = DFTSolverWrapper(lib_path, self.functional)
        self.solver.set_option("quirks", 1 if quirks else 0)
        t0 = time.time()
        self.rank, self.world = rank, world
        nao = inp.shells.nao
        lo, hi = shard_bounds(inp.grids.size, world, rank)
        ngrid = hi - lo
        self.nao, self.ngrid = nao, ngrid
        f64 = torch.float64
        n1 = max(ngrid, 1)   # an empty block keeps one dummy row so every pointer stays valid; it is never swept
        d_coords = torch.zeros((n1, 3), dtype=f64, device=self.dev)
        self.d_w = torch.zeros(n1, dtype=f64, device=self.dev)
        d_coords[:ngrid] = torch.as_tensor(inp.grids.coords[lo:hi], dtype=f64)
        self.d_w[:ngrid] = torch.as_tensor(inp.grids.weights[lo:hi], dtype=f64)
        self.d_ao = torch.zeros((n1, nao), dtype=f64, device=self.dev)
        self.d_gr = torch.zeros((3, n1, nao), dtype=f64, device=self.dev) if self.functional != "LDA" else None
        if ngrid:
            self.solver.eval_ao(inp.shells, d_coords, ngrid, self.d_ao, self.d_gr)  # grid.py:30,38 on the device
        self.d_eri = self.d_chol = self.d_cocc = None
        if inp.eri is not None:
            if rank == 0:
                self.d_eri = torch.as_tensor(inp.eri.reshape(nao * nao, nao * nao), dtype=f64, device=self.dev)  # dft.py:166
        else:  # factorised J/K (DFT_ComputeJKFactorized): Cholesky vectors stay resident instead of the ERI
            plo, phi = vector_bounds(inp.chol.shape[0], world, rank)
            self.d_chol = torch.as_tensor(inp.chol[plo:phi], dtype=f64, device=self.dev)
            self.d_cocc = torch.zeros((nao, inp.nocc), dtype=f64, device=self.dev)
        self.nocc = inp.nocc
        self._pins = {}
        self.diis_device = self.dev if nao >= 200 else None   # DIIS products on the GPU once they cost more than the hops
        if world > 1:
            self._sharded = ShardedFock(nao, self._local_sweep, self._local_jk, self.dev, group)
            self.fock_parts = self._fock_parts
        self.d_dm = torch.zeros((nao, nao), dtype=f64, device=self.dev)
        self.d_J = torch.zeros_like(self.d_dm); self.d_K = torch.zeros_like(self.d_dm); self.d_v = torch.zeros_like(self.d_dm)
        # "exact" / "auto": eigh every cycle (the reference's loop).  Opt-in experiments, both with the full
        # solver as fallback: "refine" (the previous cycle's eigenvectors refined on the GPU), "subspace"
        # (filtered subspace iteration)
        if eigensolver == "subspace":
            self.eigh = SubspaceDiagonaliser(inp.S, inp.nocc, self.dev)
        elif eigensolver == "refine":
            self.eigh = RefinedDiagonaliser(inp.S, inp.nocc, FockDiagonaliser(inp.S, self.dev), self.dev)
        else:
            self.eigh = FockDiagonaliser(inp.S, self.dev)
        torch.cuda.synchronize()
        self.init_time = time.time() - t0

    def _upload(self, dst, host, key):
        """numpy -> device through a pinned staging buffer (a pageable copy of the 10.6 MB density matrix
        at nao 1150 took 11 ms, ~1 GB/s)."""
        pin = self._pins.get(key)
        if pin is None:
            pin = self._pins[key] = self.torch.empty(dst.shape, dtype=dst.dtype).pin_memory()
        pin.copy_(self.torch.as_tensor(np.ascontiguousarray(host), dtype=dst.dtype))
        dst.copy_(pin, non_blocking=True)

    def set_dm(self, dm):
        self._upload(self.d_dm, dm, "dm")                                         # dft.py:200

    def set_cocc(self, cocc):
        """cocc (nao, nocc) with dm = cocc cocc^T; only the factorised exchange needs it."""
        if self.d_cocc is not None:
            self._upload(self.d_cocc, cocc, "cocc")

    def _jk_device(self, want_k):
        """This rank's J (and K) into d_J / d_K; zeros when it holds no vectors / not the dense ERI."""
        if self.d_chol is not None and self.d_chol.shape[0]:
            self.solver.compute_jk_factorized(self.nao, self.d_chol.shape[0], self.nocc, self.d_chol, self.d_dm,
                                              self.d_cocc if want_k else None, self.d_J, self.d_K if want_k else None)
        elif self.d_eri is not None and want_k:
            self.solver.compute_jk(self.nao, self.d_eri, self.d_dm, self.d_J, self.d_K)
        elif self.d_eri is not None:
            self.solver.compute_coulomb(self.nao, self.d_eri, self.d_dm, self.d_J)    # dft.py:203
        else:
            self.d_J.zero_(); self.d_K.zero_()

    def jk(self, want_k):
        self._jk_device(want_k)
        return self.d_J.cpu().numpy(), (self.d_K.cpu().numpy() if want_k else None)

    def xc(self):
        t0 = time.time()
        exc = self._xc_device()
        return exc, self.d_v.cpu().numpy(), time.time() - t0

    def _xc_device(self):
        if not self.ngrid:
            self.d_v.zero_()
            return 0.0
        exc = self.solver.compute_xc(self.ngrid, self.nao, self.d_dm, self.d_ao, self.d_w, self.d_v, self.d_gr)
        self.torch.cuda.synchronize()                                                # dft.py:205-208
        return exc

    # world > 1: the two local steps as ShardedFock wants them, and the all-reduced cycle
    def _local_sweep(self, dm):
        return self._xc_device(), self.d_v

    def _local_jk(self, dm, cocc):
        self._jk_device(self._want_k)
        return self.d_J, (self.d_K if self._want_k else None)

    def _fock_parts(self, want_k):
        """(J, K, Exc, Vxc_raw, seconds in the local sweep) identical on every rank."""
        self._want_k = want_k
        t0 = time.time()
        parts = self._sharded.compute(self.d_dm, self.d_cocc)
        return (parts.J.cpu().numpy(), parts.K.cpu().numpy() if want_k else None, parts.exc,
                parts.vxc.cpu().numpy(), time.time() - t0)


def run_scf(inp, backend, functional, max_cycle=200, conv_e=1e-8, conv_dm=1e-6, log=print):
    from .hostinfo import blas_threads
    # host LAPACK/BLAS never on every visible core (256 on a 16-core share: ~90 ms stalls); below 400
    # functions one thread is fastest for everything left on the host (dsyevd at n = 114: 0.67 ms on one
    # thread, 0.87 on 16), above it the pool gets the CPU share
    with blas_threads(1 if inp.S.shape[0] < 400 else None):
        return _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log)


def _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log):
    functional = functional.upper()
    c_hf = 0.2 if functional == "B3LYP" else 0.0                                       # dft.py:197
    Hcore, S, nocc = inp.Hcore, inp.S, inp.nocc
    solve = getattr(backend, "eigh", None) or (lambda F: eigh(F, S))
    e, C = solve(Hcore)                                                                # dft.py:181
    dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T
    set_cocc = getattr(backend, "set_cocc", None)
    diis = CDIIS(device=getattr(backend, "diis_device", None))
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
        if hasattr(backend, "fock_parts"):     # multi-GPU: local XC + local J/K, one all-reduce
            J, K, E_xc, Vraw, t_xc = backend.fock_parts(functional == "B3LYP")
            jk_times.append(0.0)
        else:
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
