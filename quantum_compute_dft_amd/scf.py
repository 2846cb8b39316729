"""The SCF loop of the reference driver (dft.py:181-266) with the device work behind a backend.

Loop contract kept exactly: core guess `eigh(Hcore, S)`; per cycle J (compute_coulomb), XC
(compute_xc), `Vxc = (V + V^T)/2` (dft.py:212), for B3LYP K and `F = H + J + Vxc - 0.5*0.2*K`
(dft.py:217-221), DIIS on (S, dm, F), `eigh(F, S)`, energies from the NEW density with J/K/Exc of
the old one (dft.py:230-236), convergence |dE| < 1e-8 and ||d dm||_F < 1e-6 (dft.py:243),
200 cycles.  Differences: J and K come from ONE pass over the ERI (DFT_ComputeJK), AO values
are evaluated on the device.  Two forms of the loop body:

* host part on the host (`_run_scf`, below `device_from` = 200 basis functions, where one LAPACK
  thread beats everything the device offers: n = 114 0.5 ms of dsyevd against 1.9 ms for hipSOLVER and 1.4-1.6 ms
  per cycle for the device loop with the rotation solver against 1.2): per cycle ONE pinned upload [dm | cocc]
  and ONE pinned download [J | K | Vxc] cross PCIe;
* device-resident (`_run_scf_device`, from 200 functions, or `HipBackend(device_resident=True)`): dm,
  cocc, J, K, Vxc, F, the DIIS history, the eigenproblem (occupied-subspace rotation; its few full solves
  through hipSOLVER from 400 functions, through one host LAPACK thread below) and dm = 2 C_occ C_occ^T all
  stay in HBM; per cycle one 4-double download (E_one, E_coul, E_ex, |d dm|) besides the Exc the ABI returns.
  Anthracene B3LYP/def2-SVP (n = 246): 4.3 against 6.0 ms per cycle for the host form.

With world > 1 rank 0 is authoritative: it alone runs DIIS + eigh and broadcasts [dm | cocc | scalars]
(grid_shard.ReplicaSync), so replicas cannot drift apart and every rank leaves the loop in the same cycle
(the stop decision is made from the broadcast scalars).
"""
import time

import numpy as np
from scipy.linalg import eigh


class CDIIS:
    """Pulay DIIS on the commutator SDF - FDS (PySCF scf.diis.CDIIS as used at dft.py:184,225).

    The history is a ring of `space` (F, e) pairs in two flat (space, n^2) arrays; per cycle the Gram matrix of the
    error vectors gets ONE new row (a GEMV) and the extrapolation is one GEMV over the stored Fock matrices (the
    textbook form recomputes 36 dot products and sums 8 scaled matrices: 0.12 against 0.07 ms of a Benzene cycle).
    The commutator is formed through the thin factor when the caller passes cocc with dm = cocc cocc^T: three
    n^2 n_occ products instead of two n^3 ones.  With `device` arrays and products live on the GPU (rocBLAS through
    torch; n = 494 costs 2.3 ms per cycle on 16 host cores) and only the (space+1)^2 system is solved on the host."""

    def __init__(self, space=8, device=None):
        self.space, self.dev = space, device
        self._slots = None
        if device is not None:
            import torch
            self.torch = torch

    def update(self, S, dm, F, keep_on_device=False, cocc=None):
        """Extrapolated Fock matrix.  Device mode accepts numpy or device tensors; `keep_on_device` returns
        the device tensor (device-resident loop) instead of a numpy copy."""
        dev = self.dev is not None
        if dev:
            t = self.torch
            f64 = t.float64
            conv = lambda a: a if t.is_tensor(a) else t.as_tensor(a, dtype=f64, device=self.dev)
        n2 = F.shape[0] * F.shape[1]
        if self._slots is None:
            if dev:
                self._Fb, self._Eb = (t.zeros((self.space, n2), dtype=f64, device=self.dev) for _ in range(2))
                self._S = conv(S)
            else:
                self._Fb, self._Eb = np.zeros((self.space, n2)), np.zeros((self.space, n2))
                self._S = S
            self._Gb, self._slots = np.zeros((self.space, self.space)), []
        slot = self._slots.pop(0) if len(self._slots) == self.space else len(self._slots)   # the oldest pair is overwritten
        Fk = conv(F) if dev else F
        if cocc is not None:
            c = conv(cocc) if dev else cocc
            sdf = (self._S @ c) @ (c.T @ Fk)
        else:
            sdf = self._S @ (conv(dm) if dev else dm) @ Fk
        self._Fb[slot] = Fk.reshape(-1)
        if dev:
            t.sub(sdf.T, sdf, out=self._Eb[slot].view(F.shape))
        else:
            np.subtract(sdf.T, sdf, out=self._Eb[slot].reshape(F.shape))
        self._slots.append(slot)
        idx = np.array(self._slots)
        row = self._Eb @ self._Eb[slot]          # against every slot; unused ones are zero and never read
        row = row.cpu().numpy() if dev else row
        self._Gb[slot, idx] = self._Gb[idx, slot] = row[idx]
        n = len(idx)
        if n < 2:
            return (Fk if keep_on_device else F)
        B = np.zeros((n + 1, n + 1)); B[0, 1:] = B[1:, 0] = 1.0
        B[1:, 1:] = self._Gb[np.ix_(idx, idx)]
        rhs = np.zeros(n + 1); rhs[0] = 1.0
        try:
            cf = np.linalg.solve(B, rhs)[1:]
        except np.linalg.LinAlgError:
            cf = np.linalg.lstsq(B, rhs, rcond=None)[0][1:]
        m = int(idx.max()) + 1
        cs = np.zeros(m); cs[idx] = cf
        if not dev:
            return (cs @ self._Fb[:m]).reshape(F.shape)
        out = (t.as_tensor(cs, device=self.dev) @ self._Fb[:m]).view(F.shape)
        return out if keep_on_device else out.cpu().numpy()


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


class OccupiedRotation:
    """The occupied orbitals of F C = S C e WITHOUT a dense eigensolve per cycle.

    The SCF loop only consumes the occupied subspace (dm = 2 C_occ C_occ^T, dft.py:182,228), and between two
    cycles the Fock matrix moves little.  A full S-orthonormal basis U = [U_o | U_v] is kept from the last full
    diagonalisation; each cycle forms A = U^T F U (two n^3 GEMMs, the only n^3 work) and finds the rotation that
    decouples the two blocks: with K (n_virt x n_occ) solving the Riccati equation

        A_vo + A_vv K - K A_oo - K A_ov K = 0

    the columns of U_o + U_v K span the new occupied space exactly.  K is found by the diagonally preconditioned
    fixed point K <- K - R / (a_v - a_o) -- denominators only ACROSS the gap, so near-degenerate orbitals inside
    either block (Benzene's e pairs, the dense virtual spectrum of a TZVP basis) never enter, which is what
    defeated the refinement and filtering attempts of round 1 -- at O(n_virt^2 n_occ) per step.  The orthogonal
    completion  U_o' = (U_o + U_v K)(1 + K^T K)^-1/2,  U_v' = (U_v - U_o K^T)(1 + K K^T)^-1/2  needs only the
    n_occ x n_occ eigen-decomposition of K^T K; the occupied block is then made canonical (an n_occ x n_occ eigh), so
    its side of the next cycle's denominators is exact.  Falls back to the full solver (which also does the first
    cycle) when the first-order rotation exceeds 0.5, the fixed point has not reached `tol` in `max_inner` steps or
    grows, or the aufbau order is in doubt (highest occupied level within 1e-3 Ha of the lowest virtual diagonal).
    The virtual block is never diagonalised, so that test is a necessary one only: run_scf therefore checks the
    CONVERGED state once against eigh(F, S) (which also supplies the full spectrum it reports) and resumes with the
    full solver and a fresh DIIS history if the occupied space it followed is not the aufbau one.

    Where it pays (profiles/r02_eigensolver.txt): solved to 1e-10 the fixed point needs 15-20 steps in the middle of
    an SCF run (the Fock matrix still moves by 1e-2) and 4-6 at its end; stopped at 1e-3 x the last density change
    (`accuracy`, inexact diagonalisation -- the orbitals stay exactly orthonormal) it needs 3-5 per cycle.  Per SCF
    cycle of the real molecules: n = 494 (device) 12.5 against 23.4 ms, n = 246 (device) 4.3 against 6.9 ms, n = 114
    (host, plain numpy: ~25 small BLAS/LAPACK calls) 0.86 against 1.23 ms; below ~80 functions dsyevd itself is
    cheaper than those calls -- `eigensolver="auto"` uses it from 80 functions.  Same converged energies (to the
    SCF's own thresholds) and cycle counts within one of the exact loop (tests/test_scf_cpu.py)."""

    def __init__(self, S, nocc, device=None, tol=1e-10, max_inner=60):
        import torch
        self.t, self.no, self.tol, self.max_inner = torch, int(nocc), tol, max_inner
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        s, V = np.linalg.eigh(S)
        self.host = self.dev.type == "cpu"      # host form: plain numpy (a torch-CPU operation costs 3-5 us, ~70 of them per cycle)
        self.X = V / np.sqrt(s) if self.host else torch.as_tensor(V / np.sqrt(s), dtype=torch.float64, device=self.dev)   # S^-1/2 (columns)
        self.U = None
        self.stats = {"exact": 0, "rotated": 0, "inner_steps": 0}

    def reset(self):
        """Forget the followed subspace and the counters: the next call is a full solve (a second SCF on the same backend)."""
        self.U = None
        self.stats = {"exact": 0, "rotated": 0, "inner_steps": 0}

    def _exact(self, F):
        t = self.t
        Fp = self.X.T @ F @ self.X
        if self.host:
            e, Cp = eigh(Fp, driver="evd")
            self.U = self.X @ Cp
            self.stats["exact"] += 1
            return e, self.U[:, :self.no]
        if F.shape[0] < 400:
            # one LAPACK thread beats hipSOLVER below ~400 functions (FockDiagonaliser: n = 246 2.5 against ~6 ms), also
            # from the device-resident loop: the few full solves of a run cross PCIe (2 n^2 doubles), the rotations do not
            e, Cp = eigh(Fp.cpu().numpy(), driver="evd")
            e, Cp = t.from_numpy(e).to(self.dev), t.from_numpy(Cp).to(self.dev)
        else:
            e, Cp = t.linalg.eigh(Fp)
        self.U = self.X @ Cp
        self.stats["exact"] += 1
        return e, self.U[:, :self.no]

    def occupied(self, F, accuracy=None):
        """(orbital energies, C_occ (n, nocc)) for the Fock matrix F (numpy array or tensor on self.dev); the
        energies are exact for the occupied block, diagonal estimates for the virtual one after a rotation.
        `accuracy`: residual at which the fixed point may stop in THIS call (never below self.tol); the SCF loop
        passes 1e-3 x the last density change, so a cycle that still moves the density by 1e-3 is not solved to
        1e-10 -- the returned orbitals are exactly orthonormal either way, i.e. always a valid trial density."""
        t, no = self.t, self.no
        tol = self.tol if accuracy is None else min(max(self.tol, float(accuracy)), 1e-5)
        if self.host:
            return self._occupied_host(np.asarray(F), tol)
        F = F if t.is_tensor(F) else t.as_tensor(F, dtype=t.float64, device=self.dev)
        if self.U is None or no == 0 or no == F.shape[0]:
            return self._exact(F)
        U = self.U
        A = U.T @ (F @ U)
        d = t.diagonal(A)
        do, dv = d[:no], d[no:]
        Aoo, Aov, Avo, Avv = A[:no, :no], A[:no, no:], A[no:, :no], A[no:, no:]
        rden = 1.0 / (dv[:, None] - do[None, :])
        K = -(Avo * rden)
        if not float(K.abs().max()) <= 0.5:
            return self._exact(F)
        # every operation below is a kernel launch (~8 us each at these sizes, the whole cost of the device form):
        # fused multiply-adds (addmm / addcmul), a convergence test -- a host sync -- every third step only, and
        # ONE download / ONE upload for the two n_occ x n_occ eigen-decompositions of the completion
        prev, ok = float("inf"), False
        for it in range(self.max_inner):
            R = t.addmm(Avo, Avv, K).addmm_(K, t.addmm(Aoo, Aov, K), alpha=-1.0)   # Avo + Avv K - K (Aoo + Aov K)
            self.stats["inner_steps"] += 1
            if it % 3 == 0:
                r = float(R.abs().max())
                if r < tol:
                    ok = True
                    break
                if not (r < 4.0 * prev):    # diverging (or NaN)
                    break
                prev = min(prev, r)
            K = t.addcmul(K, R, rden, value=-1.0)
        if not ok:
            return self._exact(F)
        Kt = K.T
        Fo = t.addmm(t.addmm(Aoo, Aov, K), Kt, t.addmm(Avo, Avv, K))          # Y^T F Y in the U basis, Y = Uo + Uv K
        pack = t.cat([(Kt @ K).reshape(-1), Fo.reshape(-1), dv.min().reshape(1)]).cpu().numpy()
        M, Foh, dvmin = pack[:no * no].reshape(no, no), pack[no * no:2 * no * no].reshape(no, no), pack[-1]
        lam, V = np.linalg.eigh(M)
        lam = np.maximum(lam, 0.0)
        isq = 1.0 / np.sqrt(1.0 + lam)
        Mo = (V * isq) @ V.T                                                  # (1 + K^T K)^-1/2
        G = (V * np.where(lam > 1e-12, (isq - 1.0) / np.maximum(lam, 1e-300), -0.5)) @ V.T   # (1 + K K^T)^-1/2 = 1 + K G K^T
        Aoo2 = Mo @ Foh @ Mo                                                  # occupied block in the rotated basis
        eo, Vo = np.linalg.eigh(0.5 * (Aoo2 + Aoo2.T))
        if eo[-1] > dvmin - 1e-3:                                             # aufbau order in doubt
            return self._exact(F)
        up = t.as_tensor(np.concatenate([G.ravel(), (Mo @ Vo).ravel(), eo]), device=self.dev)
        G_d, c_d, eo_d = up[:no * no].view(no, no), up[no * no:2 * no * no].view(no, no), up[2 * no * no:]
        Uo, Uv = U[:, :no], U[:, no:]
        Un = t.empty_like(U)
        t.mm(t.addmm(Uo, Uv, K), c_d, out=Un[:, :no])                         # canonical occupied orbitals (Uo + Uv K) Mo Vo
        T = t.addmm(Uv, Uo, Kt, alpha=-1.0)
        Un[:, no:] = t.addmm(T, (T @ K) @ G_d, Kt)
        self.U = Un
        self.stats["rotated"] += 1
        return t.cat([eo_d, dv]), Un[:, :no]

    def _occupied_host(self, F, tol):
        """The same algorithm in numpy (returns numpy arrays): at n = 114 one cycle is ~25 small BLAS/LAPACK calls."""
        no = self.no
        if self.U is None or no == 0 or no == F.shape[0]:
            return self._exact(F)
        U = self.U
        A = U.T @ (F @ U)
        d = np.diagonal(A)
        do, dv = d[:no], d[no:]
        Aoo, Aov, Avo, Avv = A[:no, :no], A[:no, no:], A[no:, :no], A[no:, no:]
        rden = 1.0 / (dv[:, None] - do[None, :])
        K = -Avo * rden
        if not (np.abs(K).max() <= 0.5):
            return self._exact(F)
        prev, ok = float("inf"), False
        for it in range(self.max_inner):
            R = Avo + Avv @ K - K @ (Aoo + Aov @ K)
            self.stats["inner_steps"] += 1
            r = np.abs(R).max()
            if r < tol:
                ok = True
                break
            if not (r < 4.0 * prev):        # diverging (or NaN)
                break
            prev = min(prev, r)
            K = K - R * rden
        if not ok:
            return self._exact(F)
        lam, V = np.linalg.eigh(K.T @ K)
        lam = np.maximum(lam, 0.0)
        isq = 1.0 / np.sqrt(1.0 + lam)
        Mo = (V * isq) @ V.T                                                  # (1 + K^T K)^-1/2
        g = np.where(lam > 1e-12, (isq - 1.0) / np.maximum(lam, 1e-300), -0.5)
        G = (V * g) @ V.T                                                     # (1 + K K^T)^-1/2 = 1 + K G K^T
        Uo, Uv = U[:, :no], U[:, no:]
        T = Uv - Uo @ K.T
        Uv2 = T + ((T @ K) @ G) @ K.T
        Uo2 = (Uo + Uv @ K) @ Mo
        AvvK = Avv @ K
        Aoo2 = Mo @ (Aoo + Aov @ K + K.T @ Avo + K.T @ AvvK) @ Mo             # occupied block in the rotated basis
        eo, Vo = np.linalg.eigh(0.5 * (Aoo2 + Aoo2.T))
        if eo[-1] > dv.min() - 1e-3:                                          # aufbau order in doubt
            return self._exact(F)
        Un = np.empty_like(U)
        Un[:, :no] = Uo2 @ Vo                                                 # canonical occupied orbitals
        Un[:, no:] = Uv2
        self.U = Un
        self.stats["rotated"] += 1
        return np.concatenate([eo, dv]), Un[:, :no]


class HipBackend:
    """Device side of the loop: libdft.so through DFTSolverWrapper, torch tensors as buffers.

    One process per GPU.  With world > 1 (torch.distributed initialised by the caller) this rank
    keeps grid block shard_bounds(ngrid, world, rank) -- AO values are only ever evaluated for it --
    and Cholesky-vector slice vector_bounds(naux, world, rank) resident; a cycle is the local XC
    sweep + local J/K followed by ONE all-reduce of [Vxc | J | K | Exc] (grid_shard.ShardedFock).  A
    dense ERI is sharded by ROWS (ij): rank r contracts rows eri_row_bounds(nao^2, world, r) into its
    rows of J (and, through the (i,k) view, its partial K); the all-reduce assembles them."""

    def __init__(self, inp, functional, lib_path=None, quirks=True, rank=0, world=1, device=None, group=None,
                 device_resident=None, device_from=200, eigensolver="auto", ao_mode="resident", ao_chunk=0, xc_occ=True,
                 fused_tail=None):
        import torch
        from .build import library_path
        from .grid_shard import ReplicaSync, ShardedFock, eri_row_bounds, shard_bounds, vector_bounds
        from .solver import DFTSolverWrapper
        assert torch.cuda.is_available(), "the SCF driver needs a GPU (there is no CPU fallback)"
        self.torch, self.dev = torch, torch.device(device if device is not None else "cuda")
        if self.dev.index is not None:
            torch.cuda.set_device(self.dev)
        self.functional = functional.upper()
        self.solver = DFTSolverWrapper(lib_path or library_path(), self.functional)
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
        # "resident" (the reference's layout, dft.py:155,172): AO values / gradients of the whole grid block stay in
        # HBM for the run (8 ngrid nao (1 or 4) bytes; 53 GB for BASELINE config 5).  "direct": only the grid and the
        # shell table stay; every cycle re-evaluates them chunk by chunk inside DFT_ComputeXCDirect.
        if ao_mode not in ("resident", "direct"):
            raise ValueError(f"ao_mode {ao_mode!r}: expected 'resident' or 'direct'")
        self.ao_mode, self.ao_chunk, self.shells, self.d_coords = ao_mode, int(ao_chunk), inp.shells, d_coords
        # the sweep's density step through the occupied orbitals (DFT_ComputeXCOcc: the loop holds cocc with
        # dm = cocc cocc^T in every cycle, dft.py:181-182); False = the reference's call with the full matrix
        self.xc_occ = bool(xc_occ)
        self.d_ao = self.d_gr = None
        if ao_mode == "resident":
            self.d_ao = torch.zeros((n1, nao), dtype=f64, device=self.dev)
            self.d_gr = torch.zeros((3, n1, nao), dtype=f64, device=self.dev) if self.functional != "LDA" else None
            if ngrid:
                self.solver.eval_ao(inp.shells, d_coords, ngrid, self.d_ao, self.d_gr)  # grid.py:30,38 on the device
        else:
            self._d_exc = torch.zeros(1, dtype=f64, device=self.dev)
        self.d_eri = self.d_chol = self.d_cocc = None
        self.eri_rows = (0, nao * nao)
        if inp.eri is not None:
            # dft.py:166 uploads the whole (nao^2, nao^2) matrix; with world > 1 each rank keeps its ROW block
            rlo, rhi = eri_row_bounds(nao, world, rank)
            self.eri_rows = (rlo, rhi)
            if rhi > rlo:
                self.d_eri = torch.as_tensor(np.ascontiguousarray(inp.eri.reshape(nao * nao, nao * nao)[rlo:rhi]), dtype=f64, device=self.dev)
            # (ij|kl) = (kl|ij): when the matrix in HBM really is symmetric (checked here, once) and only J is wanted, the Coulomb
            # pass streams its upper triangle alone -- half the bytes of dft_solver.cu:550-555's GEMV, which is all a J build costs
            # ... and when it is symmetric in each index pair too, (ij|kl) = (ji|kl) = (ij|lk), the unique eighth alone (k_j_sym8;
            # the loop's dm = cocc cocc^T is symmetric bit for bit)
            if world == 1 and self.functional != "B3LYP" and self.d_eri is not None and bool(torch.equal(self.d_eri, self.d_eri.T)):
                e4 = self.d_eri.view(nao, nao, nao * nao)
                eightfold = bool(torch.equal(e4, e4.transpose(0, 1)))
                self.solver.set_option("eri_symmetric", 2 if eightfold else 1)
                del e4
        else:  # factorised J/K (DFT_ComputeJKFactorized): Cholesky vectors stay resident instead of the ERI
            if getattr(inp, "chol_range", None) is not None:      # inputs.build(world > 1) handed over this rank's slice only
                self.d_chol = torch.as_tensor(inp.chol, dtype=f64, device=self.dev)
            else:
                plo, phi = vector_bounds(inp.chol.shape[0], world, rank)
                self.d_chol = torch.as_tensor(inp.chol[plo:phi], dtype=f64, device=self.dev)
        self.nocc = inp.nocc
        # flat device buffers: [dm | cocc] arrives in one upload, [J | K | Vxc] leaves in one download
        n2 = nao * nao
        self._up = torch.zeros(n2 + nao * inp.nocc, dtype=f64, device=self.dev)
        self.d_dm, self.d_cocc = self._up[:n2].view(nao, nao), self._up[n2:].view(nao, inp.nocc)
        self._down = torch.zeros(3 * n2, dtype=f64, device=self.dev)
        self.d_J, self.d_K, self.d_v = (self._down[k * n2:(k + 1) * n2].view(nao, nao) for k in range(3))
        self._pin_up = torch.empty(self._up.shape, dtype=f64).pin_memory()
        self._pin_down = torch.empty(self._down.shape, dtype=f64).pin_memory()
        self.device_resident = (nao >= device_from) if device_resident is None else bool(device_resident)
        self.diis_device = self.dev if (nao >= 200 or self.device_resident) else None   # DIIS products on the GPU once they cost more than the hops
        self.replica_sync = ReplicaSync(self.dev, group) if world > 1 else None
        if world > 1:
            self._sharded = ShardedFock(nao, self._local_sweep, self._local_jk, self.dev, group)
        # "exact": eigh(F, S) every cycle, the reference's loop (dft.py:227); "rotate": occupied-subspace rotation with
        # the full solver as first cycle and fallback; "auto": rotate where it pays -- from 80 basis functions (per
        # cycle of the real molecules: n = 24 0.23 against 0.21 ms and n = 36 0.37 against 0.34, so not there;
        # n = 114 0.86 against 1.23; n = 246 4.3 against 6.9; n = 494 12.5 against 23.4)
        # the full solver: hipSOLVER from 400 functions whatever the loop form; the device loop needs X in HBM always
        self.eigh = FockDiagonaliser(inp.S, self.dev, device_from=0 if self.device_resident else 400)
        self.occ_solver = None
        if eigensolver not in ("auto", "rotate", "exact"):
            raise ValueError(f"eigensolver {eigensolver!r}: expected 'auto', 'rotate' or 'exact'")
        if eigensolver == "rotate" or (eigensolver == "auto" and nao >= 80):
            self.occ_solver = OccupiedRotation(inp.S, inp.nocc, self.dev if self.device_resident else None)
        # The end of the cycle (Fock assembly, DIIS, rotation, density, energy traces) as six launches of libdft.so behind the
        # cycle's J / K / Vxc (scf_tail.py, csrc/scf_tail.hip) where the rotation solver is in use and the sizes fit its
        # single-workgroup rotation kernel: Benzene/def2-SVP 0.44 ms of host work per cycle -> ~0.1 ms of device work.
        # None = auto (one rank, `device_resident` not forced off); the host and torch loops remain for everything else.
        from . import scf_tail
        self.tail = None
        # With several ranks rank 0 alone runs the tail (it is authoritative for the replicated state, grid_shard.ReplicaSync) and
        # [dm | cocc | scalars] go out in one broadcast; `self.fused` tells every rank to walk that loop.
        want_tail = (self.occ_solver is not None and device_resident is not False and ao_mode == "resident"
                     and scf_tail.supported(nao, inp.nocc)) if fused_tail is None else bool(fused_tail)
        self.fused = bool(want_tail)
        if want_tail:
            if not scf_tail.supported(nao, inp.nocc) or ao_mode != "resident":
                raise ValueError(f"fused_tail: resident AO planes, nao <= {scf_tail.MAX_NAO} and nocc <= {scf_tail.MAX_NOCC} are needed")
            if self.occ_solver is None:
                self.occ_solver = OccupiedRotation(inp.S, inp.nocc, None)      # the counters the drivers report; the rotation itself runs in the kernel
            if rank == 0:
                self.tail = scf_tail.ScfTail(self.solver.lib, inp.Hcore, inp.S, inp.nocc, self.dev)
        if self.device_resident or self.diis_device is not None or self.eigh.on_device:
            # rocBLAS / hipSOLVER load their code objects and create their handles on first use (~0.1-0.3 s in all):
            # done here, on operands of the run's own shapes, so that it is booked as initialisation -- where the
            # reference books cublasCreate (a member of XCSolver, dft_solver.cu:532, created with the solver) -- not as SCF time
            a = torch.eye(nao, dtype=f64, device=self.dev)
            (a @ a)[:, :inp.nocc].T @ a
            torch.linalg.solve(a[:9, :9], a[:9, :1])
            if self.device_resident or self.eigh.on_device:
                torch.linalg.eigh(a)
                torch.linalg.eigh(a[:inp.nocc, :inp.nocc])
            del a
        torch.cuda.synchronize()
        self.init_time = time.time() - t0

    # ---- host-loop interface ------------------------------------------------------------------
    def set_state(self, dm, cocc):
        """[dm | cocc] in ONE pinned upload (dft.py:200 uploads dm alone; cocc = sqrt(2) C_occ feeds the
        factorised exchange)."""
        n2 = self.nao * self.nao
        self._pin_up[:n2].copy_(self.torch.as_tensor(np.ascontiguousarray(dm)).reshape(-1))
        self._pin_up[n2:].copy_(self.torch.as_tensor(np.ascontiguousarray(cocc)).reshape(-1))
        self._up.copy_(self._pin_up, non_blocking=True)

    def fock_parts(self, want_k):
        """(J, K or None, Exc, Vxc_raw, seconds in the XC sweep, seconds in J/K) as numpy arrays: the device
        work of one cycle and ONE pinned download of [J | K | Vxc].  Identical on every rank."""
        t0 = time.time()
        exc, t_xc = self._device_parts(want_k)
        t_dev = time.time() - t0
        self._pin_down.copy_(self._down, non_blocking=True)
        self.torch.cuda.synchronize()
        n, n2 = self.nao, self.nao * self.nao
        h = self._pin_down.numpy()
        J, K, V = (h[k * n2:(k + 1) * n2].reshape(n, n).copy() for k in range(3))
        return J, (K if want_k else None), exc, V, t_xc, t_dev - t_xc

    # ---- device work of one cycle (both loops) --------------------------------------------------
    def _device_parts(self, want_k):
        """J, K, Vxc_raw into d_J / d_K / d_v (all-reduced over the ranks when world > 1); returns (Exc, XC seconds)."""
        if self.world > 1:
            self._want_k = want_k
            t0 = time.time()
            parts = self._sharded.compute(self.d_dm, self.d_cocc)
            self.d_v.copy_(parts.vxc); self.d_J.copy_(parts.J)
            if want_k:
                self.d_K.copy_(parts.K)
            return parts.exc, time.time() - t0
        self._jk_device(want_k)
        self.torch.cuda.synchronize()      # J/K are asynchronous: without this the XC bracket below would include them
        t0 = time.time()
        exc = self._xc_device()
        return exc, time.time() - t0

    def _jk_device(self, want_k):
        """This rank's J (and K) into d_J / d_K; zeros when it holds no vectors / no ERI rows."""
        n = self.nao
        if self.d_chol is not None and self.d_chol.shape[0]:
            self.solver.compute_jk_factorized(n, self.d_chol.shape[0], self.nocc, self.d_chol, self.d_dm,
                                              self.d_cocc if want_k else None, self.d_J, self.d_K if want_k else None)
        elif self.d_eri is not None and self.world == 1 and want_k:
            self.solver.compute_jk(n, self.d_eri, self.d_dm, self.d_J, self.d_K)      # dft.py:203 + 218 in one pass
        elif self.d_eri is not None and self.world == 1:
            self.solver.compute_coulomb(n, self.d_eri, self.d_dm, self.d_J)          # dft.py:203
        elif self.d_eri is not None:
            # row block [rlo, rhi) of the ERI: J.ravel()[rlo:rhi] = ERI[rlo:rhi, :] . vec(D) (dft_solver.cu:550-555 on
            # a row slice), K_ik += sum_jl (ij|kl) D_jl over this rank's (ij) rows (dft.py:218); the all-reduce of
            # ShardedFock assembles both (disjoint rows of J, partial sums of K)
            rlo, rhi = self.eri_rows
            self.solver.compute_jk_rows(n, rlo // n, rhi // n, self.d_eri, self.d_dm, self.d_J, self.d_K if want_k else None)
        else:
            self.d_J.zero_(); self.d_K.zero_()

    def _xc_device(self):
        if not self.ngrid:
            self.d_v.zero_()
            return 0.0
        if self.ao_mode == "direct":
            self.solver.compute_xc_direct(self.shells, self.ngrid, self.d_coords, self.d_w, self.d_dm, self.d_v, self._d_exc, self.ao_chunk)
            return float(self._d_exc.item())                                             # device sync, like the ABI call
        if self.xc_occ:
            exc = self.solver.compute_xc_occ(self.ngrid, self.nao, self.nocc, self.d_cocc, self.d_ao, self.d_w, self.d_v, self.d_gr, self.d_dm)
        else:
            exc = self.solver.compute_xc(self.ngrid, self.nao, self.d_dm, self.d_ao, self.d_w, self.d_v, self.d_gr)
        self.torch.cuda.synchronize()                                                # dft.py:205-208
        return exc

    # world > 1: the two local steps as ShardedFock wants them
    def _local_sweep(self, dm):
        return self._xc_device(), self.d_v

    def _local_jk(self, dm, cocc):
        self._jk_device(self._want_k)
        return self.d_J, (self.d_K if self._want_k else None)


def run_scf(inp, backend, functional, max_cycle=200, conv_e=1e-8, conv_dm=1e-6, log=print):
    from .hostinfo import blas_threads
    # host LAPACK/BLAS never on every visible core (256 on a 16-core share: ~90 ms stalls); below 400
    # functions one thread is fastest for everything left on the host (dsyevd at n = 114: 0.67 ms on one
    # thread, 0.87 on 16), above it the pool gets the CPU share
    with blas_threads(1 if inp.S.shape[0] < 400 else None):
        if getattr(backend, "fused", False):
            return _run_scf_fused(inp, backend, functional, max_cycle, conv_e, conv_dm, log)
        if getattr(backend, "device_resident", False):
            return _run_scf_device(inp, backend, functional, max_cycle, conv_e, conv_dm, log)
        return _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log)


def _log_header(log):
    if log:
        log("\nSCF started!"); log("-" * 80)
        log(f"{'epoch':>4} {'tot energy':>15} {'Δenergy':>12} {'Δdensity':>12} {'HF_Ex':>12}"); log("-" * 80)


def _finish(res, t_start, xc_times, jk_times, it_times):
    res["total_time"] = time.time() - t_start
    res["xc_ms_avg"] = 1e3 * sum(xc_times) / max(1, len(xc_times))     # dft.py:259 (includes the first call's allocations)
    steady = lambda ts: 1e3 * float(np.median(ts[1:] if len(ts) > 1 else ts))
    res["xc_ms"], res["jk_ms"], res["iter_ms"] = steady(xc_times), steady(jk_times), steady(it_times)  # medians past cycle 1
    res["cycle_ms"] = [round(1e3 * t, 4) for t in it_times]          # every cycle, the first (lazy allocations) included
    res["nelec_grid"] = None
    return res


def _run_scf(inp, backend, functional, max_cycle, conv_e, conv_dm, log):
    """Host form of the loop body (dft.py:199-266).  `backend` supplies either fock_parts(want_k) (HipBackend)
    or the reference-shaped pair jk(want_k) / xc() (the oracle backend of the tests)."""
    functional = functional.upper()
    c_hf = 0.2 if functional == "B3LYP" else 0.0                                       # dft.py:197
    Hcore, S, nocc = inp.Hcore, inp.S, inp.nocc
    root = getattr(backend, "rank", 0) == 0
    sync = getattr(backend, "replica_sync", None)
    solve_full = getattr(backend, "eigh", None) or (lambda F: eigh(F, S))
    occ = getattr(backend, "occ_solver", None)

    last_ddm = [None]

    rot = [occ is not None]   # cleared if the converged state fails the check against the full solver (below)

    def solve(F):   # (orbital energies, C_occ): the loop never uses the virtual orbitals (dft.py:182,228)
        if rot[0]:
            e_, co_ = occ.occupied(F, None if last_ddm[0] is None else 1e-3 * last_ddm[0])
            return np.asarray(e_), np.asarray(co_)
        e_, C_ = solve_full(F)
        return e_, C_[:, :nocc]

    e, C = solve(Hcore)                                                                # dft.py:181
    cocc = np.ascontiguousarray(np.sqrt(2.0) * C)
    dm = cocc @ cocc.T                                                                 # = 2 C_occ C_occ^T, dft.py:182
    if sync:
        sync.broadcast_numpy([dm, cocc])                                               # replicas start from rank 0's guess
    diis = CDIIS(device=getattr(backend, "diis_device", None))
    _log_header(log)
    E_old, xc_times, jk_times, it_times, t_start = 0.0, [], [], [], time.time()
    res = {"converged": False}
    want_k = functional == "B3LYP"
    import os
    prof = [] if os.environ.get("QCDFT_SCF_PROFILE") else None      # per-part times of the host side of a cycle
    for cycle in range(max_cycle):
        t_it = time.time()
        if hasattr(backend, "fock_parts"):
            backend.set_state(dm, cocc)                                                # dft.py:200
            J, K, E_xc, Vraw, t_xc, t_jk = backend.fock_parts(want_k)
        else:
            backend.set_dm(dm)
            J, K = backend.jk(want_k)
            t_jk = time.time() - t_it
            E_xc, Vraw, t_xc = backend.xc()
        jk_times.append(t_jk); xc_times.append(t_xc)
        dm_new, cocc_new, scal = np.empty_like(dm), np.empty_like(cocc), np.zeros(4)
        if root or sync is None:   # rank 0 is authoritative: DIIS + eigh run once, replicas receive the result
            tp = [time.time()]
            Vxc = 0.5 * (Vraw + Vraw.T)                                                # dft.py:212
            F = Hcore + J + Vxc - (c_hf * 0.5 * K if K is not None else 0.0)           # dft.py:221,223
            tp.append(time.time())
            F = diis.update(S, dm, F, cocc=cocc)
            tp.append(time.time())
            e, C = solve(F)
            tp.append(time.time())
            cocc_new = np.ascontiguousarray(np.sqrt(2.0) * C)
            dm_new = cocc_new @ cocc_new.T
            scal = np.array([np.sum(dm_new * Hcore), 0.5 * np.sum(dm_new * J),
                             -0.25 * c_hf * np.sum(dm_new * K) if K is not None else 0.0,
                             np.linalg.norm(dm_new - dm)])
            tp.append(time.time())
            if prof is not None:
                prof.append([1e3 * (tp[0] - t_it - t_jk - t_xc)] + [1e3 * (b - a) for a, b in zip(tp, tp[1:])])
        if sync:
            sync.broadcast_numpy([dm_new, cocc_new, scal])
        E_one, E_coul, E_ex, ddm = (float(x) for x in scal)
        last_ddm[0] = ddm
        E_tot = E_one + E_coul + E_xc + E_ex + inp.E_nuc                               # dft.py:236
        dE = E_tot - E_old
        it_times.append(time.time() - t_it)
        if log:
            log(f"{cycle + 1:4d} {E_tot:18.8f} {dE:15.6e} {ddm:15.6e} {E_ex:12.6f}")
        res.update(E_tot=E_tot, E_one=E_one, E_coul=E_coul, E_xc=E_xc, E_ex_hf=E_ex, cycles=cycle + 1,
                   dm=dm_new, mo_energy=e)
        if abs(dE) < conv_e and ddm < conv_dm:                                         # dft.py:243; same scalars on every rank
            # The rotation solver follows the occupied space continuously; that it is still the AUFBAU one is checked
            # once, here, against eigh(F, S) of the converged Fock matrix (which also supplies the exact orbital
            # energies).  A mismatch (an occupied/virtual level crossing it followed through) resumes the loop with the
            # full solver -- the reference's loop.  Rank 0 decides, every rank hears it.
            ok = np.ones(1)
            if rot[0]:
                if root or sync is None:
                    e_x, C_x = solve_full(F)
                    e_x, C_x = np.asarray(e_x), np.asarray(C_x)
                    ok[0] = float(np.linalg.norm(2.0 * C_x[:, :nocc] @ C_x[:, :nocc].T - dm_new) < 1e-4)
                    if ok[0]:
                        res["mo_energy"] = e_x
                if sync:
                    sync.broadcast_numpy([ok])
            if ok[0]:
                res["converged"] = True
                break
            rot[0] = False
            diis = CDIIS(device=getattr(backend, "diis_device", None))   # its history belongs to the other state
            if log:
                log("     converged occupied space is not the aufbau one: continuing with eigh(F, S) every cycle")
        dm, cocc, E_old = dm_new, cocc_new, E_tot
    if prof:
        med = np.median(np.array(prof[1:] if len(prof) > 1 else prof), axis=0)
        res["host_parts_ms"] = dict(zip(("transfers", "fock", "diis", "eigen", "density_energies"), (float(x) for x in med)))
        if log:
            log("host parts (median ms per cycle): " + ", ".join(f"{k} {v:.3f}" for k, v in res["host_parts_ms"].items()))
    return _finish(res, t_start, xc_times, jk_times, it_times)


def _run_scf_device(inp, backend, functional, max_cycle, conv_e, conv_dm, log):
    """Device-resident form of the same loop: nothing but scalars crosses PCIe per cycle."""
    t = backend.torch
    dev, f64 = backend.dev, backend.torch.float64
    functional = functional.upper()
    c_hf = 0.2 if functional == "B3LYP" else 0.0
    nocc = inp.nocc
    root, sync = backend.rank == 0, backend.replica_sync
    H = t.as_tensor(inp.Hcore, dtype=f64, device=dev)
    S = t.as_tensor(inp.S, dtype=f64, device=dev)
    X = backend.eigh.X                                                                # S^-1/2 (FockDiagonaliser, once per S)
    sqrt2 = float(np.sqrt(2.0))

    last_ddm = [None]

    rot = [backend.occ_solver is not None]   # see the convergence check in _run_scf

    def eigh_full(F):
        if X.shape[0] < 400:      # as OccupiedRotation._exact: one host LAPACK thread beats hipSOLVER below ~400 functions
            e_, Cp = eigh((X.T @ F @ X).cpu().numpy(), driver="evd")
            return t.as_tensor(e_, device=dev), X @ t.as_tensor(Cp, device=dev)
        e_, Cp = t.linalg.eigh(X.T @ F @ X)
        return e_, X @ Cp

    def eigh_occ(F):                                                                   # dft.py:181,227 on the device
        if rot[0]:
            e, co = backend.occ_solver.occupied(F, None if last_ddm[0] is None else 1e-3 * last_ddm[0])
            return e, co * sqrt2
        e, C = eigh_full(F)
        return e, C[:, :nocc] * sqrt2

    e, cocc = eigh_occ(H)
    dm = cocc @ cocc.T
    if sync:
        sync.broadcast([dm, cocc])
    diis = CDIIS(device=dev)
    _log_header(log)
    E_old, xc_times, jk_times, it_times, t_start = 0.0, [], [], [], time.time()
    res = {"converged": False}
    want_k = functional == "B3LYP"
    scal = t.zeros(4, dtype=f64, device=dev)
    import os
    prof = [] if os.environ.get("QCDFT_SCF_PROFILE") else None      # per-part times of the host-side-of-the-cycle (adds syncs)
    for cycle in range(max_cycle):
        t_it = time.time()
        backend.d_dm.copy_(dm); backend.d_cocc.copy_(cocc)                             # device to device
        E_xc, t_xc = backend._device_parts(want_k)
        t.cuda.synchronize()
        xc_times.append(t_xc); jk_times.append(time.time() - t_it - t_xc)
        J, K, V = backend.d_J, backend.d_K, backend.d_v
        if root or sync is None:
            tp = [time.time()]
            mark = (lambda: (t.cuda.synchronize(), tp.append(time.time()))) if prof is not None else (lambda: None)
            F = H + J + 0.5 * (V + V.T)                                                # dft.py:212,223
            if want_k:
                F = F - (0.5 * c_hf) * K                                               # dft.py:221
            mark()
            F = diis.update(S, dm, F, keep_on_device=True, cocc=cocc)
            mark()
            e, cocc_new = eigh_occ(F)
            mark()
            dm_new = cocc_new @ cocc_new.T
            dv = dm_new.reshape(-1)
            tr = backend._down.view(3, -1) @ dv                # [J:D, K:D, Vraw:D] in one launch (J, K, V are its rows)
            scal = t.stack([t.dot(dv, H.reshape(-1)), 0.5 * tr[0], (-0.25 * c_hf if want_k else 0.0) * tr[1],
                            t.linalg.norm(dm_new - dm)])
            mark()
            if prof is not None and len(tp) == 5:
                prof.append([1e3 * (b - a) for a, b in zip(tp, tp[1:])])
        else:
            dm_new, cocc_new = t.empty_like(dm), t.empty_like(cocc)
        if sync:
            sync.broadcast([dm_new, cocc_new, scal])
        E_one, E_coul, E_ex, ddm = scal.tolist()                                       # the cycle's only download
        last_ddm[0] = ddm
        E_tot = E_one + E_coul + E_xc + E_ex + inp.E_nuc
        dE = E_tot - E_old
        it_times.append(time.time() - t_it)
        if log:
            log(f"{cycle + 1:4d} {E_tot:18.8f} {dE:15.6e} {ddm:15.6e} {E_ex:12.6f}")
        res.update(E_tot=E_tot, E_one=E_one, E_coul=E_coul, E_xc=E_xc, E_ex_hf=E_ex, cycles=cycle + 1)
        if abs(dE) < conv_e and ddm < conv_dm:
            ok = t.ones(1, dtype=f64, device=dev)
            if rot[0]:                                                                 # as in _run_scf
                if root or sync is None:
                    e_x, C_x = eigh_full(F)
                    ok[0] = float(float(t.linalg.norm(2.0 * C_x[:, :nocc] @ C_x[:, :nocc].T - dm_new)) < 1e-4)
                    if float(ok[0]):
                        e = e_x
                if sync:
                    sync.broadcast([ok])
            if float(ok[0]):
                res["converged"] = True
                dm = dm_new
                break
            rot[0] = False
            diis = CDIIS(device=dev)
            if log:
                log("     converged occupied space is not the aufbau one: continuing with eigh(F, S) every cycle")
        dm, cocc, E_old = dm_new, cocc_new, E_tot
    res["dm"] = dm.cpu().numpy()
    res["mo_energy"] = e.cpu().numpy()
    if prof:
        med = np.median(np.array(prof[1:] if len(prof) > 1 else prof), axis=0)
        res["device_parts_ms"] = dict(zip(("fock", "diis", "eigen", "density_energies"), (float(x) for x in med)))
        if log:
            log("device-resident parts (median ms per cycle): " + ", ".join(f"{k} {v:.3f}" for k, v in res["device_parts_ms"].items()))
    return _finish(res, t_start, xc_times, jk_times, it_times)


def _run_scf_fused(inp, backend, functional, max_cycle, conv_e, conv_dm, log):
    """The loop with its host part on the device (scf_tail.ScfTail): per cycle the J / K and XC kernels, then the tail's
    launches for dft.py:212-236, then ONE wait on host-mapped memory for the energy / convergence scalars.  The few full
    diagonalisations of a run (first cycle, refused rotations, the final aufbau check) are LAPACK calls on the host below
    400 functions and hipSOLVER above, as in the other two loops.  With several ranks the cycle's device work is the sharded
    one (grid_shard.ShardedFock: local sweep + local J/K + one all-reduce), rank 0 alone runs the tail, and
    [dm | cocc | scalars] reach the other ranks in one broadcast."""
    t = backend.torch
    tail, rot_stats = backend.tail, backend.occ_solver.stats
    root, sync, world = backend.rank == 0, backend.replica_sync, backend.world
    functional = functional.upper()
    c_hf = 0.2 if functional == "B3LYP" else 0.0
    want_k = functional == "B3LYP"
    nocc, Xh = inp.nocc, backend.eigh.Xh
    sqrt2 = float(np.sqrt(2.0))
    on_dev = inp.S.shape[0] >= 400          # hipSOLVER from 400 functions, one LAPACK thread below (FockDiagonaliser)
    Xd = t.as_tensor(Xh, dtype=t.float64, device=backend.dev) if (on_dev and root) else None

    def full_solve(F_dev):                                                             # dft.py:181,227: (energies, eigenvectors on the device)
        if on_dev:
            e_, Cp = t.linalg.eigh(Xd.T @ F_dev @ Xd)
            return e_.cpu().numpy(), Xd @ Cp
        e_, Cp = eigh(Xh.T @ F_dev.cpu().numpy() @ Xh, driver="evd")
        return e_, t.from_numpy(np.ascontiguousarray(Xh @ Cp)).to(backend.dev)

    def diagonalise_into_basis(F_dev):
        e_, U = full_solve(F_dev)
        tail.basis.copy_(U)
        rot_stats["exact"] += 1
        return e_

    e = None
    if root:
        tail.reset()
        e = diagonalise_into_basis(tail.d_h)
        backend.d_cocc.copy_(sqrt2 * tail.basis[:, :nocc]); backend.d_dm.copy_(backend.d_cocc @ backend.d_cocc.T)   # dft.py:182
    scal = t.zeros(8, dtype=t.float64, device=backend.dev)                           # what travels with [dm | cocc] when there are replicas
    if sync:
        sync.broadcast([backend._up])                                                  # [dm | cocc] is one flat buffer (HipBackend)
    _log_header(log)
    E_old, xc_times, jk_times, it_times, t_start = 0.0, [], [], [], time.time()
    res = {"converged": False}
    rotate, last_ddm = True, None
    d_K = backend.d_K if want_k else None
    d_exc = t.zeros(1, dtype=t.float64, device=backend.dev)
    marks, tail_log = [], []                                                           # (start, J/K done, XC done) events; (status, fixed-point steps, Jacobi sweeps)
    sol = backend.solver
    pool = [t.cuda.Event(enable_timing=True) for _ in range(3 * 40)] if world == 1 else []   # made here: not in the cycles' time
    # One rank: the next cycle's J/K and sweep are queued BEHIND this cycle's tail before the host has seen its result -- they only
    # need dm / cocc, which the tail leaves in place -- so the GPU never waits for the host between cycles (30 us of a 0.65 ms
    # Benzene cycle).  Two sets of [J | K | Vxc] in turn: a cycle whose rotation is refused after all still owns intact matrices
    # for DFT_ScfTailFinish, and the parts queued ahead of it (from the density that was not replaced) are simply queued again.
    n2 = backend.nao * backend.nao
    sets = [(backend.d_J, backend.d_K, backend.d_v)]
    if world == 1:
        spare = t.zeros_like(backend._down)
        sets.append(tuple(spare[k * n2:(k + 1) * n2].view(backend.nao, backend.nao) for k in range(3)))

    def enqueue_parts(k):
        backend.d_J, backend.d_K, backend.d_v = sets[k % 2]
        ev = pool[3 * len(marks):3 * len(marks) + 3] if 3 * len(marks) + 3 <= len(pool) else [t.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        backend._jk_device(want_k)
        ev[1].record()
        if backend.xc_occ:
            sol.compute_xc_occ_async(backend.ngrid, backend.nao, nocc, backend.d_cocc, backend.d_ao, backend.d_w, backend.d_v, d_exc,
                                     backend.d_gr, backend.d_dm)
        else:
            sol.compute_xc_async(backend.ngrid, backend.nao, backend.d_dm, backend.d_ao, backend.d_w, backend.d_v, d_exc, backend.d_gr)
        ev[2].record()
        marks.append(ev)

    queued, last_status = -1, None
    for cycle in range(max_cycle):
        t_it = time.time()
        if world == 1:
            if queued < cycle:
                enqueue_parts(cycle); queued = cycle
            d_J, d_Kc, d_V = sets[cycle % 2]
            d_K = d_Kc if want_k else None
            E_xc = None
        else:
            d_J, d_V = backend.d_J, backend.d_v
            E_xc, t_xc = backend._device_parts(want_k)                                 # sharded: ends in the all-reduce of [Vxc | J | K | Exc]
            xc_times.append(t_xc); jk_times.append(time.time() - t_it - t_xc)
        if root:
            tol = 1e-10 if last_ddm is None else min(max(1e-10, 1e-3 * last_ddm), 1e-5)    # as OccupiedRotation.occupied(accuracy)
            tail.step(rotate, c_hf, tol, d_J, d_K, d_V, backend.d_dm, backend.d_cocc, d_exc=d_exc if world == 1 else None)
            if world == 1 and last_status == 0 and rotate:                             # the last rotation went through: expect this one to
                enqueue_parts(cycle + 1); queued = cycle + 1
            E_one, E_coul, E_ex, ddm, status, steps, sweeps, exc_dev = tail.wait()
            tail_log.append((status, steps, sweeps))
            last_status = status
            if status != 0 and queued > cycle:                                         # queued ahead from a density that stays: not this cycle's successor
                queued = cycle; marks.pop()
            if status == 2:                                                            # singular Pulay system: least squares on the host
                tail.step(rotate, c_hf, tol, d_J, d_K, d_V, backend.d_dm, backend.d_cocc,
                          coef=tail.pulay_coefficients_on_host(), repeat=True)
                E_one, E_coul, E_ex, ddm, status, steps, _, _ = tail.wait()
            if status == 1:                                                            # no rotation (asked for, or possible): full solve
                e = diagonalise_into_basis(tail.fock)
                tail.finish(c_hf, d_J, d_K, backend.d_dm, backend.d_cocc)
                E_one, E_coul, E_ex, ddm, status, _, _, _ = tail.wait()
            else:
                rot_stats["rotated"] += 1; rot_stats["inner_steps"] += steps
            if status != 0:
                raise RuntimeError(f"SCF tail: status {status}")
            if world == 1:
                E_xc = exc_dev
        if sync:
            if root:
                scal[:4] = t.tensor([E_one, E_coul, E_ex, ddm], dtype=t.float64)
            sync.broadcast([backend._up, scal])
            E_one, E_coul, E_ex, ddm = scal[:4].tolist()
        last_ddm = ddm
        E_tot = E_one + E_coul + E_xc + E_ex + inp.E_nuc
        dE = E_tot - E_old
        it_times.append(time.time() - t_it)
        if log:
            log(f"{cycle + 1:4d} {E_tot:18.8f} {dE:15.6e} {ddm:15.6e} {E_ex:12.6f}")
        res.update(E_tot=E_tot, E_one=E_one, E_coul=E_coul, E_xc=E_xc, E_ex_hf=E_ex, cycles=cycle + 1)
        if abs(dE) < conv_e and ddm < conv_dm:                                         # the same scalars on every rank
            ok = t.ones(1, dtype=t.float64, device=backend.dev)
            if rotate:                                                                 # as in _run_scf: the followed space against eigh(F, S)
                if root:
                    e_x, C_x = full_solve(tail.fock)
                    ok[0] = float(float(t.linalg.norm(2.0 * C_x[:, :nocc] @ C_x[:, :nocc].T - backend.d_dm)) < 1e-4)
                    if float(ok[0]):
                        e = e_x
                if sync:
                    sync.broadcast([ok])
            if float(ok[0]):
                res["converged"] = True
                break
            rotate = False
            if root:
                tail.reset()
            if log:
                log("     converged occupied space is not the aufbau one: continuing with eigh(F, S) every cycle")
        E_old = E_tot
    backend.d_J, backend.d_K, backend.d_v = sets[0]
    res["dm"] = backend.d_dm.cpu().numpy()                                             # (waits for anything still queued ahead)
    res["mo_energy"] = None if e is None else np.asarray(e)
    res["loop"] = "fused"
    res["tail_log"] = tail_log
    for ev in marks:                                                                   # device-side durations: nothing waited in between
        jk_times.append(1e-3 * ev[0].elapsed_time(ev[1])); xc_times.append(1e-3 * ev[1].elapsed_time(ev[2]))
    return _finish(res, t_start, xc_times, jk_times, it_times)
