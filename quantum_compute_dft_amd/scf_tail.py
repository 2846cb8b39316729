"""Host side of the device-resident end of an SCF cycle (libdft.so: csrc/scf_tail.hip, DFT_ScfTail*): what dft.py:212-236
does with numpy between the cycle's J / K / Vxc and the next density -- Fock assembly, Pulay DIIS on the commutator,
the eigenproblem (as the occupied-subspace rotation of scf.OccupiedRotation), dm = 2 C_occ C_occ^T and the energy
traces -- queued as six launches behind the kernels that produced J, K and Vxc.  The host keeps the ring bookkeeping,
the few full diagonalisations of a run (LAPACK, as scf.OccupiedRotation._exact) and the convergence test."""
import ctypes

import numpy as np

MAX_NAO, MAX_NOCC, SPACE = 512, 64, 8      # up to 128 x 32 the rotation's matrices live in LDS (k_tail_rot), above in memory (k_tail_rot_big)
STATUS_DONE, STATUS_DIAGONALISE, STATUS_SINGULAR, STATUS_MORE = 0, 1, 2, 3


def supported(nao, nocc):
    return 2 <= nao <= MAX_NAO and 1 <= nocc <= MAX_NOCC and nocc < nao


class ScfTail:
    def __init__(self, lib, hcore, overlap, nocc, device):
        """`lib`: the loaded libdft.so (ctypes); hcore, overlap: (nao, nao) arrays; tensors live on `device`."""
        import torch
        self.torch, self.lib = torch, lib
        u64, dp, ip = ctypes.c_uint64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        lib.DFT_ScfTailOpen.argtypes = [ctypes.c_int, ctypes.c_int, u64, u64, u64, u64, u64]
        lib.DFT_ScfTailOpen.restype = ctypes.c_void_p
        lib.DFT_ScfTailSetStream.argtypes = [ctypes.c_void_p, u64]
        lib.DFT_ScfTailStep.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ip, dp, u64, u64, u64, u64, u64, u64]
        lib.DFT_ScfTailFinish.argtypes = [ctypes.c_void_p, ctypes.c_double, u64, u64, u64, u64]
        lib.DFT_ScfTailMore.argtypes = [ctypes.c_void_p, ctypes.c_int, u64, u64, u64, u64, u64]
        lib.DFT_ScfTailSetStepsHint.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.DFT_ScfTailWait.argtypes = [ctypes.c_void_p, dp]
        lib.DFT_ScfTailGram.argtypes = [ctypes.c_void_p, dp]
        lib.DFT_ScfTailLastError.argtypes = [ctypes.c_void_p]
        lib.DFT_ScfTailLastError.restype = ctypes.c_char_p
        lib.DFT_ScfTailClose.argtypes = [ctypes.c_void_p]
        lib.DFT_ScfTailClose.restype = None
        n = int(np.asarray(hcore).shape[0])
        self.nao, self.nocc = n, int(nocc)
        f64 = torch.float64
        self.d_h = torch.as_tensor(np.ascontiguousarray(hcore), dtype=f64, device=device)
        self.d_s = torch.as_tensor(np.ascontiguousarray(overlap), dtype=f64, device=device)
        self.basis = torch.zeros((n, n), dtype=f64, device=device)       # S-orthonormal columns, occupied first
        self.fock = torch.zeros((n, n), dtype=f64, device=device)        # DIIS-extrapolated Fock matrix of the last step
        self.mo_energy = torch.zeros(n, dtype=f64, device=device)
        self._h = lib.DFT_ScfTailOpen(n, self.nocc, self.d_h.data_ptr(), self.d_s.data_ptr(), self.basis.data_ptr(),
                                      self.fock.data_ptr(), self.mo_energy.data_ptr())
        if not self._h:
            raise RuntimeError(f"DFT_ScfTailOpen failed (nao {n} <= {MAX_NAO}, nocc {nocc} <= {MAX_NOCC} and a device are needed)")
        lib.DFT_ScfTailSetStream(self._h, torch.cuda.current_stream(device).cuda_stream)
        self._out = (ctypes.c_double * 8)()
        self.reset()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.DFT_ScfTailClose(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        """Forget the DIIS history (the basis stays)."""
        self.hist = []

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libdft: " + (self.lib.DFT_ScfTailLastError(self._h) or b"").decode())

    def step(self, rotate, c_hf, tol, d_J, d_K, d_vraw, d_dm, d_cocc, coef=None, canon_tol=0.0, max_inner=60, repeat=False, d_exc=None):
        """Queue one cycle's end.  `repeat`: the same cycle again (after status 2), into the same ring slot.  `d_exc`: the
        device scalar of a DFT_ComputeXC*Async call queued before; wait() then returns Exc too."""
        if not repeat:
            slot = self.hist.pop(0) if len(self.hist) == SPACE else len(self.hist)   # the oldest pair is overwritten (scf.CDIIS)
            self.hist.append(slot)
        slot = self.hist[-1]
        hist = (ctypes.c_int * len(self.hist))(*self.hist)
        cf = None if coef is None else (ctypes.c_double * len(self.hist))(*[float(x) for x in coef])
        self._last = (d_J, d_K, d_dm, d_cocc, d_exc)       # DFT_ScfTailMore continues this step (memory-resident rotation, status 3)
        self._check(self.lib.DFT_ScfTailStep(self._h, int(bool(rotate)), float(c_hf), float(tol), float(canon_tol), int(max_inner), slot,
                                             len(self.hist), hist, cf, d_J.data_ptr(), 0 if d_K is None else d_K.data_ptr(),
                                             d_vraw.data_ptr(), d_dm.data_ptr(), d_cocc.data_ptr(),
                                             0 if d_exc is None else d_exc.data_ptr()))

    def finish(self, c_hf, d_J, d_K, d_dm, d_cocc):
        self._check(self.lib.DFT_ScfTailFinish(self._h, float(c_hf), d_J.data_ptr(), 0 if d_K is None else d_K.data_ptr(),
                                               d_dm.data_ptr(), d_cocc.data_ptr()))

    def wait(self):
        """(tr(dm' Hcore), tr(dm' J)/2, -c_hf tr(dm' K)/4, |dm' - dm|, status, fixed-point steps, Jacobi sweeps, Exc)"""
        self._check(self.lib.DFT_ScfTailWait(self._h, self._out))
        o = self._out
        more = 8
        while int(o[4]) == STATUS_MORE:                    # above 128 functions the fixed point's steps are launches of their own,
            d_J, d_K, d_dm, d_cocc, d_exc = self._last     # queued in advance: this many were not enough -- queue more, wait again
            self._check(self.lib.DFT_ScfTailMore(self._h, more, d_J.data_ptr(), 0 if d_K is None else d_K.data_ptr(), d_dm.data_ptr(),
                                                 d_cocc.data_ptr(), 0 if d_exc is None else d_exc.data_ptr()))
            self._check(self.lib.DFT_ScfTailWait(self._h, self._out))
            more = min(2 * more, 32)
        if int(o[4]) == STATUS_DONE and int(o[5]) > 0:     # next time: two more than this cycle needed
            self.lib.DFT_ScfTailSetStepsHint(self._h, min(max(int(o[5]) + 2, 3), 16))
        return o[0], o[1], o[2], o[3], int(o[4]), int(o[5]), int(o[6]), o[7]

    def gram(self):
        g = np.zeros((SPACE, SPACE))
        self._check(self.lib.DFT_ScfTailGram(self._h, g.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return g

    def pulay_coefficients_on_host(self):
        """Least-squares solution of the DIIS system from the ring's Gram matrix (scf.CDIIS's fallback)."""
        idx = np.array(self.hist)
        G = self.gram()[np.ix_(idx, idx)]
        m = len(idx)
        B = np.zeros((m + 1, m + 1)); B[0, 1:] = B[1:, 0] = 1.0; B[1:, 1:] = G
        rhs = np.zeros(m + 1); rhs[0] = 1.0
        return np.linalg.lstsq(B, rhs, rcond=None)[0][1:]
