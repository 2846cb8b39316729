"""S, T, V, dense ERI and E_nuc for the SCF driver -- host-side counterpart of the PySCF calls at
grid.py:61-66 (`mol.intor('int1e_ovlp'|'int1e_kin'|'int1e_nuc'|'int2e')`, `mol.energy_nuc()`).
The arithmetic is csrc/integrals.c (McMurchie-Davidson, OpenMP), built in-tree with gcc.

PARITY UNPINNED against PySCF/libcint (not installed); pinned offline by quadrature on the
Becke grid, textbook H2/STO-3G integrals and RHF energies (tests/test_integrals.py)."""
import ctypes
import os
import subprocess

import numpy as np

from .basis import atomic_number
from .build import CSRC, LIB_DIR, _stamp_ok, build_lock, compile_env, source_hash

_LIB_PATH = os.path.join(LIB_DIR, "libqcint.so")
_SRC = os.path.join(CSRC, "integrals.c")
_FLAGS = ["-O2", "-fPIC", "-shared", "-fopenmp", "-std=c11"]
_lib = None


def build_integrals(force=False):
    """gcc -> lib/libqcint.so.  Called by __graft_entry__.build() and the test fixtures only; the
    loader below never compiles (same contract as build.library_path)."""
    want = source_hash([_SRC], _FLAGS)
    if force or not _stamp_ok(_LIB_PATH, _LIB_PATH + ".srchash", want):
        with build_lock(_LIB_PATH):
            if force or not _stamp_ok(_LIB_PATH, _LIB_PATH + ".srchash", want):
                tmp = _LIB_PATH + f".tmp{os.getpid()}"
                subprocess.run(["gcc"] + _FLAGS + [_SRC, "-o", tmp, "-lm"], check=True, env=compile_env())
                os.replace(tmp, _LIB_PATH)
                with open(_LIB_PATH + ".srchash", "w") as fh:
                    fh.write(want + "\n")
    return _LIB_PATH


def integrals_library_path():
    if not _stamp_ok(_LIB_PATH, _LIB_PATH + ".srchash", source_hash([_SRC], _FLAGS)):
        raise RuntimeError(f"{_LIB_PATH} is missing or older than csrc/integrals.c: run `python __graft_entry__.py` first")
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(integrals_library_path())
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        L.qc_int1e.restype = ctypes.c_int
        L.qc_int1e.argtypes = [ctypes.c_int, dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp]
        L.qc_int2e.restype = ctypes.c_int
        L.qc_int2e.argtypes = [ctypes.c_int, dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, dp]
        L.qc_eri_open.restype = ctypes.c_void_p
        L.qc_eri_open.argtypes = [ctypes.c_int, dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, ctypes.c_int]
        L.qc_eri_close.restype = None
        L.qc_eri_close.argtypes = [ctypes.c_void_p]
        L.qc_eri_diag.restype = ctypes.c_int
        L.qc_eri_diag.argtypes = [ctypes.c_void_p, dp]
        L.qc_eri_cols.restype = ctypes.c_int
        L.qc_eri_cols.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp]
        L.qc_eri_cols2.restype = ctypes.c_int
        L.qc_eri_cols2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int]
        L.qc_set_threads.restype = None
        L.qc_set_threads.argtypes = [ctypes.c_int]
        from .hostinfo import host_cpu_share
        L.qc_set_threads(host_cpu_share())
        _lib = L
    return _lib


def _args(sh):
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    keep = [np.ascontiguousarray(sh.xyz, dtype=np.float64), np.ascontiguousarray(sh.l, dtype=np.int32),
            np.ascontiguousarray(sh.nprim, dtype=np.int32), np.ascontiguousarray(sh.off, dtype=np.int32),
            np.ascontiguousarray(sh.ao, dtype=np.int32), np.ascontiguousarray(sh.exp, dtype=np.float64),
            np.ascontiguousarray(sh.coef, dtype=np.float64)]
    ptrs = [keep[0].ctypes.data_as(dp)] + [k.ctypes.data_as(ip) for k in keep[1:5]] + [k.ctypes.data_as(dp) for k in keep[5:]]
    return keep, ptrs


def set_threads(n):
    """Worker threads of the integral engine's OpenMP regions (default: this rank's CPU share, hostinfo.py)."""
    _load().qc_set_threads(int(n))


def int1e(shells, symbols, atom_xyz):
    """(S, T, V) as (nao, nao) arrays; V is the nuclear-attraction matrix (negative)."""
    keep, p = _args(shells)
    n = shells.nao
    S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    axyz = np.ascontiguousarray(atom_xyz, dtype=np.float64)
    z = np.array([atomic_number(s) for s in symbols], dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = _load().qc_int1e(shells.nshell, *p, n, len(z), axyz.ctypes.data_as(dp), z.ctypes.data_as(dp),
                          S.ctypes.data_as(dp), T.ctypes.data_as(dp), V.ctypes.data_as(dp))
    if rc != 0:
        raise ValueError("integrals: angular momentum above f is not supported")
    return S, T, V


def int2e(shells):
    """Dense (nao, nao, nao, nao) electron-repulsion tensor, chemists' notation (ij|kl)."""
    keep, p = _args(shells)
    n = shells.nao
    eri = np.zeros((n, n, n, n))
    rc = _load().qc_int2e(shells.nshell, *p, n, eri.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc != 0:
        raise ValueError("integrals: angular momentum above f is not supported")
    return eri


def energy_nuc(symbols, atom_xyz):
    z = np.array([atomic_number(s) for s in symbols], dtype=np.float64)
    xyz = np.asarray(atom_xyz, dtype=np.float64)
    e = 0.0
    for i in range(len(z)):
        for j in range(i):
            e += z[i] * z[j] / np.linalg.norm(xyz[i] - xyz[j])
    return e


class EriColumns:
    """Column-wise access to the ERI without the dense tensor (for the pivoted Cholesky
    factorisation of cholesky.py): `diag()` = (ij|ij), `cols(C, D)` = every (ij|kl) with k in shell
    C and l in shell D."""

    def __init__(self, shells):
        keep, p = _args(shells)
        self.shells, self.nao = shells, shells.nao
        self._h = _load().qc_eri_open(shells.nshell, *p, shells.nao, len(shells.exp))
        if not self._h:
            raise ValueError("integrals: angular momentum above f is not supported")

    def close(self):
        if getattr(self, "_h", None):
            _load().qc_eri_close(self._h)
            self._h = None

    __del__ = close

    def diag(self):
        d = np.zeros((self.nao, self.nao))
        _load().qc_eri_diag(self._h, d.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return d

    def cols(self, C, D, screen=1e-14, out=None, lower_only=False):
        """(nfC*nfD, nao, nao): entry [k*nfD + l] is the symmetric matrix (..|kl).  `out`: a C-contiguous float64
        buffer of nfC*nfD*nao^2 elements to write into (e.g. pinned memory), returned reshaped.  `lower_only`: only the
        elements [i][j] with i >= j are written (the rest stay zero); the caller mirrors them."""
        nf = (2 * int(self.shells.l[C]) + 1) * (2 * int(self.shells.l[D]) + 1)
        if out is None:
            out = np.empty((nf, self.nao, self.nao))
        else:
            assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.size == nf * self.nao * self.nao
        rc = _load().qc_eri_cols2(self._h, int(C), int(D), float(screen), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(bool(lower_only)))
        if rc != 0:
            raise RuntimeError("qc_eri_cols failed (diag() must be called first)")
        return out.reshape(nf, self.nao, self.nao)


def schwarz_bounds(shells, diag):
    """sqrt(max (ab|ab)) per shell pair a >= b, in the order a (a + 1) / 2 + b (what integrals.c keeps internally after
    qc_eri_diag), from the (nao, nao) diagonal `diag` of the ERI matrix."""
    ao0 = np.asarray(shells.ao, dtype=np.int64)
    blk = np.maximum.reduceat(np.maximum.reduceat(np.asarray(diag), ao0, axis=0), ao0, axis=1)     # (nshell, nshell) block maxima
    a, b = np.tril_indices(shells.nshell)                                                          # row-major: a (a + 1) / 2 + b
    return np.sqrt(np.maximum(blk[a, b], 0.0))


class DeviceEriColumns:
    """The same columns computed ON THE DEVICE (libdft.so: csrc/eri_cols.hip, DFT_EriColumns) into a torch tensor: the
    pivoted Cholesky factorisation keeps its algebra there, so the columns never cross PCIe.  `qmax`: schwarz_bounds()."""

    def __init__(self, shells, qmax, lib_path=None):
        from .build import library_path
        from .solver import load_library
        self.lib = load_library(lib_path or library_path())
        L = self.lib
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        L.DFT_EriColumnsOpen.argtypes = [ctypes.c_int, dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, ctypes.c_int, dp]
        L.DFT_EriColumnsOpen.restype = ctypes.c_void_p
        L.DFT_EriColumns.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_uint64]
        L.DFT_EriColumns.restype = ctypes.c_int
        L.DFT_EriColumnsMany.argtypes = [ctypes.c_void_p, ctypes.c_int, ip, ip, ctypes.c_double, ctypes.c_uint64, ctypes.POINTER(ctypes.c_longlong)]
        L.DFT_EriColumnsMany.restype = ctypes.c_int
        L.DFT_EriColumnsLastError.argtypes = [ctypes.c_void_p]
        L.DFT_EriColumnsLastError.restype = ctypes.c_char_p
        L.DFT_EriColumnsClose.argtypes = [ctypes.c_void_p]
        L.DFT_EriColumnsClose.restype = None
        keep, p = _args(shells)
        q = np.ascontiguousarray(qmax, dtype=np.float64)
        assert q.shape == (shells.nshell * (shells.nshell + 1) // 2,)
        self.shells, self.nao = shells, shells.nao
        self._h = L.DFT_EriColumnsOpen(shells.nshell, *p, shells.nao, len(shells.exp), q.ctypes.data_as(dp))
        if not self._h:
            raise RuntimeError("DFT_EriColumnsOpen failed (no device, or angular momentum above f)")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.DFT_EriColumnsClose(self._h)
            self._h = None

    __del__ = close

    def cols(self, C, D, screen, out):
        """Fills the torch tensor `out` ((2 l_C + 1)(2 l_D + 1) x nao x nao doubles on the device; elements i >= j only) and
        returns it viewed as (nq, nao, nao).  Asynchronous on the null stream, like torch's own kernels."""
        nq = (2 * int(self.shells.l[C]) + 1) * (2 * int(self.shells.l[D]) + 1)
        assert out.is_cuda and out.is_contiguous() and out.numel() >= nq * self.nao * self.nao
        if self.lib.DFT_EriColumns(self._h, int(C), int(D), float(screen), ctypes.c_uint64(out.data_ptr())) != 0:
            raise RuntimeError("libdft: " + (self.lib.DFT_EriColumnsLastError(self._h) or b"").decode())
        return out.view(-1)[:nq * self.nao * self.nao].view(nq, self.nao, self.nao)

    def cols_many(self, pairs, screen, out):
        """The blocks of several ket shell pairs [(C, D), ...] back to back in `out` (their kernels run side by side on the
        device); returns `out` viewed as (sum of the blocks' rows, nao, nao)."""
        n2 = self.nao * self.nao
        nqs = [(2 * int(self.shells.l[C]) + 1) * (2 * int(self.shells.l[D]) + 1) for C, D in pairs]
        offs = np.concatenate([[0], np.cumsum(nqs)[:-1]]).astype(np.int64) * n2
        tot = int(sum(nqs))
        assert out.is_cuda and out.is_contiguous() and out.numel() >= tot * n2
        Cs = (ctypes.c_int * len(pairs))(*[int(C) for C, _ in pairs]); Ds = (ctypes.c_int * len(pairs))(*[int(D) for _, D in pairs])
        of = (ctypes.c_longlong * len(pairs))(*[int(x) for x in offs])
        if self.lib.DFT_EriColumnsMany(self._h, len(pairs), Cs, Ds, float(screen), ctypes.c_uint64(out.data_ptr()), of) != 0:
            raise RuntimeError("libdft: " + (self.lib.DFT_EriColumnsLastError(self._h) or b"").decode())
        return out.view(-1)[:tot * n2].view(tot, self.nao, self.nao)

