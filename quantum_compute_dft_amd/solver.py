"""ctypes binding of libdft.so.

`DFTSolverWrapper` mirrors the reference class of the same name
(dft.py:15-95): same constructor arguments, same `compute_xc` /
`compute_coulomb` signatures and argument meaning, same error behaviour
(FileNotFoundError for a missing library, ValueError for an unknown functional,
RuntimeError when the C side returns a null solver).  Device arrays may be
torch tensors (`.data_ptr()`), CuPy arrays (`.data.ptr`, what the reference
passes) or plain integer device addresses.

Extensions beyond the reference class: compute_exchange / compute_jk /
eval_ao / set_option / timings, all thin calls into the extra C symbols.
"""
import ctypes
import os

from .build import LIB_PATH

_u64 = ctypes.c_uint64


def default_library_path():
    return LIB_PATH


def load_library(lib_path=None):
    """ctypes handle of libdft.so.  The process must end up with ONE HIP runtime: the array library
    that owns the device buffers (torch-ROCm here, CuPy-ROCm in the reference) bundles its own
    libamdhip64 and loads it by path, so it has to be imported BEFORE libdft.so, which then binds to
    that copy.  Loaded the other way round, libdft.so pulls in /opt/rocm's runtime, the process holds
    two, and whichever initialises second reports no device (measured on the MI355X box)."""
    try:
        import torch  # noqa: F401  (the device-buffer provider of this repo; brings the HIP runtime)
    except ImportError:
        pass
    return ctypes.CDLL(os.path.abspath(lib_path or default_library_path()))


def _ptr(a):
    """Raw device address of a torch tensor / CuPy array / int (0 for None)."""
    if a is None:
        return 0
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return int(a.data_ptr())
    data = getattr(a, "data", None)
    if data is not None and hasattr(data, "ptr"):
        return int(data.ptr)
    raise TypeError(f"cannot take a device pointer from {type(a).__name__}")


class DFTSolverWrapper:
    TYPE_LDA = 0
    TYPE_GGA = 1
    TYPE_B3LYP = 2

    def __init__(self, lib_path=None, functional_type="lda"):
        lib_path = lib_path or LIB_PATH
        if not os.path.exists(lib_path):
            raise FileNotFoundError(f"Shared library not found at: {lib_path}")
        self.lib = load_library(lib_path)
        self.functional_type = functional_type.upper()
        L = self.lib
        # --- the four reference symbols, declared exactly as dft.py:27-50 does
        L.DFT_CreateSolver.argtypes = [ctypes.c_int]
        L.DFT_CreateSolver.restype = ctypes.c_void_p
        L.DFT_DestroySolver.argtypes = [ctypes.c_void_p]
        L.DFT_DestroySolver.restype = None
        L.DFT_ComputeXC.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                    _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeXC.restype = ctypes.c_double
        L.DFT_ComputeCoulomb.argtypes = [ctypes.c_void_p, ctypes.c_int, _u64, _u64, _u64]
        L.DFT_ComputeCoulomb.restype = None
        # --- extensions
        L.DFT_ComputeXC64.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                      _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeXC64.restype = ctypes.c_double
        L.DFT_ComputeXCAsync.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                         _u64, _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeXCAsync.restype = ctypes.c_int
        L.DFT_ComputeXCOcc.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                       _u64, _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeXCOcc.restype = ctypes.c_double
        L.DFT_ComputeXCOccAsync.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                            _u64, _u64, _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeXCOccAsync.restype = ctypes.c_int
        L.DFT_ComputeExchange.argtypes = [ctypes.c_void_p, ctypes.c_int, _u64, _u64, _u64]
        L.DFT_ComputeExchange.restype = None
        L.DFT_ComputeJK.argtypes = [ctypes.c_void_p, ctypes.c_int, _u64, _u64, _u64, _u64]
        L.DFT_ComputeJK.restype = None
        L.DFT_ComputeJKRows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64, _u64, _u64, _u64]
        L.DFT_ComputeJKRows.restype = ctypes.c_int
        L.DFT_ComputeJKFactorized.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              _u64, _u64, _u64, _u64, _u64]
        L.DFT_ComputeJKFactorized.restype = ctypes.c_int
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        L.DFT_EvalAO.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                 dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, _u64, _u64, _u64]
        L.DFT_EvalAO.restype = ctypes.c_int
        L.DFT_ComputeXCDirect.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                          dp, ip, ip, ip, ip, dp, dp, ctypes.c_int, _u64, _u64, _u64, _u64, _u64, ctypes.c_longlong]
        L.DFT_ComputeXCDirect.restype = ctypes.c_int
        L.DFT_SetOption.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
        L.DFT_SetOption.restype = ctypes.c_int
        L.DFT_SetStream.argtypes = [ctypes.c_void_p, _u64]
        L.DFT_SetStream.restype = ctypes.c_int
        L.DFT_GetLastError.argtypes = [ctypes.c_void_p]
        L.DFT_GetLastError.restype = ctypes.c_char_p
        L.DFT_GetTimings.argtypes = [ctypes.c_void_p, dp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
        L.DFT_GetTimings.restype = ctypes.c_int
        L.DFT_GetVersion.argtypes = []
        L.DFT_GetVersion.restype = ctypes.c_int

        if self.functional_type == "LDA":
            c_type = self.TYPE_LDA
        elif self.functional_type == "GGA":
            c_type = self.TYPE_GGA
        elif self.functional_type == "B3LYP":
            c_type = self.TYPE_B3LYP
        else:
            raise ValueError(f"Unsupported functional type: {self.functional_type}")

        self.solver = L.DFT_CreateSolver(c_type)
        if not self.solver:
            raise RuntimeError("Failed to create C++ DFT Solver instance.")

    def __del__(self):
        if hasattr(self, "lib") and hasattr(self, "solver") and self.solver:
            self.lib.DFT_DestroySolver(self.solver)
            self.solver = None

    # ---- reference surface (dft.py:69-95) -------------------------------
    def compute_xc(self, ngrid, nao, d_dm, d_ao, d_weights, d_vxc, d_ao_grad=None):
        energy = self.lib.DFT_ComputeXC64(
            self.solver, int(ngrid), int(nao),
            _u64(_ptr(d_dm)), _u64(_ptr(d_ao)), _u64(_ptr(d_ao_grad)),
            _u64(_ptr(d_weights)), _u64(_ptr(d_vxc)))
        self._check()
        return energy

    def compute_coulomb(self, nao, d_eri, d_dm, d_J):
        self.lib.DFT_ComputeCoulomb(self.solver, int(nao), _u64(_ptr(d_eri)),
                                    _u64(_ptr(d_dm)), _u64(_ptr(d_J)))
        self._check()

    # ---- extensions -------------------------------------------------------
    def compute_xc_async(self, ngrid, nao, d_dm, d_ao, d_weights, d_vxc, d_exc, d_ao_grad=None):
        rc = self.lib.DFT_ComputeXCAsync(
            self.solver, int(ngrid), int(nao), _u64(_ptr(d_dm)), _u64(_ptr(d_ao)),
            _u64(_ptr(d_ao_grad)), _u64(_ptr(d_weights)), _u64(_ptr(d_vxc)), _u64(_ptr(d_exc)))
        self._check()
        return rc

    def compute_xc_occ(self, ngrid, nao, nocc, d_cocc, d_ao, d_weights, d_vxc, d_ao_grad=None, d_dm=None):
        """compute_xc with the occupied orbitals: d_cocc (nao, nocc), dm = cocc cocc^T (sqrt(2) C_occ for the closed-shell
        dm of dft.py:181-182).  Same Exc / Vxc as compute_xc(dm); the density step costs 4 nao nocc instead of 2 nao^2 flops
        per grid point where that is less.  d_dm is optional (used where the dm kernels are the better path)."""
        energy = self.lib.DFT_ComputeXCOcc(
            self.solver, int(ngrid), int(nao), int(nocc), _u64(_ptr(d_cocc)), _u64(_ptr(d_dm)), _u64(_ptr(d_ao)),
            _u64(_ptr(d_ao_grad)), _u64(_ptr(d_weights)), _u64(_ptr(d_vxc)))
        self._check()
        return energy

    def compute_xc_occ_async(self, ngrid, nao, nocc, d_cocc, d_ao, d_weights, d_vxc, d_exc, d_ao_grad=None, d_dm=None):
        rc = self.lib.DFT_ComputeXCOccAsync(
            self.solver, int(ngrid), int(nao), int(nocc), _u64(_ptr(d_cocc)), _u64(_ptr(d_dm)), _u64(_ptr(d_ao)),
            _u64(_ptr(d_ao_grad)), _u64(_ptr(d_weights)), _u64(_ptr(d_vxc)), _u64(_ptr(d_exc)))
        self._check()
        return rc

    def compute_exchange(self, nao, d_eri, d_dm, d_K):
        self.lib.DFT_ComputeExchange(self.solver, int(nao), _u64(_ptr(d_eri)),
                                     _u64(_ptr(d_dm)), _u64(_ptr(d_K)))
        self._check()

    def compute_jk(self, nao, d_eri, d_dm, d_J, d_K):
        self.lib.DFT_ComputeJK(self.solver, int(nao), _u64(_ptr(d_eri)), _u64(_ptr(d_dm)),
                               _u64(_ptr(d_J)), _u64(_ptr(d_K)))
        self._check()

    def compute_jk_rows(self, nao, i_lo, i_hi, d_eri_rows, d_dm, d_J, d_K):
        """J, K from ERI rows (i, j), i_lo <= i < i_hi; d_eri_rows is that row block (first row (i_lo, 0)).
        Partial J over all columns, rows [i_lo, i_hi) of K: the shares of all blocks add up (all-reduce)."""
        rc = self.lib.DFT_ComputeJKRows(self.solver, int(nao), int(i_lo), int(i_hi), _u64(_ptr(d_eri_rows)),
                                        _u64(_ptr(d_dm)), _u64(_ptr(d_J)), _u64(_ptr(d_K)))
        self._check()
        return rc

    def compute_jk_factorized(self, nao, naux, nocc, d_chol, d_dm, d_cocc, d_J, d_K):
        """J, K from Cholesky vectors d_chol (naux, nao, nao); dm = cocc cocc^T, cocc (nao, nocc).
        d_J / d_K (and the input only the other one needs) may be None."""
        rc = self.lib.DFT_ComputeJKFactorized(self.solver, int(nao), int(naux), int(nocc), _u64(_ptr(d_chol)),
                                              _u64(_ptr(d_dm)), _u64(_ptr(d_cocc)), _u64(_ptr(d_J)), _u64(_ptr(d_K)))
        self._check()
        return rc

    @staticmethod
    def _shell_args(shells):
        """(keep-alive arrays, ctypes arguments) of a basis.ShellTable for DFT_EvalAO / DFT_ComputeXCDirect."""
        import numpy as np
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        arrs = [np.ascontiguousarray(shells.xyz, dtype=np.float64)] + \
               [np.ascontiguousarray(a, dtype=np.int32) for a in (shells.l, shells.nprim, shells.off, shells.ao)] + \
               [np.ascontiguousarray(shells.exp, dtype=np.float64), np.ascontiguousarray(shells.coef, dtype=np.float64)]
        args = [int(shells.nao), int(len(arrs[1])), arrs[0].ctypes.data_as(dp)] + [a.ctypes.data_as(ip) for a in arrs[1:5]] + \
               [arrs[5].ctypes.data_as(dp), arrs[6].ctypes.data_as(dp), int(len(arrs[5]))]
        return arrs, args

    def eval_ao(self, shells, d_coords, ngrid, d_ao, d_ao_grad=None):
        """shells: a basis.ShellTable (host numpy arrays, see basis.py)."""
        keep, args = self._shell_args(shells)
        rc = self.lib.DFT_EvalAO(self.solver, int(ngrid), *args, _u64(_ptr(d_coords)), _u64(_ptr(d_ao)), _u64(_ptr(d_ao_grad)))
        self._check()
        return rc

    def compute_xc_direct(self, shells, ngrid, d_coords, d_weights, d_dm, d_vxc, d_exc, chunk_points=0):
        """AO -> rho -> XC -> Vxc without resident AO planes (DFT_ComputeXCDirect): results in d_vxc / d_exc once the
        solver's stream has drained (asynchronous, like compute_xc_async)."""
        keep, args = self._shell_args(shells)
        rc = self.lib.DFT_ComputeXCDirect(self.solver, int(ngrid), *args, _u64(_ptr(d_coords)), _u64(_ptr(d_weights)),
                                          _u64(_ptr(d_dm)), _u64(_ptr(d_vxc)), _u64(_ptr(d_exc)), int(chunk_points))
        self._check()
        return rc

    def set_option(self, key, value):
        if self.lib.DFT_SetOption(self.solver, key.encode(), float(value)) != 0:
            raise KeyError(key)

    def set_stream(self, hip_stream):
        self.lib.DFT_SetStream(self.solver, _u64(int(hip_stream)))

    def last_error(self):
        return (self.lib.DFT_GetLastError(self.solver) or b"").decode()

    def timings(self, max_entries=16):
        ms = (ctypes.c_double * max_entries)()
        names = (ctypes.c_char_p * max_entries)()
        n = self.lib.DFT_GetTimings(self.solver, ms, names, max_entries)
        return [(names[i].decode(), ms[i]) for i in range(n)]

    def _check(self):
        err = self.lib.DFT_GetLastError(self.solver)
        if err:
            raise RuntimeError("libdft: " + err.decode())
