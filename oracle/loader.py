"""ctypes loader for the plain-C oracle (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import fcntl
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

POINTWISE_KINDS = {
    "slater": 0, "vwn5": 1, "vwn_rpa": 2, "pw92": 3, "pbe_x": 4, "pbe_c": 5,
    "b88": 6, "lyp": 7, "lda": 8, "gga": 9, "b3lyp": 10,
}


_SRCS = ("xc_oracle.c", "ao_oracle.c", "Makefile")


def _src_hash():
    h = hashlib.sha256()
    for f in _SRCS:
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _current(path):
    try:
        return os.path.exists(path) and open(path + ".srchash").read().strip() == _src_hash()
    except OSError:
        return False


def build(omp=False):
    """Compile the oracle with gcc.  Called by __graft_entry__.build() and tests/conftest.py only:
    lib() below never compiles (bench.py times a prebuilt checker, it does not start a compiler)."""
    target = "_build/liboracle_omp.so" if omp else "_build/liboracle.so"
    path = os.path.join(_HERE, target)
    if _current(path):
        return path
    os.makedirs(os.path.join(_HERE, "_build"), exist_ok=True)
    with open(path + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not _current(path):
            env = {k: v for k, v in os.environ.items()
                   if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB") and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER"))}
            subprocess.run(["make", "-s", "-B", "-C", _HERE, target], check=True, env=env)
            with open(path + ".srchash", "w") as fh:
                fh.write(_src_hash() + "\n")
    return path


def lib(omp=False):
    key = bool(omp)
    if key not in _LIBS:
        path = os.path.join(_HERE, "_build", "liboracle_omp.so" if omp else "liboracle.so")
        if not _current(path):
            raise RuntimeError(f"{path} is missing or older than oracle/*.c: run `python __graft_entry__.py` "
                               "(tests build it in conftest.py)")
        L = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        L.orc_compute_xc.restype = ctypes.c_double
        L.orc_compute_xc.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_int, dp, dp, dp, dp,
                                     dp, ctypes.c_int, dp, dp]
        L.orc_coulomb.restype = None
        L.orc_coulomb.argtypes = [ctypes.c_int, dp, dp, dp]
        L.orc_exchange.restype = None
        L.orc_exchange.argtypes = [ctypes.c_int, dp, dp, dp]
        L.orc_pointwise.restype = None
        L.orc_pointwise.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, dp, dp, dp]
        ip = ctypes.POINTER(ctypes.c_int)
        L.orc_eval_ao.restype = ctypes.c_int
        L.orc_eval_ao.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, dp, ip, ip, ip, ip,
                                  dp, dp, dp, dp, dp]
        _LIBS[key] = L
    return _LIBS[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def compute_xc(xc_type, dm, ao, weights, ao_grad=None, quirks=True, want_density=False, omp=False):
    """Oracle of DFT_ComputeXC.  xc_type 0/1/2 (LDA/GGA/B3LYP).
    Returns (exc, vxc_raw[, rho, grad_rho])."""
    ao = _c(ao); dm = _c(dm); w = _c(weights); gr = _c(ao_grad)
    ngrid, nao = ao.shape
    assert dm.shape == (nao, nao) and w.shape == (ngrid,)
    if xc_type != 0:
        assert gr is not None and gr.shape == (3, ngrid, nao)
    vxc = np.zeros((nao, nao))
    rho = np.zeros(ngrid) if want_density else None
    grad = np.zeros((ngrid, 3)) if want_density else None
    exc = lib(omp).orc_compute_xc(int(xc_type), ngrid, nao, _p(dm), _p(ao), _p(gr), _p(w),
                                  _p(vxc), 1 if quirks else 0, _p(rho), _p(grad))
    if want_density:
        return exc, vxc, rho, grad
    return exc, vxc


def coulomb(eri, dm):
    dm = _c(dm); nao = dm.shape[0]
    eri = _c(eri).reshape(nao * nao, nao * nao)
    J = np.zeros((nao, nao))
    lib().orc_coulomb(nao, _p(eri), _p(dm), _p(J))
    return J


def exchange(eri, dm):
    dm = _c(dm); nao = dm.shape[0]
    eri = _c(eri).reshape(nao * nao, nao * nao)
    K = np.zeros((nao, nao))
    lib().orc_exchange(nao, _p(eri), _p(dm), _p(K))
    return K


def jk_from_factors(chol, dm):
    """J and K of dft.py:203,218 with the ERI replaced by its factorisation sum_P L_P (x) L_P:
    J = sum_P (L_P : dm) L_P,  K = sum_P L_P dm L_P  -- straight from dm (no occupied orbitals),
    so it checks the device's Cocc route independently.  numpy, small cases only."""
    chol, dm = _c(chol), _c(dm)
    J = np.einsum("pij,p->ij", chol, np.einsum("pij,ij->p", chol, dm))
    K = np.einsum("pik,kl,pjl->ij", chol, dm, chol, optimize=True)
    return J, K


def pointwise(kind, rho, sigma=None, quirks=True):
    """(n,3) array of (e, vrho, vsigma) for one functional kind (name or id)."""
    k = POINTWISE_KINDS[kind] if isinstance(kind, str) else int(kind)
    rho = _c(np.atleast_1d(rho))
    sigma = _c(np.atleast_1d(sigma)) if sigma is not None else np.zeros_like(rho)
    out = np.zeros((rho.size, 3))
    lib().orc_pointwise(k, 1 if quirks else 0, rho.size, _p(rho), _p(sigma), _p(out))
    return out


def eval_ao(shells, coords, deriv=0):
    """Oracle of DFT_EvalAO.  `shells` needs attributes xyz (nshell,3), l, nprim, off, ao
    (int arrays), exp, coef (normalised), nao.  Returns ao (ngrid,nao)[, grad (3,ngrid,nao)]."""
    coords = _c(coords)
    ngrid = coords.shape[0]
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    ipp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    xyz = _c(shells.xyz); ls = i32(shells.l); npr = i32(shells.nprim); off = i32(shells.off)
    aoc = i32(shells.ao); ex = _c(shells.exp); cf = _c(shells.coef)
    ao = np.zeros((ngrid, shells.nao))
    grad = np.zeros((3, ngrid, shells.nao)) if deriv else None
    rc = lib().orc_eval_ao(ngrid, shells.nao, len(ls), _p(xyz), ipp(ls), ipp(npr), ipp(off),
                           ipp(aoc), _p(ex), _p(cf), _p(coords), _p(ao), _p(grad))
    if rc != 0:
        raise ValueError("oracle eval_ao: unsupported angular momentum")
    return (ao, grad) if deriv else ao
