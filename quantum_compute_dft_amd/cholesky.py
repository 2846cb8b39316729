"""Pivoted (incomplete) Cholesky factorisation of the electron-repulsion tensor,
(ij|kl) ~= sum_P L[P,i,j] L[P,k,l], integral-direct: only the diagonal (ij|ij) and the columns of
the pivots' shell pairs are ever computed, so basis sets whose dense ERI (8 nao^4 bytes, what the
reference uploads at dft.py:166-168) cannot be stored still get J and K.

Residual bound: every element of (ij|kl) - sum_P L L is below `tol` in magnitude on exit (the
residual matrix stays positive semi-definite, so its largest element is on the diagonal).

Shell-pair blocked: when a pivot is chosen, the columns of its whole shell pair are computed once
and every index of that block whose residual diagonal is still above max(tol, span * pivot) is
decomposed before the next integral batch (the span factor of Koch / Aquilante et al.)."""
import numpy as np

from .integrals import EriColumns


def cholesky_eri(shells, tol=1e-8, span=0.01, max_vectors=None, screen=None, verbose=False):
    """Returns L of shape (naux, nao, nao), float64, every L[P] symmetric."""
    from .hostinfo import blas_threads
    with blas_threads():
        return _cholesky_eri(shells, tol, span, max_vectors, screen, verbose)


def _cholesky_eri(shells, tol, span, max_vectors, screen, verbose):
    n = shells.nao
    n2 = n * n
    eri = EriColumns(shells)
    try:
        diag = eri.diag().reshape(n2).copy()
        shell_of = np.empty(n, dtype=np.int64)
        for s in range(shells.nshell):
            shell_of[shells.ao[s]:shells.ao[s] + 2 * shells.l[s] + 1] = s
        cap = max_vectors or min(n2, 16 * n)
        L = np.empty((cap, n2))
        k = 0
        screen = tol * 1e-4 if screen is None else screen
        while k < cap:
            p = int(np.argmax(diag))
            dmax = diag[p]
            if dmax < tol:
                break
            C, D = int(shell_of[p // n]), int(shell_of[p % n])
            c0, d0 = int(shells.ao[C]), int(shells.ao[D])
            nc, nd = 2 * int(shells.l[C]) + 1, 2 * int(shells.l[D]) + 1
            qidx = ((c0 + np.arange(nc))[:, None] * n + (d0 + np.arange(nd))[None, :]).reshape(-1)
            res = eri.cols(C, D, screen).reshape(nc * nd, n2)
            if k:
                res -= L[:k, qidx].T @ L[:k]
            floor = max(tol, span * dmax)
            while k < cap:
                dq = diag[qidx]
                b = int(np.argmax(dq))
                if dq[b] < floor:
                    break
                v = res[b] / np.sqrt(dq[b])
                L[k] = v
                k += 1
                diag -= v * v
                diag[qidx[b]] = 0.0
                res -= np.outer(v[qidx], v)
            np.maximum(diag, 0.0, out=diag)
            if verbose:
                print(f"cholesky: {k} vectors, residual {diag.max():.3e}", flush=True)
        return np.ascontiguousarray(L[:k]).reshape(k, n, n)
    finally:
        eri.close()
