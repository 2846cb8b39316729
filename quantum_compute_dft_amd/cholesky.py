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


def cholesky_eri(shells, tol=1e-8, span=0.01, max_vectors=None, screen=None, verbose=False, device=None, device_columns=True):
    """Returns L of shape (naux, nao, nao), float64, every L[P] symmetric: a numpy array, or -- with `device` a
    CUDA/HIP torch device -- a tensor that stays on that device (the vectors are consumed there by
    DFT_ComputeJKFactorized; 9.5 GB at Anthracene/def2-TZVP never cross PCIe).

    With a device the linear algebra of the factorisation (the residual update res -= L_q^T L, 116 GFLOP per f-f
    shell pair at nao 494, and the rank-1 updates) runs there through torch; the integral columns still come from
    the host engine (integrals.c) through a pinned staging buffer.  On the host that algebra was 70 % of the
    factorisation time (Anthracene/def2-SVP on 8 cores: integrals 14.7 s, update GEMMs 20.0 s, rank-1 loop 13.2 s)."""
    from .hostinfo import blas_threads
    with blas_threads():
        if device is not None and str(device).startswith("cuda"):
            return _cholesky_eri_device(shells, tol, span, max_vectors, screen, verbose, device, device_columns)
        return _cholesky_eri(shells, tol, span, max_vectors, screen, verbose)


def _cholesky_eri_device(shells, tol, span, max_vectors, screen, verbose, device, device_columns=True):
    """`device_columns`: the pivots' integral columns come from the device kernel (csrc/eri_cols.hip) -- only the
    diagonal (ij|ij) and its Schwarz bounds are still the host engine's; False = host columns through pinned memory
    (round 2's path: 8.7 of the 11.5 s of an Anthracene/def2-TZVP factorisation)."""
    import torch
    from .integrals import DeviceEriColumns, schwarz_bounds
    dev = torch.device(device)
    n = shells.nao
    n2 = n * n
    eri = EriColumns(shells)
    dcols = None
    try:
        diag_h = eri.diag()
        if device_columns:
            dcols = DeviceEriColumns(shells, schwarz_bounds(shells, diag_h))
        diag = torch.as_tensor(diag_h.reshape(n2).copy(), device=dev)
        shell_of = np.empty(n, dtype=np.int64)
        for s in range(shells.nshell):
            shell_of[shells.ao[s]:shells.ao[s] + 2 * shells.l[s] + 1] = s
        cap_max = max_vectors or min(n2, 16 * n)
        cap = min(cap_max, 4 * n)
        L = torch.empty((cap, n2), dtype=torch.float64, device=dev)
        maxq = (2 * int(np.max(shells.l)) + 1) ** 2
        if dcols is None:
            stage = torch.empty((maxq, n2), dtype=torch.float64).pin_memory()    # the host engine writes the columns here
            stage_np = stage.numpy()
        k = 0
        screen = tol * 1e-4 if screen is None else screen
        # Several shell-pair blocks per step with the device columns: the blocks whose largest residual diagonal is within
        # `batch_frac` of the step's pivot (at most `batch`) are computed side by side (DFT_EriColumnsMany), take ONE residual
        # GEMM and ONE small pivoted factorisation on the host together -- the per-step host work (a sync, the pivots of the
        # block, the index tensors: 3 of the 5 s of an Anthracene/def2-TZVP factorisation, one block per step) is shared by
        # them.  Still a pivoted Cholesky with the same residual bound; the pivot ORDER differs from the one-block sequence.
        import os
        batch = int(os.environ.get("QCDFT_CHOL_BATCH", "8")) if dcols is not None else 1      # (the environment: tools/chol_dev_time.py's sweeps)
        batch_frac = float(os.environ.get("QCDFT_CHOL_FRAC", "0.1"))
        nsh = shells.nshell
        so = torch.as_tensor(shell_of, device=dev)
        hi, lo = torch.maximum(so[:, None], so[None, :]), torch.minimum(so[:, None], so[None, :])
        blk = (hi * nsh + lo).reshape(-1)                              # shell-pair block of every (i, j), (C, D) and (D, C) as one
        if dcols is not None:
            dstage = torch.empty((batch * maxq, n2), dtype=torch.float64, device=dev)
        while k < cap_max:
            if batch > 1:
                pm = torch.zeros(nsh * nsh, dtype=torch.float64, device=dev).scatter_reduce(0, blk, diag, "amax", include_self=True)
                vals, ids = torch.topk(pm, min(batch, pm.numel()))
                vals, ids = vals.cpu().numpy(), ids.cpu().numpy()
                dmax = float(vals[0])
                if dmax < tol:
                    break
                pairs = [(int(b) // nsh, int(b) % nsh) for v, b in zip(vals, ids) if v >= max(tol, batch_frac * dmax)]
            else:
                dmax, p = torch.max(diag, dim=0)
                dmax, p = float(dmax), int(p)
                if dmax < tol:
                    break
                pairs = [(int(shell_of[p // n]), int(shell_of[p % n]))]
            qs = []
            for C, D in pairs:
                c0, d0 = int(shells.ao[C]), int(shells.ao[D])
                nc, nd = 2 * int(shells.l[C]) + 1, 2 * int(shells.l[D]) + 1
                qs.append(((c0 + np.arange(nc))[:, None] * n + (d0 + np.arange(nd))[None, :]).reshape(-1))
            qidx_h = np.concatenate(qs)
            nq = len(qidx_h)
            qidx = torch.as_tensor(qidx_h, device=dev)
            if dcols is None:
                C, D = pairs[0]
                eri.cols(C, D, screen, out=stage_np[:nq], lower_only=True)     # the host writes i >= j only (integrals.c)
                low = stage[:nq].to(dev, non_blocking=True).view(nq, n, n)
            else:
                low = dcols.cols_many(pairs, screen, dstage)                   # the same, computed in HBM
            res = (torch.tril(low) + torch.tril(low, -1).transpose(1, 2)).reshape(nq, n2)   # ... and the device mirrors
            if k:
                res -= L[:k, qidx].T @ L[:k]
            floor = max(tol, span * dmax)
            # The step's vectors in ONE go.  The sequential loop (pick the largest residual diagonal of the block, divide
            # its column, subtract the rank-1 term from the block's columns, repeat) is a pivoted Cholesky of the block's own
            # nq x nq residual A = res[:, qidx]: that small matrix goes to the host once, the pivots B and the triangular
            # factor G with A[B, B] = G G^T come from it, and the vectors are V = G^-1 res[B].
            A = res[:, qidx].cpu().numpy()
            B, G = _block_pivots(A, floor, cap_max - k)
            r = len(B)
            if r:
                if k + r > cap:                                   # grow the vector store
                    cap = min(cap_max, max(2 * cap, k + r))
                    L = torch.cat([L, torch.empty((cap - L.shape[0], n2), dtype=torch.float64, device=dev)])
                Bt = torch.as_tensor(np.asarray(B), device=dev)
                # V = G^-1 res[B]: the small triangular inverse on the host (r <= a few hundred), one GEMM on the device
                # (rocBLAS's trsm wants a workspace it could not get for r x nao^2 right-hand sides above r ~ 100)
                V = torch.as_tensor(_lower_inverse(G), device=dev) @ res[Bt]
                L[k:k + r] = V
                k += r
                diag -= (V * V).sum(0)
                diag[qidx[Bt]] = 0.0
                diag[(qidx[Bt] % n) * n + qidx[Bt] // n] = 0.0     # the mirrored index (j, i) of every pivot (i, j)
            diag.clamp_(min=0.0)
            if verbose:
                print(f"cholesky: {k} vectors, residual {float(diag.max()):.3e}", flush=True)
        return L[:k].clone().reshape(k, n, n)
    finally:
        eri.close()
        if dcols is not None:
            dcols.close()


def _lower_inverse(G):
    """Inverse of a lower-triangular matrix by forward substitution, row by row (plain numpy: scipy's solve_triangular
    crashed in its BLAS under OMP_NUM_THREADS=1, the setting torch.distributed.run gives every rank)."""
    r = G.shape[0]
    X = np.zeros((r, r))
    for i in range(r):
        X[i, :i] = -(G[i, :i] @ X[:i, :i]) / G[i, i]
        X[i, i] = 1.0 / G[i, i]
    return X


def _block_pivots(A, floor, room):
    """Pivoted Cholesky of the symmetric PSD block A (nq x nq) down to `floor` on its residual diagonal, at most `room`
    pivots: (pivot indices B in the order taken, lower-triangular G with A[B][:, B] = G G^T).  The same pivots, in the
    same order, as the vector-by-vector loop of the host factorisation takes inside a shell-pair block."""
    A = np.array(A, dtype=np.float64)
    nq = A.shape[0]
    d = np.diag(A).copy()
    W = np.zeros((0, nq))                    # rows: the block's part of the vectors found so far
    B = []
    while len(B) < min(nq, room):
        b = int(np.argmax(d))
        if d[b] < floor:
            break
        w = (A[b] - (W[:, b] @ W if len(B) else 0.0)) / np.sqrt(d[b])
        W = np.vstack([W, w])
        B.append(b)
        d -= w * w
        d[b] = 0.0
    G = W[:, B].T if B else np.zeros((0, 0))  # G[j, i] = (vector i)[pivot j]: lower-triangular by construction
    return B, np.tril(G)


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
