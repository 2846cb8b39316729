"""Grid-point sharding of the XC sweep across the GPUs of one node.

The reference is single-GPU (no collective anywhere in src/dft_solver.cu or dft.py);
this is the multi-GPU form of the same call: grid points are independent, so every
rank keeps a contiguous block of (ao, ao_grad, weights) resident in its HBM, the
density matrix is replicated, and one step is

    local DFT_ComputeXC on the shard  ->  ONE all-reduce(sum) of [Vxc (nao^2) | Exc (1)]

over RCCL (torch.distributed backend "nccl" on ROCm; xGMI inside the node).  The
payload is nao^2+1 doubles (0.10 MB at nao 114, 10.6 MB at nao 1150): latency-bound,
so it is a single flat buffer, never bucketed.  One process per GPU.
"""
from dataclasses import dataclass

import torch


def shard_bounds(ngrid, world_size, rank, align=16):
    """Contiguous [lo, hi) block of grid points for `rank`.  Blocks are equal up to
    `align` (the kernels' 16-point sub-tile) and cover [0, ngrid) exactly."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    per = -(-ngrid // world_size)
    per = -(-per // align) * align
    lo = min(ngrid, rank * per)
    hi = min(ngrid, lo + per)
    return lo, hi


@dataclass
class ShardResult:
    exc: float
    vxc: torch.Tensor  # (nao, nao), identical on every rank


class ShardedXC:
    """One rank's part of the sharded sweep.

    local_sweep(dm) must return (exc_local: float or 0-dim tensor, vxc_local: (nao, nao)
    tensor on `device`) for THIS rank's grid block.  On a GPU box it is a closure over
    DFTSolverWrapper.compute_xc(_async) and the rank's resident shard (see bench.py);
    tests inject a CPU closure to exercise the partition + collective under gloo.
    """

    def __init__(self, nao, local_sweep, device, group=None, collect_always=False):
        self.nao = nao
        self.local_sweep = local_sweep
        self.device = device
        self.group = group
        self.collect_always = collect_always   # run the collective in a group of ONE rank too (the RCCL smoke test on a one-GPU box)
        self.buf = torch.zeros(nao * nao + 1, dtype=torch.float64, device=device)  # [Vxc | Exc]

    def compute_xc(self, dm):
        import torch.distributed as dist
        exc, vxc = self.local_sweep(dm)
        self.buf[: self.nao * self.nao].copy_(vxc.reshape(-1))
        self.buf[self.nao * self.nao] = exc
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.collect_always):
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        n2 = self.nao * self.nao
        return ShardResult(float(self.buf[n2].item()), self.buf[:n2].reshape(self.nao, self.nao))


def vector_bounds(naux, world_size, rank):
    """Contiguous [lo, hi) slice of the Cholesky vectors for `rank` (factorised J/K, cd_kernels.hpp):
    J = sum_P (L_P:D) L_P and K = sum_P L_P D L_P are sums over vectors, so every rank keeps
    naux/world vectors resident and contributes a partial J and K."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    per = -(-naux // world_size)
    lo = min(naux, rank * per)
    return lo, min(naux, lo + per)


def eri_row_bounds(nao, world_size, rank):
    """Row block [i_lo*nao, i_hi*nao) of the dense (nao^2, nao^2) ERI for `rank`: whole first indices i
    (so a rank's K rows are complete), contiguous, covering all rows (ERI rows (ij) sharded
    over the GPUs; the reference keeps the whole matrix on one, dft.py:166)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    per = -(-nao // world_size)
    i_lo = min(nao, rank * per)
    i_hi = min(nao, i_lo + per)
    return i_lo * nao, i_hi * nao


def scatter_vectors(L, nao, device, world_size, rank, group=None):
    """Rank 0 holds the Cholesky vectors L (naux, nao, nao) (torch tensor on any device, or numpy); every rank receives
    its slice vector_bounds(naux, world, rank) as a tensor on `device` -- the factorisation runs ONCE per node (rank 0,
    on the whole node's CPU allowance) instead of once per rank on 1/N of the cores.  Returns (slice, naux).
    Point-to-point sends of the slices (no rank ever holds more than rank 0 already does); under gloo the payload goes
    through host memory."""
    import numpy as np
    import torch.distributed as dist
    dev = torch.device(device)
    meta = [int(L.shape[0]) if rank == 0 else None]
    dist.broadcast_object_list(meta, src=0, group=group)
    naux = meta[0]
    host = dist.get_backend(group) == "gloo"
    if rank == 0:
        Lt = L if torch.is_tensor(L) else torch.from_numpy(np.ascontiguousarray(L))
        for r in range(1, world_size):
            lo, hi = vector_bounds(naux, world_size, r)
            if hi > lo:
                part = Lt[lo:hi].contiguous()
                dist.send(part.cpu() if host else part.to(dev), dst=r, group=group)
        lo, hi = vector_bounds(naux, world_size, 0)
        return Lt[lo:hi].to(dev).contiguous(), naux
    lo, hi = vector_bounds(naux, world_size, rank)
    buf = torch.empty((hi - lo, nao, nao), dtype=torch.float64, device="cpu" if host else dev)
    if hi > lo:
        dist.recv(buf, src=0, group=group)
    return buf.to(dev), naux


class ReplicaSync:
    """Rank 0 is authoritative for the small replicated state of the SCF loop (dm, cocc, the convergence
    scalars): it alone runs DIIS + eigh and every other rank receives the result in ONE broadcast per cycle.
    Replicas therefore cannot drift (different host LAPACK thread counts, different GPUs behind hipSOLVER),
    and since the stop decision is taken from the broadcast scalars every rank leaves the loop in the same
    cycle -- no rank is left blocking in the next all-reduce."""

    def __init__(self, device, group=None, collect_always=False):
        self.device, self.group, self.collect_always = torch.device(device), group, collect_always

    def _bcast(self, flat):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.group) == 1 and not self.collect_always):
            return flat
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":   # rehearsal on one card: gloo moves host memory
            h = flat.cpu(); dist.broadcast(h, 0, group=self.group); flat.copy_(h)
        else:
            dist.broadcast(flat, 0, group=self.group)
        return flat

    def broadcast(self, tensors):
        """In place, from rank 0, as one flat message."""
        flat = self._bcast(torch.cat([t.reshape(-1) for t in tensors]))
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t)); o += t.numel()

    def broadcast_numpy(self, arrays):
        """The same for C-contiguous float64 numpy arrays (host loop): through device memory under RCCL,
        straight from host memory under gloo."""
        import torch.distributed as dist
        ts = [torch.from_numpy(a) for a in arrays]          # views: the arrays are updated in place
        if dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) != "gloo" and self.device.type == "cuda":
            dev = [t.to(self.device) for t in ts]
            self.broadcast(dev)
            for t, d in zip(ts, dev):
                t.copy_(d)
        else:
            self.broadcast(ts)


@dataclass
class FockParts:
    exc: float
    vxc: torch.Tensor  # (nao, nao) each, identical on every rank
    J: torch.Tensor
    K: torch.Tensor


class ShardedFock:
    """The whole device side of one SCF cycle on N GPUs: grid block -> partial Vxc, Exc;
    vector slice -> partial J, K; then ONE all-reduce(sum) of the flat [Vxc | J | K | Exc]
    (3 nao^2 + 1 doubles; BASELINE config 5, nao 1150: 31.7 MB).  `local_sweep(dm)` as in
    ShardedXC; `local_jk(dm, cocc)` returns this rank's (J_partial, K_partial or None)."""

    def __init__(self, nao, local_sweep, local_jk, device, group=None, collect_always=False):
        self.nao, self.local_sweep, self.local_jk, self.group = nao, local_sweep, local_jk, group
        self.collect_always = collect_always
        self.buf = torch.zeros(3 * nao * nao + 1, dtype=torch.float64, device=device)

    def compute(self, dm, cocc=None):
        import torch.distributed as dist
        n2 = self.nao * self.nao
        exc, vxc = self.local_sweep(dm)
        J, K = self.local_jk(dm, cocc)
        self.buf[:n2].copy_(vxc.reshape(-1))
        self.buf[n2:2 * n2].copy_(J.reshape(-1))
        if K is not None:
            self.buf[2 * n2:3 * n2].copy_(K.reshape(-1))
        else:
            self.buf[2 * n2:3 * n2].zero_()
        self.buf[3 * n2] = exc
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.collect_always):
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        m = lambda k: self.buf[k * n2:(k + 1) * n2].reshape(self.nao, self.nao)
        return FockParts(float(self.buf[3 * n2].item()), m(0), m(1), m(2))
