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

    def __init__(self, nao, local_sweep, device, group=None):
        self.nao = nao
        self.local_sweep = local_sweep
        self.device = device
        self.group = group
        self.buf = torch.zeros(nao * nao + 1, dtype=torch.float64, device=device)  # [Vxc | Exc]

    def compute_xc(self, dm):
        import torch.distributed as dist
        exc, vxc = self.local_sweep(dm)
        self.buf[: self.nao * self.nao].copy_(vxc.reshape(-1))
        self.buf[self.nao * self.nao] = exc
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
        n2 = self.nao * self.nao
        return ShardResult(float(self.buf[n2].item()), self.buf[:n2].reshape(self.nao, self.nao))
