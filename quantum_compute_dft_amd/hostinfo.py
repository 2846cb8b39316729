"""Host-side resources of the box a rank runs on."""
import contextlib
import os


def host_cpu_share(cap=16, whole_node=False):
    """CPUs this process may really use: affinity mask, cgroup quota, capped (16 = the GPU box's
    share for one GPU); os.cpu_count() reports the whole host (256 there).  `whole_node`: the allowance of ALL
    the node's ranks together (cap x LOCAL_WORLD_SIZE) -- for a phase in which one rank works and the others wait
    (the Cholesky factorisation of the ERI, inputs.build)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    # ranks of one node (torch.distributed.run exports LOCAL_WORLD_SIZE) share that allowance: without the division
    # N ranks start N x share threads on the same cores (integrals, LAPACK) and slow each other down
    try:
        lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        lws = 1
    if whole_node:
        return max(1, min(n, cap * lws))
    return max(1, min(max(1, n // lws), cap))


def blas_threads(n=None):
    """Context manager pinning the BLAS/LAPACK/OpenMP pools behind numpy/scipy to the CPU share.  With
    one thread per visible core (256 on the GPU box) the workers of one BLAS call keep spinning after
    it; on a 16-core cgroup share that exhausts the CPU quota and the whole process is throttled for
    tens of ms a little later -- seen as an SCF cycle whose 0.7 ms `eigh` takes 25-90 ms every few
    calls.  Entry points (bench.py, dft.py) enter this before their first numpy call."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return contextlib.nullcontext()
    return threadpool_limits(limits=n or host_cpu_share())
