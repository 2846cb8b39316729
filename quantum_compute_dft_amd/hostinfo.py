"""Host-side resources of the box a rank runs on."""
import contextlib
import os


def host_cpu_share(cap=16):
    """CPUs this process may really use: affinity mask, cgroup quota, capped (16 = the GPU box's
    share for one GPU); os.cpu_count() reports the whole host (256 there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def blas_threads(n=None):
    """Context manager pinning the BLAS/LAPACK pools behind numpy/scipy to the CPU share: with one
    thread per visible core (256) the nao x nao eigh of an SCF cycle stalls for ~90 ms every few
    calls on a 16-core share (measured, Benzene/def2-SVP: 0.95 ms pinned)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return contextlib.nullcontext()
    return threadpool_limits(limits=n or host_cpu_share())
