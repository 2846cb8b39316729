"""Builds libdft.so (the C-ABI shared library) in-tree with hipcc for gfx950.

Two entry points with different contracts:

* `build_library()`  -- compiles when the library is missing or older than its sources.  Only
  `__graft_entry__.build()`, `python -m quantum_compute_dft_amd.build` and the test fixtures call it.
* `library_path()`   -- NEVER compiles: returns the path of an up-to-date library or raises.  Every
  product entry point (bench.py, dft.py, tools/, smoke()) uses this one, so a bench or profiler run
  cannot start a compiler from a GPU-initialised (or profiler-preloaded) process, and N ranks cannot
  race on one libdft.so.

"Up to date" is decided by a content hash of the sources stored next to the library (mtimes do not
survive the snapshot copy to the GPU box).
"""
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdft.so")
STAMP_PATH = LIB_PATH + ".srchash"
SOURCES = ["dft_api.hip", "xc_occ.hip"]   # one object each, compiled in parallel, linked into libdft.so
HEADERS = ["xc_functionals.hpp", "xc_kernels.hpp", "xc_ws_kernels.hpp", "xc_ws16_kernels.hpp", "xc_big_kernels.hpp",
           "xc_occ_kernels.hpp", "xc_occ_launch.hpp", "jk_kernels.hpp", "ao_kernels.hpp", "cd_kernels.hpp", "device_util.hpp",
           os.path.join("..", "..", "include", "dft_solver.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950's register file is unified);
# without it hipcc 7.2 can wrap every MFMA group of a loop in v_accvgpr_write/read copy storms
# (measured on the fp64 probe: 35 -> 75 TFLOP/s, profiles/r01_mfma_f64_probe2.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]


def source_hash(files=None, flags=None):
    h = hashlib.sha256(" ".join(flags if flags is not None else FLAGS).encode())
    for f in (files if files is not None else [os.path.join(CSRC, f) for f in SOURCES + HEADERS]):
        if os.path.exists(f):
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _stamp_ok(lib, stamp, want):
    try:
        return os.path.exists(lib) and open(stamp).read().strip() == want
    except OSError:
        return False


def compile_env():
    """Environment for compiler children: no profiler / tool preloads (a child of a process running
    under rocprofv3 would otherwise initialise the GPU inside make/gcc/hipcc)."""
    env = dict(os.environ)
    for k in list(env):
        if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") or k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "RPD_")):
            env.pop(k)
    return env


@contextlib.contextmanager
def build_lock(path):
    """Exclusive file lock: concurrent builders (ranks, pytest-xdist workers) serialise, the late
    ones find the stamp current and skip."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path + ".lock", "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def library_path():
    """Path of an up-to-date libdft.so; raises instead of compiling (see module docstring)."""
    if not _stamp_ok(LIB_PATH, STAMP_PATH, source_hash()):
        raise RuntimeError(f"{LIB_PATH} is missing or older than csrc/: run `python __graft_entry__.py` "
                           "(or `python -m quantum_compute_dft_amd.build`) first; product entry points never compile")
    return LIB_PATH


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> quantum_compute_dft_amd/lib/libdft.so."""
    want = source_hash()
    if not force and _stamp_ok(LIB_PATH, STAMP_PATH, want):
        return LIB_PATH
    with build_lock(LIB_PATH):
        if not force and _stamp_ok(LIB_PATH, STAMP_PATH, want):
            return LIB_PATH
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        tmp = LIB_PATH + f".tmp{os.getpid()}"
        objs = [os.path.join(LIB_DIR, f"{os.path.splitext(f)[0]}.tmp{os.getpid()}.o") for f in SOURCES]
        cmds = [[hipcc] + FLAGS + ["-c", os.path.join(CSRC, f), "-o", o] for f, o in zip(SOURCES, objs)]
        try:
            procs = []
            for cmd in cmds:                      # the translation units compile side by side
                if verbose:
                    print(" ".join(cmd))
                procs.append(subprocess.Popen(cmd, env=compile_env()))
            rcs = [p.wait() for p in procs]
            if any(rcs):
                raise subprocess.CalledProcessError(max(rcs), cmds[rcs.index(max(rcs))])
            link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(link))
            subprocess.run(link, check=True, env=compile_env())
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
        os.replace(tmp, LIB_PATH)               # atomic: a concurrent loader sees the old or the new file
        with open(STAMP_PATH, "w") as fh:
            fh.write(want + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
