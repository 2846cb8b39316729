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
import json
import os
import re
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdft.so")
STAMP_PATH = LIB_PATH + ".srchash"
SOURCES = ["dft_api.hip", "xc_occ.hip", "eri_cols.hip", "scf_tail.hip", "xc_tiny.hip"]   # one object each, compiled in parallel, linked into libdft.so
HEADERS = ["xc_functionals.hpp", "xc_kernels.hpp", "xc_ws_kernels.hpp", "xc_ws16_kernels.hpp", "xc_big_kernels.hpp",
           "xc_occ_kernels.hpp", "xc_occ_launch.hpp", "xc_tiny_kernels.hpp", "xc_tiny_launch.hpp", "jk_kernels.hpp", "ao_kernels.hpp", "cd_kernels.hpp", "device_util.hpp",
           os.path.join("..", "..", "include", "dft_solver.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950's register file is unified);
# without it hipcc 7.2 can wrap every MFMA group of a loop in v_accvgpr_write/read copy storms
# (measured on the fp64 probe: 35 -> 75 TFLOP/s, profiles/r01_mfma_f64_probe2.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]
# per-source additions (the reason is at the top of the source file)
EXTRA_FLAGS = {"xc_tiny.hip": ["-mllvm", "-disable-machine-licm"]}
RESOURCES_PATH = LIB_PATH + ".resources.json"

# Register-spill guard.  Every kernel's resource usage is read from the compiler's own report
# (-Rpass-analysis=kernel-resource-usage) and the build FAILS when a kernel spills vector registers or uses scratch
# memory -- a spill inside a hot loop costs more than any tuning gains, and it appears silently when a tile
# shape or launch bound changes (round 2: the half transform at 49-64 occupied orbitals spilled 10-75 VGPRs
# unnoticed).  Exempt: the first-generation validation kernels (options path = 1 / 2, never on the product path),
# whose per-thread arrays are indexed dynamically, and the entries of SPILL_ALLOW, each with its reason.
VALIDATION_KERNELS = ("k_rho_mfma", "k_vxc_mfma", "k_rho_valu", "k_vxc_valu")
SPILL_ALLOW = {
    # demangled-name prefix: (max scratch bytes per lane, reason)
    "void qcdft::k_vxc_ws<8, true, false, false>": (8, "one dword (the thread index) stored before and reloaded after the step loop, "
                                                       "never inside it (ISA checked); the 8-byte-load GGA variant for odd nao / "
                                                       "unaligned planes at nao 113-128 only"),
}


def parse_resource_usage(text):
    """{mangled kernel name: {vgprs, agprs, sgprs, vgpr_spill, sgpr_spill, scratch, occupancy, lds}} from the remarks
    hipcc prints with -Rpass-analysis=kernel-resource-usage."""
    out = {}
    for blk in re.split(r"remark: Function Name: ", text)[1:]:
        name = blk.split()[0]
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else 0
        out[name] = {"vgprs": g(r"  VGPRs"), "agprs": g("AGPRs"), "sgprs": g(r"  SGPRs"), "vgpr_spill": g("VGPRs Spill"),
                     "sgpr_spill": g("SGPRs Spill"), "scratch": g(r"ScratchSize \[bytes/lane\]"),
                     "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": g(r"LDS Size \[bytes/block\]")}
    return out


def check_spills(res):
    """Complaints (empty = ok) for the kernels of `res` (parse_resource_usage, keys demangled)."""
    bad = []
    for name, r in sorted(res.items()):
        if not (r["vgpr_spill"] or r["scratch"]):
            continue
        if any(v in name for v in VALIDATION_KERNELS):
            continue
        allow = [v for k, v in SPILL_ALLOW.items() if name.startswith(k)]
        if allow and r["scratch"] <= allow[0][0]:
            continue
        bad.append(f"{name}: {r['vgpr_spill']} VGPRs spilled, {r['scratch']} bytes of scratch per lane ({r['vgprs']} VGPRs, occupancy {r['occupancy']})")
    return bad


def source_hash(files=None, flags=None):
    h = hashlib.sha256((" ".join(flags if flags is not None else FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
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
        cmds = [[hipcc] + FLAGS + EXTRA_FLAGS.get(f, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", o] for f, o in zip(SOURCES, objs)]
        logs = [o + ".log" for o in objs]
        try:
            procs = []
            for cmd, log in zip(cmds, logs):      # the translation units compile side by side
                if verbose:
                    print(" ".join(cmd))
                procs.append(subprocess.Popen(cmd, env=compile_env(), stderr=open(log, "w")))
            rcs = [p.wait() for p in procs]
            texts = [open(log).read() for log in logs]
            for t in texts:                        # warnings and errors, without the thousands of remark lines
                keep = [l for l in t.splitlines() if re.search(r"(warning|error|fatal error):", l)]
                if keep:
                    print("\n".join(keep))
            if any(rcs):
                raise subprocess.CalledProcessError(max(rcs), cmds[rcs.index(max(rcs))])
            res = {}
            for t in texts:
                res.update(parse_resource_usage(t))
            names = list(res)
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
            res = {d if d else n: res[n] for n, d in zip(names, dem + [""] * (len(names) - len(dem)))}
            bad = check_spills(res)
            if bad:
                raise RuntimeError("register spills in product kernels (re-tile or relax the launch bound):\n  " + "\n  ".join(bad))
            with open(RESOURCES_PATH, "w") as fh:
                json.dump(res, fh, indent=0, sort_keys=True)
            link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(link))
            subprocess.run(link, check=True, env=compile_env())
        finally:
            for o in objs + logs:
                if os.path.exists(o):
                    os.remove(o)
        os.replace(tmp, LIB_PATH)               # atomic: a concurrent loader sees the old or the new file
        with open(STAMP_PATH, "w") as fh:
            fh.write(want + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
