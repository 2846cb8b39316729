"""Builds libdft.so (the C-ABI shared library) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdft.so")
SOURCES = ["dft_api.hip"]
HEADERS = ["xc_functionals.hpp", "xc_kernels.hpp", "xc_ws_kernels.hpp", "xc_big_kernels.hpp", "jk_kernels.hpp", "ao_kernels.hpp", "cd_kernels.hpp", "device_util.hpp",
           os.path.join("..", "..", "include", "dft_solver.h")]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> quantum_compute_dft_amd/lib/libdft.so."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950's register file is unified);
    # without it hipcc 7.2 can wrap every MFMA group of a loop in v_accvgpr_write/read copy storms
    # (measured on the fp64 probe: 35 -> 75 TFLOP/s, profiles/r01_mfma_f64_probe2.txt).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]
    cmd += [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
