"""MI355X-native XC/Fock engine: drop-in for the XCSolver C-ABI of
knight46/Quantum_compute_DFT (src/dft_solver.h) and its dft.py driver surface.

The compute path is libdft.so (hand-written HIP for gfx950, see csrc/); this
package is the thin host side.  There is no CPU fallback: if the library is
missing or no GPU is usable, calls raise.
"""
from .build import LIB_PATH, build_library, library_path  # noqa: F401
from .solver import DFTSolverWrapper, default_library_path, load_library  # noqa: F401

__all__ = ["DFTSolverWrapper", "default_library_path", "load_library", "build_library", "library_path", "LIB_PATH"]
