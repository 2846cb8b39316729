"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/xc_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (quantum_compute_dft_amd) never does.
"""
from .loader import (  # noqa: F401
    build, lib, compute_xc, coulomb, exchange, jk_from_factors, pointwise, eval_ao, POINTWISE_KINDS,
)
