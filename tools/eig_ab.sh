#!/bin/bash
# SCF cycle time with the occupied-subspace rotation (default) against eigh(F, S) every cycle.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
D="python -m quantum_compute_dft_amd.dft"
for es in auto; do
  for job in "B3LYP Anthracene --basis def2-tzvp --eri cholesky --chol-tol 1e-7"; do
    echo "=== $es : $job"
    timeout -k 10 500 $D $job --eigensolver $es 2>&1 | grep -E "Total Energy:|Median per cycle|Eigensolver|Host part|Unconverged|Error|Traceback"
  done
done
