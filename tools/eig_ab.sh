#!/bin/bash
# SCF cycle time with the occupied-subspace rotation against eigh(F, S) every cycle (the driver end to end).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
D="python -m quantum_compute_dft_amd.dft"
for job in "GGA Benzene --basis def2-svp --eri cholesky --chol-tol 1e-8" "B3LYP Anthracene --basis def2-svp --eri cholesky --chol-tol 1e-8" "B3LYP Anthracene --basis def2-tzvp --eri cholesky --chol-tol 1e-7"; do
  for es in rotate exact; do
    echo "=== $es : $job"
    timeout -k 10 500 $D $job --eigensolver $es 2>&1 | grep -E "Total Energy:|Median per cycle|Eigensolver|Unconverged|Error|Traceback"
  done
done
