cd $GRAFT_REPO_ROOT
D="python -m quantum_compute_dft_amd.dft"
for rep in 1 2; do
  for e in "--eri cholesky --chol-tol 1e-8" ""; do
  echo "=== Benzene GGA def2-svp $e"
  timeout -k 10 200 $D GGA Benzene --basis def2-svp $e 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
  done
  echo "=== H2O LDA def2-svp"
  timeout -k 10 200 $D LDA H2O --basis def2-svp 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
done
