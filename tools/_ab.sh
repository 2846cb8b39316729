cd $GRAFT_REPO_ROOT
D="python -m quantum_compute_dft_amd.dft"
for rep in 1 2; do
for opt in "--eigensolver exact" "--eigensolver rotate"; do
  echo "=== Benzene GGA def2-svp cholesky $opt"
  timeout -k 10 200 $D GGA Benzene --basis def2-svp --eri cholesky --chol-tol 1e-8 $opt 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
  echo "=== Benzene GGA def2-svp dense $opt"
  timeout -k 10 200 $D GGA Benzene --basis def2-svp $opt 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
  echo "=== H2O LDA def2-svp $opt"
  timeout -k 10 200 $D LDA H2O --basis def2-svp $opt 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
  echo "=== Benzene GGA sto-3g $opt"
  timeout -k 10 200 $D GGA Benzene --basis sto-3g $opt 2>&1 | grep -E "Total Energy:|Total Time|Median per cycle|rotat"
done
done
