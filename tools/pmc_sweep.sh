#!/bin/bash
# rocprofv3 counter passes over the sweep kernels (each --pmc set in its own run, kernel-trace only; the program goes
# directly after --): MFMA-pipe utilisation and HBM traffic of k_rho_ws / k_vxc_ws / k_rho_occ_rs (Benzene shape) and
# k_rho_big64 / k_vxc_big / k_rho_occ (Anthracene/def2-TZVP shape).  Summarised by tools/pmc_sweep_summary.py.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${TAG:-r03}
run() { # name counters...
  local name=$1; shift
  rm -rf $OUT/pmc_sweep_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_sweep_$name -- python3 tools/pmc_sweep_driver.py > $OUT/${TAG}_pmc_sweep_$name.log 2>&1
  echo "pmc $name rc=$?"
  find $OUT/pmc_sweep_$name -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_sweep_${name}_counter_collection.csv \;
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
python3 tools/pmc_sweep_summary.py $TAG $OUT
