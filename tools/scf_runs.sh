#!/bin/bash
# The driver end to end on the BASELINE molecules (real shells, real level-3 grids): logs for profiles/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r03}
rm -f $OUT/${TAG}_scf.jsonl
run() { name=$1; shift; echo "=== $name"; timeout -k 10 ${TO:-300} "$@" > $OUT/${TAG}_scf_$name.log 2>&1; echo "rc=$?"; grep -E "Total Energy|Converged|Median per cycle|Host part|Cholesky vectors|Unconverged|Error|Traceback" $OUT/${TAG}_scf_$name.log | head -8; }
D="python -m quantum_compute_dft_amd.dft"
run h2o_lda_def2svp $D LDA H2O --basis def2-svp --both-quirks --json $OUT/${TAG}_scf.jsonl
run benzene_gga_def2svp $D GGA Benzene --basis def2-svp --both-quirks --json $OUT/${TAG}_scf.jsonl
run benzene_gga_def2svp_cholesky $D GGA Benzene --basis def2-svp --eri cholesky --chol-tol 1e-8 --json $OUT/${TAG}_scf.jsonl
run benzene_gga_def2svp_2ranks python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 -m quantum_compute_dft_amd.dft GGA Benzene --basis def2-svp --dist-backend gloo --json $OUT/${TAG}_scf.jsonl
run anthracene_b3lyp_def2svp_cholesky $D B3LYP Anthracene --basis def2-svp --eri cholesky --chol-tol 1e-8 --json $OUT/${TAG}_scf.jsonl
run benzene_gga_def2svp_hostloop $D GGA Benzene --basis def2-svp --device-resident 0 --json $OUT/${TAG}_scf.jsonl
run anthracene_b3lyp_def2svp_cholesky_hostloop $D B3LYP Anthracene --basis def2-svp --eri cholesky --chol-tol 1e-8 --device-resident 0 --json $OUT/${TAG}_scf.jsonl
run anthracene_b3lyp_def2svp_cholesky_torchloop $D B3LYP Anthracene --basis def2-svp --eri cholesky --chol-tol 1e-8 --fused-tail 0 --json $OUT/${TAG}_scf.jsonl
TO=500 run anthracene_b3lyp_def2tzvp_cholesky $D B3LYP Anthracene --basis def2-tzvp --eri cholesky --chol-tol 1e-7 --json $OUT/${TAG}_scf.jsonl
# the tail kernels of the fused loop: per-kernel durations of a Benzene run, phases of the rotation kernel at three sizes
export TMPDIR=/tmp
rm -rf $OUT/prof_tail; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tail -- python3 -m quantum_compute_dft_amd.dft GGA Benzene --basis def2-svp > $OUT/${TAG}_tail_prof.log 2>&1; echo "tail prof rc=$?"
cp $(ls $OUT/prof_tail/*/*kernel_stats.csv | head -1) $OUT/${TAG}_tail_kernel_stats.csv
(timeout -k 10 100 python tools/tail_time.py; timeout -k 10 100 python tools/tail_time.py 246 47; timeout -k 10 100 python tools/tail_time.py 494 47) > $OUT/${TAG}_tail_time.txt 2>&1; echo "tail time rc=$?"
timeout -k 10 300 python tools/chol_dev_time.py > $OUT/${TAG}_chol_dev_time.txt 2>&1; echo "chol rc=$?"
timeout -k 10 200 python tools/graph_time.py h2o h2o_gga benzene_sto3g benzene > $OUT/${TAG}_graph_time.txt 2>&1; echo "graph rc=$?"
echo done
