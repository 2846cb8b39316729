#!/bin/bash
# One GPU-box session for the round's evidence: tests, bench line, rocprofv3 kernel stats of the bench,
# and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) of a short bench.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${TAG:-r03}
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest_gpu.log
timeout -k 10 400 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
rm -rf $OUT/prof_$TAG; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-extra-legs --no-k-build > $OUT/${TAG}_bench_prof.json 2> $OUT/${TAG}_prof.err; echo "rocprof rc=$?"
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c; timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 5 --warmup 2 --spinup-ms 5 --no-cpu-baseline --no-extra-legs --no-k-build > /dev/null 2> $OUT/${TAG}_pmc_$c.err; echo "pmc $c rc=$?"
  find $OUT/pmc_$c -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_${c}_counter_collection.csv \;
done
rm -rf $OUT/pmc_kbuild; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_kbuild -- python3 tools/cd_time.py TZVP > $OUT/${TAG}_pmc_kbuild.log 2>&1; echo "pmc kbuild rc=$?"
find $OUT/pmc_kbuild -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_kbuild_counter_collection.csv \;
find $OUT/pmc_kbuild -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_pmc_kbuild_kernel_trace.csv \;
rm -rf $OUT/prof_ao; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ao -- python3 tools/ao_time.py > $OUT/${TAG}_ao_time.log 2>&1; echo "ao prof rc=$?"
find $OUT/prof_ao -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_ao_kernel_stats.csv \;
rm -rf $OUT/pmc_ao; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_ao -- python3 tools/ao_time.py > /dev/null 2>&1; echo "ao pmc rc=$?"
find $OUT/pmc_ao -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_ao_WRITE_SIZE_counter_collection.csv \;
echo done
