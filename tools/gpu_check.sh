#!/bin/bash
# One GPU-box session: probe, parity tests, bench, rocprof summary.  Stops after a timeout/kill.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "    rc=$rc"; tail -n "${TAILN:-15}" "$OUT/$name.log"
  if [ $rc -ge 124 ]; then echo "TIMEOUT/KILL in $name: stopping"; exit $rc; fi
  return 0
}
[ -x tools/mfma_f64_probe ] && [ -z "$SKIP_PROBE" ] && step probe 120 tools/mfma_f64_probe
[ -z "$SKIP_TESTS" ] && step pytest_gpu 900 python -m pytest tests -q -m gpu ${PYTEST_ARGS:-}
[ -z "$SKIP_BENCH" ] && step bench 600 python bench.py ${BENCH_ARGS:-}
if [ -n "$DO_PROF" ]; then
  export TMPDIR=/tmp
  step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$PWD/$OUT/prof" -- python3 bench.py --no-cpu-baseline ${BENCH_ARGS:-}
  find $OUT/prof -name "*kernel_stats.csv" | head -3
fi
echo "=== done"
