#!/bin/bash
# One gpurun call: GPU parity tests -> smoke -> bench -> rocprofv3 kernel trace (+ PMC passes).
# A step that was killed by its timeout stops the chain (no further GPU work after a hang).
# usage: bash tools/gpu_round.sh TAG [notests] [pmc]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${1:-r01}
shift || true
NOTESTS=0; PMC=0
for a in "$@"; do [ "$a" = notests ] && NOTESTS=1; [ "$a" = pmc ] && PMC=1; done
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
R="$GRAFT_REPO_ROOT"

if [ $NOTESTS -eq 0 ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?
  echo "pytest exit=$rc" | tee -a gpurun_out/gpu_tests_$TAG.log
  tail -5 gpurun_out/gpu_tests_$TAG.log
  ok $rc || exit 1
  timeout -k 10 600 python __graft_entry__.py --smoke > gpurun_out/smoke_$TAG.log 2>&1; rc=$?
  echo "smoke exit=$rc"; tail -6 gpurun_out/smoke_$TAG.log
  ok $rc || exit 1
fi

timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
echo "bench exit=$rc"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
ok $rc || exit 1

cd /tmp && export TMPDIR=/tmp
for SEM in two_pass reference; do
  OUT="$R/gpurun_out/prof_${TAG}_$SEM"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
      python "$R/bench.py" --steps 10 --warmup 2 --semantics $SEM --no-cpu-baseline --no-variants \
      > "$R/gpurun_out/bench_prof_${TAG}_$SEM.json" 2> "$R/gpurun_out/prof_${TAG}_$SEM.err"; rc=$?
  echo "rocprof $SEM exit=$rc"
  ok $rc || exit 1
  find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
done

if [ $PMC -eq 1 ]; then
  for CTR in FETCH_SIZE WRITE_SIZE; do
    OUT="$R/gpurun_out/pmc_${TAG}_$CTR"
    timeout -k 10 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT" -- \
        python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-variants \
        > /dev/null 2> "$R/gpurun_out/pmc_${TAG}_$CTR.err"; rc=$?
    echo "pmc $CTR exit=$rc"
    ok $rc || exit 1
  done
  python "$R/tools/summarize_pmc.py" "$R/gpurun_out" "$TAG" | tee "$R/gpurun_out/pmc_summary_$TAG.txt"
fi
