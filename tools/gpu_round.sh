#!/bin/bash
# One gpurun call: GPU parity tests -> smoke -> bench -> rocprofv3 kernel trace.
# A step that was killed by its timeout stops the chain (no further GPU work after a hang).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${1:-r01}
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }

timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?
echo "pytest exit=$rc" | tee -a gpurun_out/gpu_tests_$TAG.log
tail -5 gpurun_out/gpu_tests_$TAG.log
ok $rc || exit 1

timeout -k 10 600 python __graft_entry__.py --smoke > gpurun_out/smoke_$TAG.log 2>&1; rc=$?
echo "smoke exit=$rc"; tail -6 gpurun_out/smoke_$TAG.log
ok $rc || exit 1

timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
echo "bench exit=$rc"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
ok $rc || exit 1

OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-variants \
    > "$GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.err"; rc=$?
echo "rocprof exit=$rc"; cat "$GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.json"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -20
