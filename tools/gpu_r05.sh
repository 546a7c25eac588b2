#!/bin/bash
# Round-5 GPU steps, one per gpurun call:  bash tools/gpu_r05.sh WHAT TAG
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WHAT=${1:-dropout}
TAG=${2:-r05}
R="$GRAFT_REPO_ROOT"
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
prof() {  # prof NAME program args...: rocprofv3 kernel stats of a python program -> gpurun_out/${TAG}_NAME_kernel_stats.csv
  local name=$1; shift
  local OUT="$R/gpurun_out/prof_${TAG}_$name"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$@" \
      > "$R/gpurun_out/${TAG}_${name}_under_rocprof.json" 2> "$OUT.err" ); local rc=$?
  echo "rocprof $name exit=$rc"; ok $rc || return 1
  local f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/${TAG}_${name}_kernel_stats.csv"; head -8 "$f" | cut -c1-170
  rm -rf "$OUT"
}
case "$WHAT" in
dropout)
  timeout -k 10 900 python -m pytest tests/test_gpu_dropout.py -x -q --durations=10 -s > gpurun_out/${TAG}_dropout.log 2>&1; rc=$?
  tail -30 gpurun_out/${TAG}_dropout.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python tools/time_default_pricer.py > gpurun_out/${TAG}_default_pricer.txt 2>&1; rc=$?
  cat gpurun_out/${TAG}_default_pricer.txt; ok $rc || exit 1
  prof default_pricer "$R/tools/prof_default_pricer.py" || exit 1
  cat gpurun_out/${TAG}_default_pricer_under_rocprof.json
  ;;
q16)
  timeout -k 10 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_mlp.py -x -q --durations=8 > gpurun_out/${TAG}_q16_tests.log 2>&1; rc=$?
  tail -15 gpurun_out/${TAG}_q16_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  [ $rc -eq 0 ] || exit 1
  for Q in 0 -1; do
    for B in 256 512 1024 2048; do
      OMC_MLP_Q16=$Q timeout -k 10 200 python tools/bench_mlp.py 225057 $B 4 0.1 3 128 >> gpurun_out/${TAG}_q16_sweep.jsonl 2>> gpurun_out/${TAG}_q16_sweep.err; rc=$?
      ok $rc || exit 1
    done
    OMC_MLP_Q16=$Q timeout -k 10 200 python tools/bench_mlp.py 225057 256 4 0.1 2 64 >> gpurun_out/${TAG}_q16_sweep.jsonl 2>> gpurun_out/${TAG}_q16_sweep.err
  done
  cat gpurun_out/${TAG}_q16_sweep.jsonl
  timeout -k 10 300 python tools/time_default_pricer.py > gpurun_out/${TAG}_default_pricer.txt 2>&1; rc=$?
  cat gpurun_out/${TAG}_default_pricer.txt; ok $rc || exit 1
  prof default_pricer "$R/tools/prof_default_pricer.py" || exit 1
  ;;
tests)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
  tail -25 gpurun_out/${TAG}_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?
  tail -4 gpurun_out/${TAG}_smoke.log; echo "smoke exit=$rc"
  ;;
*) echo "unknown step $WHAT"; exit 2;;
esac
