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
seeds)
  timeout -k 10 300 python tools/nn_seed_study.py 42 1 2 3 4 5 6 7 > gpurun_out/${TAG}_seeds_q16.jsonl 2> gpurun_out/${TAG}_seeds.err; rc=$?; ok $rc || exit 1
  cat gpurun_out/${TAG}_seeds_q16.jsonl
  OMC_MLP_Q16=0 timeout -k 10 300 python tools/nn_seed_study.py 42 1 2 3 4 5 6 7 > gpurun_out/${TAG}_seeds_quad.jsonl 2>> gpurun_out/${TAG}_seeds.err; rc=$?; ok $rc || exit 1
  cat gpurun_out/${TAG}_seeds_quad.jsonl
  timeout -k 10 900 python tools/nn_seed_study.py --torch 42 1 2 3 4 5 6 7 > gpurun_out/${TAG}_seeds_torch.jsonl 2>> gpurun_out/${TAG}_seeds.err; rc=$?
  cat gpurun_out/${TAG}_seeds_torch.jsonl; tail -3 gpurun_out/${TAG}_seeds.err
  ;;
pmc)  # HBM bytes per launch for c2 (two flows), c3, c4: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md)
  cd /tmp && export TMPDIR=/tmp
  for CFG in c2 c3 c4; do
    case $CFG in c2) PPG=1000000; SEMS="two_pass reference";; c3) PPG=8000000; SEMS="two_pass";; c4) PPG=4000000; SEMS="two_pass";; esac
    for SEM in $SEMS; do
      EXTRA=""; [ $SEM = reference ] && EXTRA="--group 16 --steps 16"
      for CTR in FETCH_SIZE WRITE_SIZE; do
        OUT="$R/gpurun_out/pmc_${TAG}${CFG}${SEM}_$CTR"
        timeout -k 10 400 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT" -- \
            python3 "$R/bench.py" --config $CFG --steps 3 --warmup 1 --min-warmup-seconds 0.02 --semantics $SEM $EXTRA --only-timed > /dev/null 2> "$OUT.err"; rc=$?
        echo "pmc $CFG $SEM $CTR exit=$rc"; ok $rc || exit 1
      done
      python3 "$R/tools/summarize_pmc.py" "$R/gpurun_out" "${TAG}${CFG}${SEM}" $CFG $PPG "$TAG" | tee "$R/gpurun_out/pmc_summary_${TAG}_${CFG}_${SEM}.txt"
      rm -rf "$R/gpurun_out/pmc_${TAG}${CFG}${SEM}_FETCH_SIZE" "$R/gpurun_out/pmc_${TAG}${CFG}${SEM}_WRITE_SIZE"
    done
  done
  ;;
bench)  # the driver's command, then every config's line, then rocprofv3 kernel stats of the same commands
  cd "$R"
  timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; rc=$?
  echo "bench (driver's command) exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c3 --steps 10 --warmup 5 --no-variants > gpurun_out/bench_${TAG}_c3.json 2> gpurun_out/bench_${TAG}_c3.err; rc=$?
  echo "bench c3 exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c4 --steps 20 --warmup 5 --no-variants > gpurun_out/bench_${TAG}_c4.json 2> gpurun_out/bench_${TAG}_c4.err; rc=$?
  echo "bench c4 exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/bench_${TAG}_c5.json 2> gpurun_out/bench_${TAG}_c5.err; rc=$?
  echo "bench c5 exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c1nn --steps 5 --warmup 2 > gpurun_out/bench_${TAG}_c1nn.json 2> gpurun_out/bench_${TAG}_c1nn.err; rc=$?
  echo "bench c1nn exit=$rc"; ok $rc || exit 1
  prof two_pass "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof reference "$R/bench.py" --gpus 1 --steps 32 --warmup 5 --semantics reference --group 16 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c3 "$R/bench.py" --config c3 --steps 10 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c4 "$R/bench.py" --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c5 "$R/bench.py" --config c5 --steps 2 --warmup 1 || exit 1
  prof c1nn "$R/bench.py" --config c1nn --steps 5 --warmup 2 || exit 1
  ;;
nn)  # the NN flow's tests + the multi-rank files (rows sweep, dropout oracle, K vote)
  timeout -k 10 1100 python -m pytest tests/test_gpu_nn.py tests/test_gpu_nn_full.py tests/test_gpu_dropout.py tests/test_gpu_nn_curve.py tests/test_gpu_nn_dist.py tests/test_gpu_multirank.py tests/test_gpu_facade_ranks.py tests/test_gpu_dist.py -q --durations=12 -s > gpurun_out/${TAG}_nn_tests.log 2>&1; rc=$?
  grep -v "^\.*$" gpurun_out/${TAG}_nn_tests.log | tail -60; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python tools/time_c5.py > gpurun_out/${TAG}_c5.json 2> gpurun_out/${TAG}_c5.err; cat gpurun_out/${TAG}_c5.json | cut -c1-700
  ;;
tests)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
  tail -25 gpurun_out/${TAG}_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?
  tail -4 gpurun_out/${TAG}_smoke.log; echo "smoke exit=$rc"
  ;;
soak)  # the seeded fuzz sweeps against the C oracle, 30 (or $4) times as many cases from seeds shifted by $3 (one process, no -x: count every failure)
  OMC_FUZZ_SCALE=${4:-30} OMC_FUZZ_SEED=${3:-1000} timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_fuzz_soak.log 2>&1; rc=$?
  grep -v "^[.s]*\( *\[ *[0-9]*%\]\)\?$" gpurun_out/${TAG}_fuzz_soak.log | tail -40; echo "pytest exit=$rc"
  ;;
*) echo "unknown step $WHAT"; exit 2;;
esac
