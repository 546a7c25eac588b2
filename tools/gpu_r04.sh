#!/bin/bash
# Round-4 measurement steps, one per gpurun call:
#   bash tools/gpu_r04.sh tests TAG   the driver's GPU test command (pytest -x -q -m gpu), then smoke
#   bash tools/gpu_r04.sh bench TAG   the driver's exact bench command + rocprofv3 kernel stats (two_pass, reference) + c3 / c4 lines
#   bash tools/gpu_r04.sh pmc TAG     PMC passes: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WHAT=${1:-bench}
TAG=${2:-r04}
R="$GRAFT_REPO_ROOT"
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
case "$WHAT" in
tests)
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
  tail -25 gpurun_out/${TAG}_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?
  tail -4 gpurun_out/${TAG}_smoke.log; echo "smoke exit=$rc"
  ;;
bench)
  timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; rc=$?
  echo "bench (driver's command) exit=$rc"; ok $rc || exit 1
  timeout -k 10 500 python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; rc=$?
  echo "bench (defaults) exit=$rc"; ok $rc || exit 1
  cd /tmp && export TMPDIR=/tmp
  for SEM in two_pass reference; do
    OUT="$R/gpurun_out/prof_${TAG}_$SEM"
    EXTRA=""; [ $SEM = reference ] && EXTRA="--group 16 --steps 32"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
        python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --semantics $SEM $EXTRA --no-cpu-baseline --no-variants --no-sustained \
        > "$R/gpurun_out/bench_prof_${TAG}_$SEM.json" 2> "$R/gpurun_out/prof_${TAG}_$SEM.err"; rc=$?
    echo "rocprof $SEM exit=$rc"; ok $rc || exit 1
    f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/${TAG}_${SEM}_kernel_stats.csv"; head -9 "$f" | cut -c1-150
    rm -rf "$OUT"
  done
  cd "$R"
  timeout -k 10 400 python bench.py --config c3 --steps 10 --warmup 5 --no-variants > gpurun_out/bench_${TAG}_c3.json 2> gpurun_out/bench_${TAG}_c3.err; rc=$?
  echo "bench c3 exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c4 --steps 20 --warmup 5 --no-variants > gpurun_out/bench_${TAG}_c4.json 2> gpurun_out/bench_${TAG}_c4.err; rc=$?
  echo "bench c4 exit=$rc"
  ;;
c5)
  # config 5 (NN 2 x 64): the drop-in call timed, then the same under rocprofv3 kernel stats
  timeout -k 10 300 python tools/time_c5.py > gpurun_out/${TAG}_c5.json 2> gpurun_out/${TAG}_c5.err; rc=$?
  echo "c5 exit=$rc"; cat gpurun_out/${TAG}_c5.json | cut -c1-600; ok $rc || exit 1
  cd /tmp && export TMPDIR=/tmp
  OUT="$R/gpurun_out/prof_${TAG}_c5"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/time_c5.py" \
      > "$R/gpurun_out/${TAG}_c5_under_rocprof.json" 2> "$OUT.err"; rc=$?
  echo "rocprof c5 exit=$rc"; ok $rc || exit 1
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/${TAG}_c5_kernel_stats.csv"; head -12 "$f" | cut -c1-170
  rm -rf "$OUT"
  cd "$R"
  # dropout's share of a tile: the trainer with and without it, batch 2^17 and 2^19
  timeout -k 10 300 python tools/bench_mlp.py 16777216 131072,524288 2 0.0,0.1 > gpurun_out/${TAG}_mlp_dropout.jsonl 2> gpurun_out/${TAG}_mlp_dropout.err; rc=$?
  echo "mlp dropout on/off exit=$rc"; cat gpurun_out/${TAG}_mlp_dropout.jsonl
  ;;
heston_sq)
  # SQ counters of the Heston generator at config 4
  cd /tmp && export TMPDIR=/tmp
  OUT="$R/gpurun_out/pmc_sq_${TAG}_c4"
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -- \
    python3 "$R/bench.py" --config c4 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed > /dev/null 2> "$OUT.err"; rc=$?
  echo "heston sq exit=$rc"; ok $rc || exit 1
  python3 - "$OUT" <<'PY' | tee "$R/gpurun_out/pmc_sq_summary_${TAG}_c4.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "omc::" not in k: continue
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:24s} {s / n:16.0f}   ({n} dispatches)")
PY
  rm -rf "$OUT"
  ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  for SEM in two_pass reference; do
    EXTRA=""; [ $SEM = reference ] && EXTRA="--group 16 --steps 16"
    for CTR in FETCH_SIZE WRITE_SIZE; do
      OUT="$R/gpurun_out/pmc_${TAG}${SEM}_$CTR"
      timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT" -- \
          python3 "$R/bench.py" --steps 3 --warmup 1 --min-warmup-seconds 0.02 --semantics $SEM $EXTRA --only-timed > /dev/null 2> "$OUT.err"; rc=$?
      echo "pmc $SEM $CTR exit=$rc"; ok $rc || exit 1
    done
    python3 "$R/tools/summarize_pmc.py" "$R/gpurun_out" "${TAG}${SEM}" c2 1000000 "$TAG" | tee "$R/gpurun_out/pmc_summary_${TAG}${SEM}.txt"
    rm -rf "$R/gpurun_out/pmc_${TAG}${SEM}_FETCH_SIZE" "$R/gpurun_out/pmc_${TAG}${SEM}_WRITE_SIZE"
  done
  ;;
esac
