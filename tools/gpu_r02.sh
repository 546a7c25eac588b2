#!/bin/bash
# Round-2 GPU steps, one per gpurun call (a call is capped at 20 minutes):
#   bash tools/gpu_r02.sh tests TAG      GPU parity tests + smoke
#   bash tools/gpu_r02.sh sweep TAG      instruction-rate microbenchmarks + kernel experiment sweeps
#   bash tools/gpu_r02.sh bench TAG      bench.py default line + rocprofv3 kernel traces (+ per-config lines)
#   bash tools/gpu_r02.sh pmc TAG        PMC passes (HBM traffic, SQ counters)
# A step killed by its timeout stops the chain (no further GPU work after a hang).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WHAT=${1:-tests}
TAG=${2:-r02}
R="$GRAFT_REPO_ROOT"
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }

case "$WHAT" in
tests)
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?
  echo "pytest exit=$rc" | tee -a gpurun_out/gpu_tests_$TAG.log
  tail -15 gpurun_out/gpu_tests_$TAG.log
  ok $rc || exit 1
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke_$TAG.log 2>&1; rc=$?
  echo "smoke exit=$rc"; tail -6 gpurun_out/smoke_$TAG.log
  ;;
sweep)
  timeout -k 10 200 tools/_ubench > gpurun_out/ubench_$TAG.txt 2>&1; rc=$?
  echo "ubench exit=$rc"; cat gpurun_out/ubench_$TAG.txt
  ok $rc || exit 1
  for W in ${3:-step pass1 heston}; do
    timeout -k 10 400 python tools/r02_sweep.py $W > gpurun_out/sweep_${W}_$TAG.jsonl 2> gpurun_out/sweep_${W}_$TAG.err; rc=$?
    echo "sweep $W exit=$rc"
    python - "$R/gpurun_out/sweep_${W}_$TAG.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "error" in d:
        print("ERR", d["spec"], d["error"][-300:]); continue
    s = d["spec"]
    print(f"{s['sem']:9s} {s.get('model','gbm'):6s} M={s['M']:>8d} opt={s.get('options',{})} env={d['env']} wall={d['ms_wall']:.3f} "
          f"paths={d['ms_paths']:.3f} lsm={d['ms_lsm_avg']:.3f} p1={d['ms_pass1']:.3f} p2={d['ms_pass2']:.3f} price={d['price']:.6f}")
PY
    ok $rc || exit 1
  done
  ;;
bench)
  timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
  echo "bench exit=$rc"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
  ok $rc || exit 1
  cd /tmp && export TMPDIR=/tmp
  for SEM in two_pass reference; do
    OUT="$R/gpurun_out/prof_${TAG}_$SEM"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
        python "$R/bench.py" --semantics $SEM --no-cpu-baseline --no-variants --no-sustained \
        > "$R/gpurun_out/bench_prof_${TAG}_$SEM.json" 2> "$R/gpurun_out/prof_${TAG}_$SEM.err"; rc=$?
    echo "rocprof $SEM exit=$rc"
    ok $rc || exit 1
    find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
  done
  OUT="$R/gpurun_out/prof_${TAG}_c4"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
      python "$R/bench.py" --config c4 --steps 20 --warmup 20 --no-cpu-baseline --no-variants --no-sustained \
      > "$R/gpurun_out/bench_prof_${TAG}_c4.json" 2> "$R/gpurun_out/prof_${TAG}_c4.err"; rc=$?
  echo "rocprof c4 exit=$rc"
  find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r head -8
  ok $rc || exit 1
  cd "$R"
  # the other single-GPU configurations as plain bench lines (no profiler)
  timeout -k 10 400 python bench.py --config c3 --steps 10 --warmup 5 --no-variants > gpurun_out/bench_${TAG}_c3.json 2> gpurun_out/bench_${TAG}_c3.err; rc=$?
  echo "bench c3 exit=$rc"; ok $rc || exit 1
  timeout -k 10 400 python bench.py --config c4 --steps 20 --warmup 20 --no-variants > gpurun_out/bench_${TAG}_c4.json 2> gpurun_out/bench_${TAG}_c4.err; rc=$?
  echo "bench c4 exit=$rc"; ok $rc || exit 1
  timeout -k 10 500 python bench.py --config c3x1 --steps 5 --warmup 3 --no-variants > gpurun_out/bench_${TAG}_c3x1.json 2> gpurun_out/bench_${TAG}_c3x1.err; rc=$?
  echo "bench c3x1 exit=$rc"
  ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  # HBM traffic: separate passes per counter (TCC slots); every dispatch in a pass is the timed workload
  for SEM in two_pass reference; do
    for CTR in FETCH_SIZE WRITE_SIZE; do
      OUT="$R/gpurun_out/pmc_${TAG}${SEM}_$CTR"
      timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT" -- \
          python "$R/bench.py" --steps 3 --warmup 1 --semantics $SEM --only-timed \
          > /dev/null 2> "$OUT.err"; rc=$?
      echo "pmc $SEM $CTR exit=$rc"
      ok $rc || exit 1
    done
    python "$R/tools/summarize_pmc.py" "$R/gpurun_out" "${TAG}${SEM}" c2 1000000 "$TAG" | tee "$R/gpurun_out/pmc_summary_${TAG}${SEM}.txt"
  done
  for CTR in FETCH_SIZE WRITE_SIZE; do
    OUT="$R/gpurun_out/pmc_${TAG}c4_$CTR"
    timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT" -- \
        python "$R/bench.py" --config c4 --steps 3 --warmup 1 --only-timed > /dev/null 2> "$OUT.err"; rc=$?
    echo "pmc c4 $CTR exit=$rc"
    ok $rc || exit 1
  done
  python "$R/tools/summarize_pmc.py" "$R/gpurun_out" "${TAG}c4" c4 4000000 "$TAG" | tee "$R/gpurun_out/pmc_summary_${TAG}c4.txt"
  # where do the wave-cycles go (SQ counters), C2 two_pass and C4 (Heston generator)
  for CFG in c2 c4; do
    OUT="$R/gpurun_out/pmc_sq_${TAG}$CFG"
    timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d "$OUT" -- python "$R/bench.py" --config $CFG --steps 5 --warmup 10 --only-timed \
        > /dev/null 2> "$OUT.err"; rc=$?
    echo "pmc sq $CFG exit=$rc"
    ok $rc || exit 1
    python - "$OUT" <<'PY' | tee "$R/gpurun_out/pmc_sq_summary_${TAG}$CFG.txt"
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
  done
  ;;
persist)
  timeout -k 10 120 python tools/persist_stamps.py 1000000 > gpurun_out/persist_stamps_$TAG.txt 2>&1
  timeout -k 10 120 python tools/persist_stamps.py 8000000 >> gpurun_out/persist_stamps_$TAG.txt 2>&1
  timeout -k 10 120 python tools/step_stamps.py 1000000 > gpurun_out/step_stamps_$TAG.txt 2>&1
  cat gpurun_out/persist_stamps_$TAG.txt gpurun_out/step_stamps_$TAG.txt
  ;;
esac
