#!/bin/bash
# Round-6 GPU steps, one per gpurun call:  bash tools/gpu_r06.sh WHAT [TAG]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WHAT=${1:-new}
TAG=${2:-r06}
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
pmc() {  # pmc NAME "COUNTERS" program args...: one rocprofv3 --pmc pass (kernel trace only) -> gpurun_out/pmc_${TAG}_NAME/
  local name=$1 ctrs=$2; shift 2
  local OUT="$R/gpurun_out/pmc_${TAG}_$name"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT" -- python3 "$@" \
      > /dev/null 2> "$OUT.err" ); local rc=$?
  echo "pmc $name [$ctrs] exit=$rc"; ok $rc || return 1
  [ $rc -eq 0 ] || tail -3 "$OUT.err"
  return 0
}
case "$WHAT" in
new)  # what round 6 added so far: the config-1 known answers, the config3 block (N = 1 and 2 / 4 ranks through the stand-in),
      # the fallback warnings; then the driver's command (does the 64M-path block fit and how long does it take?)
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nn.py::test_a_shape_outside_the_kernels_says_so_once \
      tests/test_gpu_dist.py tests/test_gpu_multirank.py -x -q -m gpu --durations=8 > gpurun_out/${TAG}_new_tests.log 2>&1; rc=$?
  tail -25 gpurun_out/${TAG}_new_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_driver.json 2> gpurun_out/bench_${TAG}_driver.err; rc=$?
  echo "bench (driver's command) exit=$rc"; tail -3 gpurun_out/bench_${TAG}_driver.err
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r06_driver.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"])
print("config3", json.dumps(d.get("config3"))[:1500])
PY
  ;;
sq)  # SQ counters of the three c2 kernels on the current library, one pass per counter group
  for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
              "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
    N=$(echo $PASS | cut -d' ' -f1)
    pmc "sqc2_$N" "$PASS" "$R/bench.py" --config c2 --steps 6 --warmup 2 --min-warmup-seconds 0.05 --only-timed || exit 1
  done
  python3 "$R/tools/summarize_sq.py" "$R/gpurun_out" "pmc_${TAG}_sqc2_" | tee "$R/gpurun_out/pmc_sq_summary_${TAG}_c2.txt"
  for d in "$R"/gpurun_out/pmc_${TAG}_sqc2_*; do [ -d "$d" ] && rm -rf "$d"; done
  true
  ;;
tch)  # (full storage: the step predates folded storage, whose pass 1 runs chunks of 32 steps) pass 1's reads by time-chunk length: every (tile, chunk) re-reads the terminal row and its last prefetches are clamped
      # duplicates -> exact request-size counters + timing at c3 and c2 for OMC_PASS1_TCHUNK = auto / 32 / 63 / 126 / 251
  for CFG in c3 c2; do
    for TCH in 0 32 63 126 251; do
      export OMC_PASS1_TCHUNK=$TCH
      timeout -k 10 300 python bench.py --config $CFG --steps 12 --warmup 4 --only-timed --storage full > gpurun_out/${TAG}_tch_${CFG}_${TCH}.json 2> gpurun_out/${TAG}_tch_${CFG}_${TCH}.err; rc=$?
      ok $rc || exit 1
      python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_tch_${CFG}_${TCH}.json')); k={x['kernel']:x for x in d['roofline_kernels']}
print('$CFG tchunk=$TCH ms_per_step', round(d['ms_per_step'],4), 'pass1 ms', round(k['lsm_pass1_kernel']['ms_per_launch'],4), 'frac', round(k['lsm_pass1_kernel']['frac'],4), 'gen', round(k['gbm_paths_kernel']['ms_per_launch'],4), 'pass2', round(k['lsm_pass2_kernel']['ms_per_launch'],4))" | tee -a gpurun_out/${TAG}_tchunk_sweep.txt
    done
  done
  for TCH in 0 126 251; do
    export OMC_PASS1_TCHUNK=$TCH
    pmc "tch_c3_${TCH}_a" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
    pmc "tch_c3_${TCH}_b" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
    pmc "tch_c3_${TCH}_c" "TCC_HIT_sum TCC_MISS_sum" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
    pmc "tch_c3_${TCH}_d" "FETCH_SIZE" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
  done
  unset OMC_PASS1_TCHUNK
  python3 "$R/tools/summarize_sq.py" "$R/gpurun_out" "pmc_${TAG}_tch_c3_" --by-dir | tee "$R/gpurun_out/${TAG}_pass1_reads_by_tchunk.txt"
  for d in "$R"/gpurun_out/pmc_${TAG}_tch_c3_*; do [ -d "$d" ] && rm -rf "$d"; done
  true
  ;;
calib)  # f-3 measured: the calibrator's objective evaluation, surface call vs per-expiry loop; kernel stats; VALU share of the simulation
  timeout -k 10 600 python -m pytest tests/test_gpu_calibrator.py tests/test_gpu_ols7.py tests/test_gpu_multirank.py::test_ols7_a_rank_without_room_for_its_paths_fails_the_call_on_every_rank tests/test_gpu_nn_dist.py -x -q -m gpu --durations=5 > gpurun_out/${TAG}_calib_tests.log 2>&1; rc=$?
  tail -12 gpurun_out/${TAG}_calib_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  [ $rc -eq 0 ] || exit 1
  timeout -k 10 300 python tools/bench_calibrator.py > gpurun_out/${TAG}_calibrator.json 2> gpurun_out/${TAG}_calibrator.err; rc=$?
  cat gpurun_out/${TAG}_calibrator.json; tail -2 gpurun_out/${TAG}_calibrator.err; ok $rc || exit 1
  prof calibrator "$R/tools/bench_calibrator.py" --evals 100 --no-cpu || exit 1
  pmc "calib_a" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "$R/tools/bench_calibrator.py" --evals 20 --no-cpu || exit 1
  pmc "calib_b" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "$R/tools/bench_calibrator.py" --evals 20 --no-cpu || exit 1
  python3 "$R/tools/summarize_sq.py" "$R/gpurun_out" "pmc_${TAG}_calib_" | tee "$R/gpurun_out/${TAG}_pmc_calibrator.txt" | head -60
  for d in "$R"/gpurun_out/pmc_${TAG}_calib_*; do [ -d "$d" ] && rm -rf "$d"; done
  true
  ;;
dup)  # (needs commit 38a8e3a: the OMC_PASS1_DUP switch) pass 1: where the chunk's last two (unused) prefetches point -- its last row (0) or the row being processed (1)
  for REP in 1 2; do
    for CFG in c3 c2; do
      for DUP in 0 1; do
        export OMC_PASS1_DUP=$DUP
        timeout -k 10 300 python bench.py --config $CFG --steps 12 --warmup 4 --only-timed > gpurun_out/${TAG}_dup_${CFG}_${DUP}.json 2> gpurun_out/${TAG}_dup_${CFG}_${DUP}.err; rc=$?
        ok $rc || exit 1
        python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_dup_${CFG}_${DUP}.json')); k={x['kernel']:x for x in d['roofline_kernels']}
print('$CFG dup=$DUP rep $REP ms_per_step', round(d['ms_per_step'],4), 'pass1 ms', round(k['lsm_pass1_kernel']['ms_per_launch'],4), 'frac', round(k['lsm_pass1_kernel']['frac'],4))" | tee -a gpurun_out/${TAG}_pass1_dup.txt
      done
    done
  done
  for DUP in 0 1; do
    export OMC_PASS1_DUP=$DUP
    pmc "dup_c3_${DUP}_a" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
    pmc "dup_c3_${DUP}_b" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "$R/bench.py" --config c3 --steps 3 --warmup 1 --min-warmup-seconds 0.02 --only-timed --storage full || exit 1
  done
  unset OMC_PASS1_DUP
  python3 "$R/tools/summarize_sq.py" "$R/gpurun_out" "pmc_${TAG}_dup_c3_" --by-dir | grep -A3 "====\|pass1" | tee -a "$R/gpurun_out/${TAG}_pass1_dup.txt"
  for d in "$R"/gpurun_out/pmc_${TAG}_dup_c3_*; do [ -d "$d" ] && rm -rf "$d"; done
  true
  ;;
fused)  # EXPERIMENT (needs commit 38a8e3a, where the kernels live): the default call's optimizer step without its Adam launch (OMC_MLP_FUSED = 1 | 2) vs the product
  for CASE in "225057 256 4 128 3 0.1" "225057 256 4 64 2 0.1" "5000 256 8 128 3 0.1" "225057 1024 4 128 3 0.1" "225057 256 4 128 3 0.0"; do
    timeout -k 10 900 python tools/exp_fused_step.py $CASE 2>&1 | tee -a gpurun_out/${TAG}_fused_step_experiment.txt; rc=${PIPESTATUS[0]}
    ok $rc || exit 1
  done
  true
  ;;
fusedprof)  # rocprofv3 kernel stats of the three step forms (the experiment's child process, 2 epochs of the default call's shape)
  for F in 0 1 2; do
    export OMC_MLP_FUSED=$F
    prof fused_step_$F "$R/tools/exp_fused_step.py" --child 225057 256 2 128 3 0.1 || exit 1
    cat gpurun_out/${TAG}_fused_step_${F}_under_rocprof.json | cut -c1-300
  done
  unset OMC_MLP_FUSED
  true
  ;;
c5)  # config 5 and the default call after the rows rework: lines + kernel stats
  timeout -k 10 400 python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/bench_${TAG}_c5.json 2> gpurun_out/bench_${TAG}_c5.err; rc=$?
  echo "bench c5 exit=$rc"; ok $rc || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_${TAG}_c5.json')); print('c5 ms_per_step', d['ms_per_step'], 'timings', d['timings_ms'], 'price', d['price'], 'rows', d['rows'])"
  prof c5 "$R/bench.py" --config c5 --steps 2 --warmup 1 || exit 1
  grep "rows_\|Name" gpurun_out/${TAG}_c5_kernel_stats.csv | cut -c1-200
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
  # (the driver's command also runs the config3 block: the SAME kernels on 64M paths.  --stats averages by kernel name, so
  #  the c2 averages the roofline is recomputed from are taken with --no-config3; the full command's stats beside them)
  prof two_pass "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained --no-config3 || exit 1
  prof driver_full "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof reference "$R/bench.py" --gpus 1 --steps 32 --warmup 5 --semantics reference --group 16 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c3 "$R/bench.py" --config c3 --steps 10 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c4 "$R/bench.py" --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  prof c5 "$R/bench.py" --config c5 --steps 2 --warmup 1 || exit 1
  prof c1nn "$R/bench.py" --config c1nn --steps 5 --warmup 2 || exit 1
  ;;
rows)  # NN pass 1 alone at config 5's size: timing, kernel stats, SQ counters of the two sweeps
  timeout -k 10 300 python tools/time_rows.py > gpurun_out/${TAG}_rows.json 2> gpurun_out/${TAG}_rows.err; rc=$?
  cat gpurun_out/${TAG}_rows.json | cut -c1-400; ok $rc || exit 1
  timeout -k 10 300 python tools/time_rows.py 10000 50 12 > gpurun_out/${TAG}_rows_c1.json 2>> gpurun_out/${TAG}_rows.err; rc=$?
  cat gpurun_out/${TAG}_rows_c1.json | cut -c1-300; ok $rc || exit 1
  prof rows "$R/tools/time_rows.py" || exit 1
  pmc "rows_a" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "$R/tools/time_rows.py" 1000000 252 2 || exit 1
  pmc "rows_b" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "$R/tools/time_rows.py" 1000000 252 2 || exit 1
  pmc "rows_c" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$R/tools/time_rows.py" 1000000 252 2 || exit 1
  python3 "$R/tools/summarize_sq.py" "$R/gpurun_out" "pmc_${TAG}_rows_" | grep -A40 "^rows_" | tee "$R/gpurun_out/${TAG}_pmc_rows.txt" | head -90
  for d in "$R"/gpurun_out/pmc_${TAG}_rows_*; do [ -d "$d" ] && rm -rf "$d"; done
  true
  ;;
prof2)  # the c2 kernel averages without the config3 block mixed in, and the driver's full command beside them
  prof two_pass "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained --no-config3 || exit 1
  prof driver_full "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-sustained || exit 1
  ;;
rowstest)  # the NN pass-1 kernels against the oracle (quirks, ragged sizes, config 5's size, fuzz), then their timing
  timeout -k 10 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_nn_full.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py tests/test_gpu_nn_dist.py tests/test_gpu_ols7.py -x -q -m gpu -k "rows or normalis or quirk or pass1 or config5 or full or sharded or collective" --durations=5 > gpurun_out/${TAG}_rows_tests.log 2>&1; rc=$?
  tail -12 gpurun_out/${TAG}_rows_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  [ $rc -eq 0 ] || exit 1
  ;;
soak)  # the seeded fuzz sweeps against the C oracle, 30 (or $4) times as many cases from seeds shifted by $3 (one process, no -x: count every failure)
  OMC_FUZZ_SCALE=${4:-30} OMC_FUZZ_SEED=${3:-1000} timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_fuzz_soak.log 2>&1; rc=$?
  grep -v "^[.s]*\( *\[ *[0-9]*%\]\)\?$" gpurun_out/${TAG}_fuzz_soak.log | tail -40; echo "pytest exit=$rc"
  ;;
fold)  # antithetic-folded storage: its tests against the folded oracle, then the headline both ways (timed region only)
  timeout -k 10 600 python -m pytest tests/test_gpu_fold.py -q -m gpu --durations=5 > gpurun_out/${TAG}_fold_tests.log 2>&1; rc=$?
  tail -30 gpurun_out/${TAG}_fold_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  [ $rc -eq 0 ] || exit 1
  for ST in folded full; do
    timeout -k 10 300 python bench.py --gpus 1 --steps 50 --warmup 20 --storage $ST --only-timed --no-cpu-baseline > gpurun_out/bench_${TAG}_$ST.json 2> gpurun_out/bench_${TAG}_$ST.err; rc=$?
    echo "bench $ST exit=$rc"; ok $rc || exit 1
    python3 - "$ST" "$TAG" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_{sys.argv[2]}_{sys.argv[1]}.json"))
print(sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "price", d["price"])
for k in d["roofline_kernels"]:
    print("   ", k["kernel"], round(k["ms_per_launch"], 4), "ms", round(k["achieved"]), "GB/s", round(k["frac"], 3))
print("    whole", d["roofline_whole_pricing"])
PY
  done
  ;;
foldtune)  # folded kernels: tiles per wave of pass 1 (OMC_FOLD_TPW) x columns per thread of pass 2 (OMC_FOLD_P2_VEC)
  for CFG in c2 c3; do
    for KNOB in "2 2 0" "1 2 0" "4 2 0" "2 4 0" "2 2 2" "2 2 1"; do
      set -- $KNOB
      OMC_FOLD_TPW=$1 OMC_FOLD_P2_VEC=$2 timeout -k 10 300 python bench.py --config $CFG --steps 30 --warmup 10 --only-timed --option gbm_vec=$3 > gpurun_out/${TAG}_ft_${CFG}_$1_$2.json 2> gpurun_out/${TAG}_ft_${CFG}_$1_$2.err; rc=$?
      ok $rc || exit 1
      python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_ft_${CFG}_$1_$2.json')); k=d['roofline_kernels']
print('$CFG tpw=$1 p2vec=$2 gbm_vec=$3 ms_per_step', round(d['ms_per_step'],4), ' '.join(x['kernel'].replace('lsm_','').replace('_kernel','')+' '+str(round(x['ms_per_launch'],4)) for x in k), 'price', d['price'])" | tee -a gpurun_out/${TAG}_fold_tune.txt
    done
  done
  ;;
tests)
  timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
  tail -25 gpurun_out/${TAG}_tests.log; echo "pytest exit=$rc"; ok $rc || exit 1
  timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?
  tail -4 gpurun_out/${TAG}_smoke.log; echo "smoke exit=$rc"
  ;;
*) echo "unknown step $WHAT"; exit 2;;
esac
