#!/bin/bash
# Experiment: move the four target-free regression sums from pass 1 (VALU-bound) into the path generator
# (store-bound, 41 % of its issue cycles idle)?  Builds the library with -DOMC_EXP_USUMS (generator does the extra
# arithmetic + a per-step wave reduction, pass 1 drops the four sums; PRICES ARE WRONG in this build) and compares the
# kernels' times with the default build.  The default library is rebuilt on exit.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
trap 'env -u OMC_HIPCC_FLAGS python -c "from options_model_amd import _build; _build.build(force=True)" > /dev/null 2>&1' EXIT
for V in "base:" "usums:-DOMC_EXP_USUMS"; do
  TAG=${V%%:*}; FLAGS=${V#*:}
  OMC_HIPCC_FLAGS="$FLAGS" timeout -k 10 400 python -c "from options_model_amd import _build; _build.build(force=True)" > gpurun_out/usums_build_$TAG.log 2>&1 || { tail -5 gpurun_out/usums_build_$TAG.log; exit 1; }
  for CFG in c2 c3; do
    OMC_HIPCC_FLAGS="$FLAGS" timeout -k 10 200 python bench.py --config $CFG --steps 20 --warmup 5 --only-timed > gpurun_out/usums_${TAG}_$CFG.json 2> gpurun_out/usums_${TAG}_$CFG.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
    python - "$TAG" "$CFG" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/usums_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "ms_per_step", round(d["ms_per_step"], 4), [(k["kernel"], round(k["ms_per_launch"], 4)) for k in d["roofline_kernels"]])
PY
  done
done
