#!/bin/bash
# A/B on the GPU box: pass-2 variants (rows per block U, nontemporal loads) -- whole pricing and per-kernel times.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# whatever happens, leave the DEFAULT library behind (experiment builds must not outlive the experiment)
trap 'env -u OMC_HIPCC_FLAGS python -c "from options_model_amd import _build; _build.build(force=True)" > /dev/null 2>&1' EXIT
for V in "base:" "u4:-DOMC_P2_U=4" "u16:-DOMC_P2_U=16" "nt:-DOMC_P2_NT=1" "u16nt:-DOMC_P2_U=16 -DOMC_P2_NT=1"; do
  TAG=${V%%:*}; FLAGS=${V#*:}
  OMC_HIPCC_FLAGS="$FLAGS" timeout -k 10 400 python -c "from options_model_amd import _build; _build.build(force=True)" > gpurun_out/p2_build_$TAG.log 2>&1 || { tail -5 gpurun_out/p2_build_$TAG.log; exit 1; }
  timeout -k 10 200 python bench.py --no-variants --only-timed > gpurun_out/p2_$TAG.json 2> gpurun_out/p2_$TAG.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
  python - "$TAG" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/p2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), [(k["kernel"], round(k["ms_per_launch"], 4)) for k in d["roofline_kernels"]], "price", d["price"])
PY
done
