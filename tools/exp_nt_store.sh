#!/bin/bash
# A/B on the GPU box: path rows stored normally vs with the nontemporal hint (whole pricing, per-kernel times).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# whatever happens, leave the DEFAULT library behind (experiment builds must not outlive the experiment)
trap 'env -u OMC_HIPCC_FLAGS python -c "from options_model_amd import _build; _build.build(force=True)" > /dev/null 2>&1' EXIT
timeout -k 10 200 python bench.py --no-variants --only-timed > gpurun_out/nt_base.json 2> gpurun_out/nt_base.err; rc=$?
echo "base exit=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
OMC_HIPCC_FLAGS=-DOMC_NT_STORE=1 timeout -k 10 400 python -c "from options_model_amd import _build; _build.build(force=True)" > gpurun_out/nt_build.log 2>&1 || { tail -5 gpurun_out/nt_build.log; exit 1; }
timeout -k 10 200 python bench.py --no-variants --only-timed > gpurun_out/nt_on.json 2> gpurun_out/nt_on.err; rc=$?
echo "nt exit=$rc"
python - <<'PY'
import json
for tag in ("base", "on"):
    d = json.loads(open(f"gpurun_out/nt_{tag}.json").read().strip().splitlines()[-1])
    ks = {k["kernel"]: round(k["ms"], 4) for k in d.get("roofline_kernels", [])} if isinstance(d.get("roofline_kernels"), list) else d.get("roofline_kernels")
    print(tag, "ms_per_step", d["ms_per_step"], "sustained", d.get("sustained"), "\n   ", ks)
PY
