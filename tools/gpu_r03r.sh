#!/bin/bash
# longer 4-rank rehearsal on one GPU (stand-in librccl): default steps / warm-up, sustained loop, per-step K sweep, p2p
set -o pipefail
mkdir -p gpurun_out
LIB=$(python -c "import sys; sys.path.insert(0,'tests'); import rccl_standin; print(rccl_standin.build())")
for EXTRA in "" "--p2p-exchange"; do
OMC_RCCL_LIB=$LIB timeout -k 10 500 python bench.py --gpus 4 --single-device --backend rccl --paths-per-gpu 250000 --no-variants --no-cpu-baseline $EXTRA > gpurun_out/r03r_rehearsal.json 2> gpurun_out/r03r_rehearsal.err
rc=$?; echo "rehearsal $EXTRA rc=$rc"; [ $rc -eq 0 ] || { tail -20 gpurun_out/r03r_rehearsal.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03r_rehearsal.json"))
print({k:d[k] for k in ("n_gpus","rccl_ranks","comm","seq_overlap","ms_per_step","clock_settled","kernel_event_samples")})
print("sustained", d["sustained"]["pricings"], d["sustained"]["ms_per_step"], "price_check", d["price_check"]["rel_err"])
r=d["roofline_per_step"]; print("per-step", r["pricings_per_launch"], r["exchange_across_ranks"][:40], {k:round(v["ms_per_launch"]*1e3,1) for k,v in r["by_pricings_per_launch"].items()})
PY
done
