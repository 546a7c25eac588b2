#!/bin/bash
# mlp_adam_kernel (16 parameters per workgroup) vs mlp_adam_wide_kernel (64: 256-byte reads of a partial): rocprofv3 kernel
# stats of one 2 x 64 epoch at batch 2^17 each way.  usage: prof_adam.sh TAG
set -u
R="$GRAFT_REPO_ROOT"; TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for W in 0 1; do
  OUT="$R/gpurun_out/prof_adam_${TAG}_$W"
  OMC_MLP_ADAM_WIDE=$W timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/bench_mlp.py" 16777216 131072 2 0.1 > "$R/gpurun_out/${TAG}_adam_wide$W.json" 2> "$OUT.err"; rc=$?
  echo "wide=$W exit=$rc"; [ $rc -eq 124 ] && exit 1
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/${TAG}_adam_wide${W}_kernel_stats.csv"; head -4 "$f" | cut -c1-150; cat "$R/gpurun_out/${TAG}_adam_wide$W.json"
  rm -rf "$OUT"
done
