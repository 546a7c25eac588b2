#!/bin/bash
# kernel-trace stats of the fused trainer; usage: prof_mlp.sh TAG
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
OUT="$R/gpurun_out/prof_mlp_$1"
timeout -k 10 300 python "$R/tools/bench_mlp.py" > "$R/gpurun_out/bench_mlp_$1.json" 2> "$OUT.err0" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python "$R/tools/bench_mlp.py" 16777216 131072 1 > /dev/null 2> "$OUT.err" || exit 1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cp "$f" "$R/gpurun_out/mlp_kernel_stats_$1.csv"
cat "$R/gpurun_out/bench_mlp_$1.json"; head -8 "$f"
