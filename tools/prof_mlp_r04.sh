#!/bin/bash
# Round 4: evidence for the NN trainer.  (1) us per optimizer step against the minibatch size (the
# fixed cost of a step vs the tile loop), (2) rocprofv3 kernel stats of a 2 x 64 epoch at batch 2^17 (config 5's shape),
# (3) the same for config 5 itself (bench.py's nn_2x64 variant).   usage: prof_mlp_r04.sh TAG
set -u
R="$GRAFT_REPO_ROOT"; TAG=${1:-r04}
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
cd "$R"
timeout -k 10 300 python tools/bench_mlp.py 16777216 8192,16384,32768,65536,131072,262144,524288 2 0.1 > gpurun_out/${TAG}_mlp_batch_sweep.jsonl 2> gpurun_out/${TAG}_mlp_batch_sweep.err; rc=$?
echo "batch sweep exit=$rc"; cat gpurun_out/${TAG}_mlp_batch_sweep.jsonl; ok $rc || exit 1
cd /tmp && export TMPDIR=/tmp
OUT="$R/gpurun_out/prof_mlp_$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/bench_mlp.py" 16777216 131072 2 0.1 > "$R/gpurun_out/${TAG}_mlp_under_rocprof.json" 2> "$OUT.err"; rc=$?
echo "rocprof exit=$rc"; ok $rc || exit 1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/${TAG}_mlp_kernel_stats.csv"; head -6 "$f" | cut -c1-160
rm -rf "$OUT"
