#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof_nn
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nn/run -o nn -- python3 $GRAFT_REPO_ROOT/tools/prof_nn_curve.py > $GRAFT_REPO_ROOT/gpurun_out/prof_nn/run.log 2>&1
grep "40-point" $GRAFT_REPO_ROOT/gpurun_out/prof_nn/run.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_nn/run -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_nn/nn_curve_kernel_stats.csv
head -8 "$f" | cut -c1-160
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_nn/run
