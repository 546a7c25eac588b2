#!/bin/bash
# round 3, first GPU call: new multi-rank tests (stand-in librccl), dist tests, the driver's exact bench command
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03a_tests.log
tail -5 gpurun_out/r03a_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a_bench_driver.json 2> gpurun_out/r03a_bench_driver.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03a_bench_driver.json'))
print({k:d[k] for k in ('ms_per_step','clock_settled','warmup_by_time','kernel_event_samples','timed_vs_sustained_ms')})
print('sustained', d['sustained']['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
print('pathgen', d['roofline_pathgen']['frac'], 'per_step', d['roofline_per_step']['frac'], d['roofline_per_step']['ms_per_launch'])
PY
