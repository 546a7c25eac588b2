#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_step_multi.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03c_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03c_tests.log
tail -15 gpurun_out/r03c_tests.log
