#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_contnet_batch.py tests/test_gpu_contnet.py tests/test_gpu_nn_curve.py tests/test_gpu_api.py tests/test_gpu_compat_gpu_file.py -x -q -m gpu -s > gpurun_out/r03i_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03i_tests.log
grep -n "ContNet flow\|40-point\|passed\|failed\|rc=\|Error\|assert" gpurun_out/r03i_tests.log | tail -20; tail -30 gpurun_out/r03i_tests.log
