#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_nn_curve.py tests/test_gpu_mlp.py tests/test_gpu_nn.py tests/test_gpu_contnet.py -x -q -m gpu -s > gpurun_out/r03h_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03h_tests.log
grep -n "40-point\|config-1 NN prices\|passed\|failed\|rc=" gpurun_out/r03h_tests.log | tail; tail -25 gpurun_out/r03h_tests.log
