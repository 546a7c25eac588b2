#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r03n_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r03n_tests.log
tail -30 gpurun_out/r03n_tests.log
