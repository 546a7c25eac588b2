#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_step_multi.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_contnet.py -x -q -m gpu > gpurun_out/r03e_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r03e_tests.log
tail -6 gpurun_out/r03e_tests.log
[ $rc -eq 0 ] && bash tools/gpu_r03d.sh
