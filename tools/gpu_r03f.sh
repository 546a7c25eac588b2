#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_step_multi.py tests/test_gpu_batch.py -x -q -m gpu > gpurun_out/r03f_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r03f_tests.log
tail -6 gpurun_out/r03f_tests.log
[ $rc -eq 0 ] || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --config c2 > gpurun_out/r03d_c2.json 2> gpurun_out/r03d_c2.err || { tail -5 gpurun_out/r03d_c2.err; exit 1; }
python - c2 <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r03d_{sys.argv[1]}.json'))
r=d['roofline_per_step']
print(sys.argv[1], 'best K', r['pricings_per_launch'], 'frac', round(r['frac'],3), 'ms/launch', round(r['ms_per_launch']*1e3,2),'us', 'path-steps/s', '%.3g'%r['path_steps_per_s'])
for k,v in r['by_pricings_per_launch'].items(): print('  K',k, 'frac',round(v['frac'],3),'us/launch',round(v['ms_per_launch']*1e3,2),'ms/pricing',round(v['ms_per_pricing'],3),'%.3g'%v['path_steps_per_s'])
PY
