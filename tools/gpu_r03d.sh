#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in c2 c3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline --config $cfg > gpurun_out/r03d_$cfg.json 2> gpurun_out/r03d_$cfg.err || { tail -5 gpurun_out/r03d_$cfg.err; exit 1; }
python - $cfg <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r03d_{sys.argv[1]}.json'))
r=d['roofline_per_step']
print(sys.argv[1], 'best K', r['pricings_per_launch'], 'frac', round(r['frac'],3), 'ms/launch', round(r['ms_per_launch']*1e3,2),'us', 'path-steps/s', '%.3g'%r['path_steps_per_s'])
for k,v in r['by_pricings_per_launch'].items(): print('  K',k, 'frac',round(v['frac'],3),'us/launch',round(v['ms_per_launch']*1e3,2),'ms/pricing',round(v['ms_per_pricing'],3),'%.3g'%v['path_steps_per_s'])
print('  timed', d['ms_per_step'], 'sust', d['sustained']['ms_per_step'], d['timed_vs_sustained_ms'])
PY
done
