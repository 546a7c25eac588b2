#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in "--group 10 --kernel-samples 8" "--group 20 --kernel-samples 8" "--group 20 --kernel-samples 1" "--group 20 --kernel-samples 8 --min-warmup-seconds 1.0" "--group 20 --kernel-samples 20"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline $cfg > gpurun_out/r03b_tmp.json 2>/dev/null || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r03b_tmp.json'))
print(sys.argv[1], '| timed', round(d['ms_per_step'],4), 'sustained', round(d['sustained']['ms_per_step'],4), 'ratio', round(d['timed_vs_sustained_ms'],4), 'samples', d['kernel_event_samples'], 'p1 frac', round(d['roofline']['frac'],3), 'warm', round(d['warmup_by_time']['seconds'],2), d['clock_settled'])
PY
done
