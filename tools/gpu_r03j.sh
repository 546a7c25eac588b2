#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof_cn
cd /tmp && export TMPDIR=/tmp
for w in one job; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cn/$w -o cn -- python3 $GRAFT_REPO_ROOT/tools/prof_contnet.py $w > $GRAFT_REPO_ROOT/gpurun_out/prof_cn/$w.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_cn/$w.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cn/$w -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_cn/${w}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), 'tot_us', round(float(r['TotalDurationNs'])/1e3,1), 'avg_us', round(float(r['AverageNs'])/1e3,2), r['Percentage'])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_cn/$w
done
