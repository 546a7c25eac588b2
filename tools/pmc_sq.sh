#!/bin/bash
# SQ-side counters for the kernels of one bench run (where do the wave-cycles go?)
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
OUT="$R/gpurun_out/pmc_sq_$1"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -- \
  python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> "$OUT.err"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "omc::" not in k: continue
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:24s} {s / n:16.0f}")
PY
