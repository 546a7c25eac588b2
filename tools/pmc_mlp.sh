#!/bin/bash
# SQ counters of the fused trainer kernels; usage: pmc_mlp.sh TAG [WAVES]
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
export OMC_MLP_WAVES=${2:-8}
OUT="$R/gpurun_out/pmc_mlp_$1"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -- \
  python "$R/tools/bench_mlp.py" 8388608 262144 1 0.1 > /dev/null 2> "$OUT.err"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "mlp" not in k: continue
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:28s} {s / n:16.0f}")
PY
