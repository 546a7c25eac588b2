#!/bin/bash
# SQ counters of the "ols7" kernels at 1M x 252 (tools/time_ols7.py): how busy are the vector pipes, what do the waves wait for.
# usage: pmc_ols7.sh TAG
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  N=$(echo $PASS | cut -d' ' -f1)
  OUT="$R/gpurun_out/pmc_ols7_$1_$N"
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d "$OUT" -- \
    python3 "$R/tools/time_ols7.py" 1000000 252 > /dev/null 2> "$OUT.err" || { echo "pass $N failed"; tail -3 "$OUT.err"; }
done
python3 - "$R/gpurun_out" "$1" <<'PY' | tee "$R/gpurun_out/pmc_ols7_summary_$1.txt"
import csv, glob, sys, collections, re, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in glob.glob(os.path.join(sys.argv[1], "pmc_ols7_" + sys.argv[2] + "_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
            dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
for k, d in acc.items():
    if "ols7" not in k: continue
    us = dur[k][0] / max(dur[k][1], 1) / 1e3
    print(f"{k}: {us:.1f} us per dispatch")
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.0f} per dispatch")
PY
for d in "$R"/gpurun_out/pmc_ols7_$1_*; do [ -d "$d" ] && rm -rf "$d"; done
