#!/bin/bash
# SQ / GRBM counters of the fused trainer (2 x 64, batch 2^17): which clock does the chip hold, how busy is the MFMA pipe.
# usage: pmc_mlp.sh TAG [bench_mlp.py arguments; default: 8388608 131072 1 0.1 = config 5's trainer.
#        The reference's default call: 225057 256 2 0.1 3 128]
set -u
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
TAGX=$1; shift; ARGS="${*:-8388608 131072 1 0.1}"; set -- "$TAGX"
for PASS in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  N=$(echo $PASS | cut -d' ' -f1)
  OUT="$R/gpurun_out/pmc_mlp_$1_$N"
  timeout -k 10 300 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d "$OUT" -- \
    python3 "$R/tools/bench_mlp.py" $ARGS > /dev/null 2> "$OUT.err" || { echo "pass $N failed"; tail -3 "$OUT.err"; }
done
python3 - "$R/gpurun_out" "$1" <<'PY' | tee "$R/gpurun_out/pmc_mlp_summary_$1.txt"
import csv, glob, sys, collections, re, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in glob.glob(os.path.join(sys.argv[1], "pmc_mlp_" + sys.argv[2] + "_*")):
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
    if "mlp" not in k: continue
    us = dur[k][0] / max(dur[k][1], 1) / 1e3
    print(f"{k}   mean duration under the profiler {us:.2f} us")
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:28s} {s / n:16.0f}")
    g = d.get("GRBM_GUI_ACTIVE")
    if g and us > 0:
        print(f"   -> clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {g[0] / g[1] / 8 / us / 1e3:.2f} GHz")
PY
for PASS in GRBM_GUI_ACTIVE SQ_INSTS_VALU; do rm -rf "$R/gpurun_out/pmc_mlp_$1_$PASS"; done
