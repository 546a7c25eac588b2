#!/bin/bash
# what does lsm_pass1_kernel see on the memory side when it runs (A) directly behind the path
# generator (the pricing: bench.py --only-timed) and (B) on a resident matrix (tools/exp_pass1_resident.py)?
# Four --pmc passes per workload (4 TCC slots per pass), program directly after `--`.
# usage: pmc_pass1.sh c2|c3
set -u
R="$GRAFT_REPO_ROOT"; CFG=${1:-c2}
PATHS=1000000; [ "$CFG" = c3 ] && PATHS=8000000
cd /tmp && export TMPDIR=/tmp
OUT="$R/gpurun_out/pmc_pass1_$CFG"; mkdir -p "$OUT"
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_TAG_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_STREAMING_REQ_sum"
 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/behind_$i" -- \
    python3 "$R/bench.py" --config $CFG --only-timed --steps 6 --warmup 2 --min-warmup-seconds 0.05 > /dev/null 2> "$OUT/behind_$i.err" || { echo "pass $i (behind) failed"; tail -3 "$OUT/behind_$i.err"; }
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/resident_$i" -- \
    python3 "$R/tools/exp_pass1_resident.py" $PATHS > /dev/null 2> "$OUT/resident_$i.err" || { echo "pass $i (resident) failed"; tail -3 "$OUT/resident_$i.err"; }
done
python3 - "$OUT" "$CFG" <<'PY' | tee "$R/gpurun_out/pmc_pass1_${CFG}_summary.txt"
import csv, glob, sys, collections, os
base, cfg = sys.argv[1], sys.argv[2]
def collect(kind):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in glob.glob(os.path.join(base, kind + "_*")):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("omc::", "")
                a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
A, B = collect("behind"), collect("resident")
print(f"# {cfg}: memory-side counters per dispatch (mean over the dispatches of the run); 'behind' = in the pricing, right")
print("# after the path generator / pass 1; 'resident' = the same kernel on a matrix generated long before")
for kern in sorted(set(A) | set(B)):
    if not any(s in kern for s in ("lsm_pass1_kernel", "lsm_pass2_kernel", "gbm_paths_kernel")): continue
    print(kern)
    names = sorted(set(A.get(kern, {})) | set(B.get(kern, {})))
    for c in names:
        a = A.get(kern, {}).get(c); b = B.get(kern, {}).get(c)
        fa = f"{a[0]/a[1]:16.0f} (n={a[1]})" if a else " " * 16 + "-"
        fb = f"{b[0]/b[1]:16.0f} (n={b[1]})" if b else " " * 16 + "-"
        print(f"   {c:40s} behind {fa}   resident {fb}")
PY
rm -rf "$OUT"/behind_*/ "$OUT"/resident_*/ 2>/dev/null
