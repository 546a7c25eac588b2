#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection.csv) + kernel durations (kernel_trace.csv).

usage: summarize_sq.py <gpurun_out dir> <directory prefix> [--by-dir]
  every directory <dir>/<prefix>* is one pass; counters are averaged per dispatch and kernel.  --by-dir keeps the
  passes apart (one table per directory group: the text between the prefix and the last "_x" suffix names the group)
  and adds exact read bytes where the request-size counters are present: 32 n32 + 64 n64 + 128 n128.
Derived lines (when the counters are there): clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; VALU busy = 4 x
SQ_ACTIVE_INST_VALU / (1,024 SIMDs x cycles); wave-life shares of SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES etc.
"""
import collections
import csv
import glob
import os
import re
import sys


def kname(full):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", full)
    return m.group(1) if m else full[:48]


def collect(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[kname(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = kname(r["Kernel_Name"])
                dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
    return acc, dur


def show(acc, dur, only=None):
    for k in sorted(acc, key=lambda x: -dur[x][0]):
        if only and not any(o in k for o in only):
            continue
        c = {n: v[0] / v[1] for n, v in acc[k].items()}
        us = dur[k][0] / max(dur[k][1], 1) / 1e3
        print(f"{k}: {us:.1f} us per dispatch (under the counters), {max(v[1] for v in acc[k].values())} dispatches per counter")
        for n in sorted(c):
            print(f"   {n:28s} {c[n]:16.0f}")
        if "GRBM_GUI_ACTIVE" in c and us > 0:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            print(f"   -> {cyc:.0f} cycles per XCD = {cyc / us / 1e3:.2f} GHz over the dispatch")
            if "SQ_ACTIVE_INST_VALU" in c:
                print(f"   -> VALU issuing {4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.1%} of the cycles of the 1,024 SIMDs")
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            for n in sorted(c):
                if n.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_INST_CYCLES")):
                    print(f"   -> {n} / SQ_WAVE_CYCLES = {c[n] / w:.3f}  (x4: {4 * c[n] / w:.3f})")
        if all(n in c for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
            n32, n64, n128, tot = c["TCC_EA0_RDREQ_32B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_128B_sum"], c["TCC_EA0_RDREQ_sum"]
            print(f"   -> exact read bytes 32 n32 + 64 n64 + 128 n128 = {(32 * n32 + 64 * n64 + 128 * n128) / 1e6:.2f} MB "
                  f"(requests: {n32:.0f} x 32 B, {n64:.0f} x 64 B, {n128:.0f} x 128 B; all {tot:.0f})")
        if "FETCH_SIZE" in c:
            print(f"   -> 2 x FETCH_SIZE x 1024 = {2 * c['FETCH_SIZE'] * 1024 / 1e6:.2f} MB")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            print(f"   -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.1%}")


def main():
    base, prefix = sys.argv[1], sys.argv[2]
    by_dir = "--by-dir" in sys.argv
    dirs = sorted(d for d in glob.glob(os.path.join(base, prefix + "*")) if os.path.isdir(d))
    if not dirs:
        print("no passes found under", base, prefix)
        return
    if not by_dir:
        show(*collect(dirs))
        return
    groups = collections.defaultdict(list)
    for d in dirs:
        g = os.path.basename(d)[len(prefix):]
        groups[g.rsplit("_", 1)[0]].append(d)
    for g in sorted(groups):
        print(f"==== {prefix}{g}")
        show(*collect(groups[g]), only=("lsm_pass1_kernel", "lsm_pass2_kernel", "gbm_paths_kernel"))


if __name__ == "__main__":
    main()
