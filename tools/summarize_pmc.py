#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes.

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  Counters are collected in separate passes (TCC slots).

usage: summarize_pmc.py <gpurun_out dir> <tag> [config paths_per_gpu round]  -> prints a table and writes
       <dir>/pmc_traffic_<tag>.json (copy to profiles/pmc_traffic.json to have bench.py report it)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    base, tag = sys.argv[1], sys.argv[2]
    fetch = per_kernel(os.path.join(base, f"pmc_{tag}_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(base, f"pmc_{tag}_WRITE_SIZE"), "WRITE_SIZE")
    names = sorted(set(fetch) | set(write))
    out = {"units": "bytes per dispatch; read = 2 x FETCH_SIZE x 1024 (gfx950 correction), "
                    "write = WRITE_SIZE x 1024", "kernels": {}}
    if len(sys.argv) > 5:
        out.update(config=sys.argv[3], paths_per_gpu=int(sys.argv[4]), round=sys.argv[5])
    print(f"{'kernel':58s} {'calls':>6s} {'read MB':>10s} {'write MB':>10s}")
    for n in names:
        f, nf = fetch.get(n, (0.0, 0))
        w, nw = write.get(n, (0.0, 0))
        rd, wr = 2.0 * f * 1024.0, w * 1024.0
        out["kernels"][n] = {"read_bytes": rd, "write_bytes": wr, "dispatches": max(nf, nw)}
        print(f"{n[:58]:58s} {max(nf, nw):6d} {rd / 1e6:10.2f} {wr / 1e6:10.2f}")
    k = out["kernels"]

    def tot(prefixes):
        return sum(v["read_bytes"] + v["write_bytes"] for n, v in k.items()
                   if any(p in n for p in prefixes))

    out["paths_kernel_bytes_per_launch"] = tot(["gbm_paths_kernel", "heston_paths_kernel"]) or None
    out["lsm_two_pass_bytes_per_pricing"] = tot(["lsm_pass1_kernel", "lsm_pass1_fold_kernel", "lsm_reduce_pass1_kernel",
                                                 "lsm_solve_all_kernel", "lsm_pass2_kernel", "lsm_pass2_fold_kernel",
                                                 "lsm_finalize_kernel"]) or None
    if any("_fold_kernel" in n for n in k):
        out["storage"] = "folded"
    out["lsm_step_kernel_bytes_per_launch"] = tot(["lsm_step_ind_kernel", "lsm_step_kernel"]) or None
    json.dump(out, open(os.path.join(base, f"pmc_traffic_{tag}.json"), "w"), indent=1)
    print(json.dumps({a: b for a, b in out.items() if a != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
