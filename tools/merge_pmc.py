#!/usr/bin/env python3
"""Merge the per-flow PMC summaries of one config (tools/summarize_pmc.py output, one file per flow) into
profiles/pmc_traffic_<config>.json, the file bench.py quotes `roofline.traffic` from.  Kernels of the K-pricings-per-launch
per-step flow are tagged with pricings_per_launch (bench.py matches on it).
usage: merge_pmc.py <config> <round> <paths_per_gpu> <summary.json> [<summary.json> ...] [--k 16]"""
import json, os, sys

args = sys.argv[1:]
k = 16
if "--k" in args:
    i = args.index("--k"); k = int(args[i + 1]); del args[i:i + 2]
cfg, rnd, ppg, files = args[0], args[1], int(args[2]), args[3:]
out = {"units": "bytes per dispatch; read = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024",
       "config": cfg, "paths_per_gpu": ppg, "round": rnd, "kernels": {}}
for f in files:
    j = json.load(open(f))
    for name, v in j["kernels"].items():
        if "_multi_" in name:
            v = dict(v, pricings_per_launch=k)
        if name in out["kernels"] and out["kernels"][name].get("dispatches", 0) >= v.get("dispatches", 0):
            continue  # (the path generator appears in every flow: keep the sample with more dispatches)
        out["kernels"][name] = v
    if j.get("storage") == "folded":
        out["storage"] = "two-pass flow on antithetic-folded storage (kernel names *_fold_kernel, gbm_paths_kernel<., false>)"
    for key in ("paths_kernel_bytes_per_launch", "lsm_two_pass_bytes_per_pricing", "lsm_step_kernel_bytes_per_launch"):
        if j.get(key) and not out.get(key):
            out[key] = j[key]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"pmc_traffic_{cfg}.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, len(out["kernels"]), "kernels")
