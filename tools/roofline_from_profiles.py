#!/usr/bin/env python3
"""Recompute every roofline fraction of DESIGN.md section 6.2 from the files under profiles/ alone:
    fraction = algorithmic bytes (or FLOP) per launch  /  rocprofv3's average launch duration  /  peak
The algorithmic figures are DESIGN.md section 3's (restated in ALGO below); the durations are the AverageNs column of
profiles/<round>_<config>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench.py command of that config);
peaks from /opt/skills/guides/MI355X_MICROARCH.md (HBM 8 TB/s; float32 MFMA 157.3 TFLOP/s).  Printed beside it: the
fraction the bench line of the same config carries (HIP events inside bench.py's timed region) and the PMC traffic of
profiles/pmc_traffic_<config>.json -- the three must tell the same story.
usage: roofline_from_profiles.py [round, default r05]"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
HBM, MFMA32 = 8000.0, 157.3

# config -> (M paths per GPU, N steps)
SIZES = {"two_pass": (1_000_000, 252), "reference": (1_000_000, 252), "c3": (8_000_000, 252), "c4": (4_000_000, 252)}


def algo(kernel, M, N, K=16):
    """algorithmic bytes per launch."""
    if "gbm_paths_kernel" in kernel and ", false>" in kernel:  # folded storage: first partners only (every config here is antithetic)
        return 2.0 * (N + 1) * M, "4 (N+1) M/2 written (folded)"
    if "paths_kernel" in kernel:
        return 4.0 * (N + 1) * M, "4 (N+1) M written"
    if "lsm_pass1_fold_kernel" in kernel:
        return 2.0 * N * M, "4 N M/2 read (folded)"
    if "lsm_pass2_fold_kernel" in kernel:
        return 2.0 * N * M, "<= 4 N M/2 read (folded)"
    if "lsm_pass1_kernel" in kernel:
        return 4.0 * N * M, "4 N M read"
    if "lsm_pass2_kernel" in kernel:
        return 4.0 * N * M, "<= 4 N M read"
    if "lsm_step_multi_kernel" in kernel:
        return 12.0 * M * K, f"12 M per pricing x {K} pricings"
    return None, None


def stats(name):
    f = os.path.join(P, f"{rnd}_{name}_kernel_stats.csv")
    if not os.path.exists(f):
        return None
    return {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(f))}


def bench_line(name):
    for cand in (f"bench_{rnd}_{name}.json", f"{rnd}_{name}_under_rocprof.json"):
        f = os.path.join(P, cand)
        if os.path.exists(f):
            for ln in open(f):
                if ln.startswith("{"):
                    return json.loads(ln)
    return None


def pmc(cfg):
    f = os.path.join(P, f"pmc_traffic_{cfg}.json")
    return json.load(open(f)) if os.path.exists(f) else None


print(f"{'config':10s} {'kernel':34s} {'avg us':>9s} {'calls':>6s} {'algorithmic':>14s} {'achieved':>12s} {'frac':>6s} {'bench frac':>10s} {'PMC bytes / algorithmic':>24s}")
for name, (M, N) in SIZES.items():
    st = stats(name)
    if st is None:
        print(f"{name:10s} (no profiles/{rnd}_{name}_kernel_stats.csv)")
        continue
    cfg = "c2" if name in ("two_pass", "reference") else name
    line = bench_line("driver" if name == "two_pass" else name) or bench_line(name)
    pj = pmc(cfg)
    for kname, (avg_ns, calls) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        b, why = algo(kname, M, N)
        if b is None or calls <= 2:  # (one or two launches: the line's price check on a 200k-path slice, not this workload)
            continue
        if "lsm_step_multi_kernel" in kname and name != "reference":
            continue  # (launched there by bench.py's per-step extra at other pricings-per-launch: see the `reference` rows)
        gbs = b / avg_ns  # bytes per ns = GB/s
        short = kname.split("(")[0].replace("void ", "").replace("omc::", "").replace("(anonymous namespace)::", "")
        bf = ""
        if line:
            for k in line.get("roofline_kernels", []) + [line.get("roofline_per_step", {}).get("k16", {})]:
                stem = (k.get("kernel") or "").replace("gbm_", "").replace("heston_", "")
                folded_line = str((line.get("config") or {}).get("storage", "")).startswith("antithetic-folded")
                if "gbm_paths_kernel" in short and folded_line != (", false>" in short):
                    continue  # (the other storage's generator: launched by this command's per-step extra)
                if stem and stem in short:
                    bf = f"{k.get('frac', 0):.3f}"
        tr = ""
        if pj and pj.get("round") == rnd:
            for pn, v in pj["kernels"].items():
                if pn.replace("omc::", "") == short:
                    tr = f"{(v['read_bytes'] + v['write_bytes']) / 1e6:9.1f} MB = {(v['read_bytes'] + v['write_bytes']) / b:5.3f} x"
        elif pj:
            tr = f"(pmc file is {pj.get('round')})"
        print(f"{name:10s} {short[:34]:34s} {avg_ns / 1e3:9.2f} {calls:6d} {b / 1e6:11.1f} MB {gbs:9.0f} GB/s {gbs / HBM:6.3f} {bf:>10s} {tr:>24s}   [{why}]")

# the two MFMA-bound lines
for name, pat, flop_row, rows_key in (("c5", "mlp_train_kernel<2>", 26752, "batch"), ("c1nn", "mlp_train_q16_kernel<128, 3>", 200960, "batch")):
    st, line = stats(name), bench_line(name)
    if st is None or line is None:
        print(f"{name:10s} (no kernel stats / bench line under profiles/ for {rnd})")
        continue
    batch = (line.get("info") or {}).get("batch")
    for kname, (avg_ns, calls) in st.items():
        if pat in kname and batch:
            tf = flop_row * batch / avg_ns / 1e3
            print(f"{name:10s} {pat:34s} {avg_ns / 1e3:9.2f} {calls:6d} {flop_row * batch / 1e9:10.3f} GFLOP {tf:8.2f} TFLOP/s {tf / MFMA32:6.3f} "
                  f"{(line.get('roofline') or {}).get('frac') or 0:10.3f}   [{flop_row} FLOP per row x minibatch {batch}; bench: whole optimizer steps incl. Adam]")
