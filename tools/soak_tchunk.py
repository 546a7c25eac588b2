#!/usr/bin/env python3
"""The pass-1 partition into time chunks must not change any number: price a grid of sizes with the automatic
choice and with OMC_PASS1_TCHUNK=32 (separate processes: the variable is read once) and compare bit for bit."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from options_model_amd import _ffi
ctx = _ffi.Context(0)
out = []
for M in (2, 1000, 65536, 250000, 333334, 500000, 786432, 1000000, 1500000, 3000000, 3145728, 3200000):
    for N in (2, 3, 40, 100, 252, 300):
        for model in ("gbm", "heston"):
            if model == "heston" and M > 1000000: continue
            o = ctx.price_american(_ffi.make_params(model=model, semantics="two_pass", n_paths=M, n_steps=N, seed=M %% 97 + N, is_put=(N %% 2 == 0)))
            out.append((M, N, model, o["sum"], o["sumsq"], o["n_exercised"], o["sum_nitm"]))
print(json.dumps(out))
''' % ROOT
res = {}
for tag, env in (("auto", {}), ("fixed32", {"OMC_PASS1_TCHUNK": "32"}), ("fixed126", {"OMC_PASS1_TCHUNK": "126"})):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=e, timeout=900)
    if p.returncode != 0:
        print(p.stderr[-2000:]); sys.exit(1)
    res[tag] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("[")][-1])
bad = 0
for a, b, c in zip(res["auto"], res["fixed32"], res["fixed126"]):
    if a != b or a != c:
        bad += 1
        print("MISMATCH", a, b, c)
print(f"{len(res['auto'])} pricings x 3 partitions: {bad} mismatches")
sys.exit(1 if bad else 0)
