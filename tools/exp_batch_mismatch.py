#!/usr/bin/env python3
"""Which member of a random curve batch differs from its single call, and how (tests/test_gpu_fuzz.py curve-batch sweep).
usage: OMC_FUZZ_SCALE=60 OMC_FUZZ_SEED=12 exp_batch_mismatch.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from options_model_amd import _ffi
ctx = _ffi.Context(0)
hp = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)
bad = 0
for ci, case in enumerate(fz._batch_cases(10 * fz._SCALE, 3131 + fz._SHIFT)):
    ps = []
    for q in case["probs"]:
        kw = dict(model=case["model"], semantics=case["sem"], is_put=q["is_put"], n_paths=q["M"], n_steps=q["N"], S0=q["S0"], K=100.0,
                  r=q["r"], sigma=q["sigma"], T=q["T"], seed=q["seed"], stream=q["stream"])
        if case["model"] == "heston":
            kw.update(hp)
        ps.append(_ffi.make_params(**kw))
    for name, one, many in (("american", ctx.price_american, ctx.price_american_batch), ("european", ctx.price_european, ctx.price_european_batch)):
        bat = many(ps)
        for i, (p, b, q) in enumerate(zip(ps, bat, case["probs"])):
            a = one(p)
            a2 = one(p)
            keys = ("price", "sum", "sumsq", "n_exercised", "n_zero", "sum_nitm")
            if any(a[k] != b[k] for k in keys):
                bad += 1
                print(f"case {ci} {case['model']} {case['sem']} {name} member {i}/{len(ps)} M={q['M']} N={q['N']} put={q['is_put']} r={q['r']} "
                      f"Ms={[x['M'] for x in case['probs']]} single={[a[k] for k in keys]} batch={[b[k] for k in keys]} "
                      f"single_repeat_same={all(a[k] == a2[k] for k in keys)}", flush=True)
print("mismatching members:", bad)
