#!/usr/bin/env python3
"""Generator kernel time by pairs-per-thread (option gbm_vec / heston_vec) at the bench sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from options_model_amd import _ffi
ctx = _ffi.Context(0)
for model, M in (("gbm", 1_000_000), ("gbm", 8_000_000), ("heston", 4_000_000)):
    for vec in (4, 2, 1):
        ctx.set_option("gbm_vec" if model == "gbm" else "heston_vec", vec)
        ps = [_ffi.make_params(model=model, semantics="two_pass", n_paths=M, n_steps=252, seed=42, stream=i) for i in range(12)]
        ctx.price_american_seq(ps)
        ts = []
        for rep in range(5):
            outs = ctx.price_american_seq(ps)
            ts.append(outs[0]["ms_paths"])
        print(model, M, "vec", vec, "ms_paths median %.4f" % np.median(ts), "whole %.4f" % outs[0]["ms_total"], flush=True)
ctx.close()
