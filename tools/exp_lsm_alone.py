#!/usr/bin/env python3
"""Is lsm_pass1_kernel slowed down by the generator's dirty lines?  Times the two-pass backward induction on a
RESIDENT matrix (no generator in front of it) against the fused pricing (generator -> pass 1 -> pass 2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from options_model_amd import _ffi

M, N = 1_000_000, 252
ctx = _ffi.Context(0)
S = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, seed=42)
for _ in range(5):
    ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")
ts = []
for _ in range(20):
    ts.append(ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")["ms_lsm"])
print("standalone two-pass LSM on a resident matrix: ms_lsm median %.3f min %.3f" % (np.median(ts), min(ts)))
for env in ("1", "2", "3"):
    pass
p = _ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42)
outs = ctx.price_american_seq([_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42, stream=i) for i in range(20)])
o = ctx.price_american(p)
print("fused pricing: ms_paths %.3f ms_pass1 %.3f ms_pass2 %.3f ms_lsm %.3f" % (o["ms_paths"], o["ms_pass1"], o["ms_pass2"], o["ms_lsm"]))
ctx.close()
