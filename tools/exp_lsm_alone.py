#!/usr/bin/env python3
"""Is lsm_pass1_kernel slowed down by the generator's dirty lines?  Times the two-pass backward induction on a
RESIDENT matrix (no generator in front of it) against the fused pricing (generator -> pass 1 -> pass 2).
usage: exp_lsm_alone.py [paths]   (OMC_PASS1_DIAG=1|2|3 selects the measurement builds of pass 1; they exist only in a library built with OMC_HIPCC_FLAGS=-DOMC_DIAG_BUILD)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from options_model_amd import _ffi

M, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 252
ctx = _ffi.Context(0)
S = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, seed=42)
for _ in range(5):
    ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")
ts = []
for _ in range(20):
    ts.append(ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")["ms_lsm"])
print("M=%d diag=%s  resident matrix: two-pass LSM ms median %.3f min %.3f" % (M, os.environ.get("OMC_PASS1_DIAG", "-"), np.median(ts), min(ts)), flush=True)
p = _ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42)
ctx.price_american_seq([_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42, stream=i) for i in range(10)])
o = ctx.price_american(p)
print("   fused pricing: ms_paths %.3f ms_pass1 %.3f ms_pass2 %.3f ms_lsm %.3f" % (o["ms_paths"], o["ms_pass1"], o["ms_pass2"], o["ms_lsm"]))
ctx.close()
