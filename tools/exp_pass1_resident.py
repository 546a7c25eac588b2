#!/usr/bin/env python3
"""Workload for the pass-1 counter passes: the two-pass backward induction on a RESIDENT path matrix (no generator in
front of it), 12 times.  usage: exp_pass1_resident.py [paths]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi
M, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 252
ctx = _ffi.Context(0)
S = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, seed=42)
for _ in range(12):
    o = ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")
print("resident", M, o["ms_lsm"], o["price"])
ctx.close()
