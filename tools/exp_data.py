#!/usr/bin/env python3
"""Does pass 1's time depend on the DATA (bit toggling / mask patterns), not only on the instruction stream?
Resident matrix, two-pass LSM, for real GBM paths and for degenerate ones (sigma -> 0: every path identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from options_model_amd import _ffi
M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 252
ctx = _ffi.Context(0)
for tag, S0, sigma in (("real paths, at the money", 100.0, 0.2), ("identical paths, all in the money", 90.0, 1e-7),
                       ("identical paths, all out of the money", 110.0, 1e-7), ("real paths again", 100.0, 0.2)):
    S = ctx.gbm_paths(M, N, S0, 0.05, sigma, 1.0, seed=42)
    for _ in range(5):
        ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")
    ts = [ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")["ms_lsm"] for _ in range(30)]
    print(f"M={M} {tag:40s} two-pass LSM on the resident matrix: median {np.median(ts):.4f} min {min(ts):.4f} ms", flush=True)
    S.free()
