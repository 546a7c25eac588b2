#!/usr/bin/env python3
"""One ContNet pricing (10k paths x 50 steps, the UI's point size): the single-call entry point against the batched
entry point with n = 1, 2, 4 (no host read per time step there)."""
import json
import sys
import time

sys.path.insert(0, ".")
from options_model_amd import _ffi

ctx = _ffi.default_context()
for M, N in ((10000, 50), (10000, 130), (100000, 50)):
    p = _ffi.make_params(is_put=True, n_paths=M, n_steps=N, seed=42)
    rec = dict(M=M, N=N)
    ctx.price_american_contnet(p, 32, 10, 1e-3, 1)
    t0 = time.perf_counter()
    for i in range(5):
        one = ctx.price_american_contnet(p, 32, 10, 1e-3, 7)
    rec["single_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    for n in (1, 2, 4):
        ctx.price_american_contnet_batch([p] * n, 32, 10, 1e-3, 7)
        t0 = time.perf_counter()
        for i in range(5):
            out = ctx.price_american_contnet_batch([p] * n, 32, 10, 1e-3, 7)
        rec[f"batch{n}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        rec[f"batch{n}_same"] = all(o["price"] == one["price"] for o in out)
    print(json.dumps(rec), flush=True)
