#!/usr/bin/env python3
"""Wall time of the per-step ContNet flow (omc_price_american_contnet) at the reference's UI sizes and at C1."""
import json
import sys
import time

sys.path.insert(0, ".")
from options_model_amd import _ffi

ctx = _ffi.default_context()
for M, N in ((10000, 50), (10000, 130), (100000, 50), (1000000, 50)):
    p = _ffi.make_params(is_put=True, n_paths=M, n_steps=N, seed=42)
    ctx.price_american_contnet(p, 32, 10, 1e-3, 1)
    t0 = time.perf_counter()
    reps = 3
    for i in range(reps):
        out = ctx.price_american_contnet(p, 32, 10, 1e-3, 1 + i)
    dt = (time.perf_counter() - t0) / reps
    poly = ctx.price_american(p)
    print(json.dumps(dict(M=M, N=N, ms=dt * 1e3, price=out["price"], poly=poly["price"], rows=out["sum_nitm"],
                          ms_per_step=dt * 1e3 / (N - 1))))
