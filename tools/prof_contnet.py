"""Workload for rocprofv3: the v1 / v2 default regressor (fresh ContNet per step). argv[1]: "one" = 20 x the UI's single
pricing (10k x 50), "job" = the UI's 1,620-point job once."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi
ctx = _ffi.Context(0)
what = sys.argv[1] if len(sys.argv) > 1 else "one"
if what == "one":
    one = [_ffi.make_params(semantics="reference", n_paths=10_000, n_steps=50, seed=42)]
    ctx.price_american_contnet_batch(one, 32, 10, 1e-3, 42)
    t0 = time.perf_counter()
    for _ in range(20):
        o = ctx.price_american_contnet_batch(one, 32, 10, 1e-3, 42)
    print("ms per pricing", (time.perf_counter() - t0) / 20 * 1e3, "gpu ms", o[0]["ms_total"], "price", o[0]["price"])
else:
    ps = []
    for s0 in (80, 85, 90, 95, 100, 105, 110, 115, 120):
        for i in range(180, 0, -1):
            d = i / 2.0
            ps.append(_ffi.make_params(semantics="reference", n_paths=10_000, n_steps=max(10, min(130, int(math.ceil(d)))),
                                       S0=float(s0), T=d / 365.0, seed=42))
    ctx.price_american_contnet_batch(ps[:8], 32, 10, 1e-3, 42)
    t0 = time.perf_counter()
    o = ctx.price_american_contnet_batch(ps, 32, 10, 1e-3, 42)
    print("job s", time.perf_counter() - t0, "gpu ms", o[0]["ms_total"])
ctx.close()
