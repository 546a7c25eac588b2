"""Time of a pricing with regressor "ols7" (omc_price_american_ols7) and of its LSM part alone.  usage: time_ols7.py [M N ...]
(default 10,000 x 50 and 1,000,000 x 252; 8,000,000 x 252 is one rank's shard of config 3: an 8 GB matrix, 9e8 rows)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import price_american_option, _ffi
ctx = _ffi.default_context(0)
a = [int(v) for v in sys.argv[1:]]
for M, N in (list(zip(a[0::2], a[1::2])) or [(10_000, 50), (1_000_000, 252)]):
    price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, M, N, regressor="ols7", seed=1, ctx=ctx)
    t0 = time.perf_counter()
    for i in range(5):
        r = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, M, N, regressor="ols7", seed=42 + i, ctx=ctx)
    dt = (time.perf_counter() - t0) / 5
    S = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, 42, 0)
    ctx.lsm_ols7(S, 100.0, 0.05, 1.0, True)
    t0 = time.perf_counter()
    for i in range(5):
        ctx.lsm_ols7(S, 100.0, 0.05, 1.0, True)
    dl = (time.perf_counter() - t0) / 5
    S.free()
    p = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, M, N, regressor="poly", seed=46, ctx=ctx)
    print(f"{M} x {N}: ols7 {1e3 * dt:.3f} ms per pricing (LSM part alone {1e3 * dl:.3f} ms), price {r.price:.4f} +- {r.stderr:.4f}, rows {r.sum_nitm}; poly two-pass {p.price:.4f}")
