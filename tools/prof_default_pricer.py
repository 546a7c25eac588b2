#!/usr/bin/env python3
"""The reference's default call under a profiler: AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, 'put') .price_american_enhanced_lsm
(100, 1, 10000, 50) with reference arguments (3 x 128 net, batch 256, <= 25 epochs, dropout 0.1 on), once warm, once timed."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import AdvancedOptionPricer, RNGManager

AdvancedOptionPricer(100.0, 0.05, 0.2, "put", RNGManager(7)).price_american_enhanced_lsm(100.0, 1.0, 10000, 50)
q = AdvancedOptionPricer(100.0, 0.05, 0.2, "put", RNGManager(42))
t0 = time.perf_counter(); price = q.price_american_enhanced_lsm(100.0, 1.0, 10000, 50); dt = time.perf_counter() - t0
info = q.last_result
print(json.dumps(dict(seconds=dt, price=price, **{k: info.get(k) for k in ("epochs_run", "optimizer_steps", "trainer", "batch")},
                      timings_ms={k[8:]: round(1e3 * v, 3) for k, v in q.last_result.items() if k.startswith("seconds_")})))
