#!/usr/bin/env python3
"""Wall time of the drop-in defaults: AdvancedOptionPricer (v3: 3x128 net, batch 256, <= 25 epochs) and the v1 / v2
pricers (per-step ContNet) at the reference's default call, 10k paths x 50 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import AdvancedOptionPricer, RNGManager
from options_model_amd.compat.Options_model import price_american_option as v1
from options_model_amd.compat.options_model_2 import OptionPricer as V2

p = AdvancedOptionPricer(100.0, 0.05, 0.2, "put", RNGManager(42))
p.price_american_enhanced_lsm(100.0, 1.0, 10000, 50)  # warm (library load, torch import for the scheduler)
for seed in (1, 2, 3):
    q = AdvancedOptionPricer(100.0, 0.05, 0.2, "put", RNGManager(seed))
    t0 = time.perf_counter(); price = q.price_american_enhanced_lsm(100.0, 1.0, 10000, 50); dt = time.perf_counter() - t0
    info = q.last_result
    print(f"v3 default: {dt:.3f} s  price {price:.4f}  epochs {info.get('epochs_run')} steps {info.get('optimizer_steps')} trainer {info.get('trainer')}", flush=True)
v1(100.0, 100.0, 1.0, 0.05, 0.2, 10000, 50, "put")
t0 = time.perf_counter(); m = v1(100.0, 100.0, 1.0, 0.05, 0.2, 10000, 50, "put"); dt = time.perf_counter() - t0
print(f"v1 default: {dt * 1e3:.1f} ms  price {m[0]:.4f}")
o = V2(100.0, 0.05, 0.2, "put")
o.price_american_option(100.0, 1.0)
t0 = time.perf_counter(); m = o.price_american_option(100.0, 1.0); dt = time.perf_counter() - t0
print(f"v2 default: {dt * 1e3:.1f} ms  price {m:.4f}")
