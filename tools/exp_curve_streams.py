#!/usr/bin/env python3
"""Wall time of a v1 curve with the per-step network regressor by number of concurrent contexts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd.compat import Options_model as v1
for n in (1, 4, 8, 12, 16, 24):
    os.environ["OMC_CURVE_STREAMS"] = str(n)
    v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 10000, 2, 2 * n, "put", 2, False, 2025)  # contexts created, warm
    t0 = time.perf_counter()
    recs = v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 10000, 2, 180, "put", 2, False, 2025)
    dt = time.perf_counter() - t0
    print(f"{n:2d} contexts: 180-point curve (10k paths, 10..90 steps) {dt * 1e3:7.1f} ms", flush=True)
