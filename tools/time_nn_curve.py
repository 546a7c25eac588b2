#!/usr/bin/env python3
"""Wall time of v3 NN curves (compute_curve_for_S0, options_model_3.py:697-713: default 3 x 128 net per point, 10k paths)
through the side-by-side trainer, by number of points.  Point i of a curve expires in i / IPD days and has
max(10, min(130, ceil(days))) time steps, so the networks of a long curve differ in size (minibatch 256 up to 262,144
regression rows, 512 / 1024 beyond: nn_regressor.pick_batch).
usage: [TIME_NN_IPD=1] time_nn_curve.py [points ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import AdvancedOptionPricer, RNGManager

IPD = int(os.environ.get("TIME_NN_IPD", "1"))
mk = lambda: AdvancedOptionPricer(K=100.0, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(3), use_control_variate=False)
mk().compute_curve_for_S0(100.0, 1, 2, 10_000, False)
for n in [int(a) for a in sys.argv[1:]] or [40, 128, 256]:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = mk()
    recs = p.compute_curve_for_S0(100.0, IPD, n, 10_000, False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} points ({IPD}/day): {dt:.2f} s = {1e3 * dt / n:.1f} ms per point; first {recs[0]['Option Value']:.4f} "
          f"last {recs[-1]['Option Value']:.4f}", flush=True)
