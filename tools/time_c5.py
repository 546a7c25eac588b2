#!/usr/bin/env python3
"""BASELINE config 5 (GBM put, 1M x 252, NN 2 x 64) through the drop-in call, one JSON line: stage times, the
trainer's algorithmic TFLOP/s against the float32 MFMA peak.  Under `rocprofv3 --kernel-trace --stats` the same run
gives the per-kernel averages kept in profiles/.  usage: time_c5.py [paths] [steps] [epochs]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import nn_regressor as nnr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 252
E = int(sys.argv[3]) if len(sys.argv) > 3 else 25
nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2)  # warm
torch.cuda.synchronize()
t0 = time.perf_counter()
o = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, M, N, seed=42, nn_epochs=E)
dt = time.perf_counter() - t0
flop = 2 * (8 * 64 + 64 * 64 + 64) + 2 * 2 * 64 * 64 + 2 * 8 * 64  # per row: forward, dH1, gW2, gW1
tk = o.timings_ms.get("train_kernels", 0.0) * 1e-3
rows_seen = o.sum_nitm * o.info.get("epochs_run", 0)
print(json.dumps(dict(config="c5", paths=M, steps=N, seconds=dt, path_steps_per_s=M * N / dt, price=o.price, stderr=o.stderr,
                      rows=o.sum_nitm, info=o.info, timings_ms={k: round(v, 3) for k, v in o.timings_ms.items()},
                      train_mfma=dict(bound="mfma", achieved=rows_seen * flop / tk / 1e12 if tk else None, peak=157.3,
                                      unit="TFLOP/s", frac=rows_seen * flop / tk / 1e12 / 157.3 if tk else None,
                                      flop_per_row=flop, rows_trained=rows_seen, kernel_seconds=tk))))
