#!/usr/bin/env python3
"""Pass 1 of the NN flow alone (omc_nn_build_rows: count + statistics sweep, scans, merges, row write) on a GBM path
matrix: ms per build, rows, normalisers.  usage: time_rows.py [paths] [steps] [repeats]   (default: config 5's 1M x 252)
For rocprofv3 / --pmc passes of the rows_* kernels (tools/gpu_r06.sh rows)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from options_model_amd import nn_regressor as nnr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 252
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
ctx = nnr._ctx_on_torch_stream(0)
S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
nnr.generate_paths(ctx, S, dict(model="gbm"), 100.0, 0.05, 0.2, 1.0, 42)
out = nnr.build_rows_fused(S, 100.0, 0.05, 1.0, True)  # warm (allocator, code objects)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    del out
    t0 = time.perf_counter()
    out = nnr.build_rows_fused(S, 100.0, 0.05, 1.0, True)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
data, fm, fs, ym, ysd = out
print(json.dumps(dict(paths=M, steps=N, rows=int(data.shape[0]), ms_per_build=sorted(ts)[len(ts) // 2], ms_all=ts,
                      feat_mean=[float(x) for x in fm], feat_std=[float(x) for x in fs], y_mean=float(ym), y_std=float(ysd))))
