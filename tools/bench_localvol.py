#!/usr/bin/env python3
"""Time local-vol path generation through the IV network (row f-4): library kernel vs PyTorch-ROCm.
usage: bench_localvol.py [paths] [steps]"""
import json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import local_vol

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 252
torch.manual_seed(0)
net = local_vol.make_iv_network(64, 4)
net.scaler = types.SimpleNamespace(m_scale=0.3, tau_scale=1.0)
model = local_vol.IVModel(net)
out = {}
for backend in ("hip", "torch"):
    local_vol.simulate_local_vol_paths(100.0, 0.05, 1.0, 2048, 4, model, 100.0, seed=1, backend=backend)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S = local_vol.simulate_local_vol_paths(100.0, 0.05, 1.0, M, N, model, 100.0, seed=2, backend=backend)
    torch.cuda.synchronize()
    out[backend] = time.perf_counter() - t0
    out[backend + "_mean_ST"] = float(S[-1].double().mean())
    del S
flop = 2 * (2 * 64 + 4 * 64 * 64 + 64)
out["hip_tflops"] = M * N * flop / out["hip"] / 1e12
out["path_steps_per_s_hip"] = M * N / out["hip"]
print(json.dumps(dict(paths=M, steps=N, **out)))
