#!/usr/bin/env python3
"""Time the fused trainer of the continuation-value network (omc_mlp_train_epoch) on synthetic rows:
usage: bench_mlp.py [rows] [batch[,batch..]] [epochs] [dropout[,dropout..]] [hidden layers 2|3] [hidden units 64|128] -> one JSON line per setting
(us per optimizer step, rows/s, TFLOP/s of the algorithmic MFMA work)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import _ffi, nn_regressor as nnr

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1 << 17]
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
drops = [float(d) for d in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0.1]
layers = int(sys.argv[5]) if len(sys.argv) > 5 else 2
hidden = int(sys.argv[6]) if len(sys.argv) > 6 else 64
dev = torch.device("cuda", 0)
ctx = _ffi.default_context(0)
data = torch.randn(rows, 8, device=dev)
net = nnr.make_net(7, hidden, layers, 0.1).to(dev)
flop_row = 2 * 8 * hidden * 2 + (layers - 1) * 3 * 2 * hidden * hidden + 2 * hidden  # L1 fwd + gW1, per connection fwd + dH + gW, output
for batch in batches:
    for drop in drops:
        p = nnr.flatten_params(net); m = torch.zeros_like(p); v = torch.zeros_like(p)
        torch.cuda.synchronize()
        args = (p.data_ptr(), m.data_ptr(), v.data_ptr())
        loss, step = ctx.mlp_train_epoch(data.data_ptr(), min(rows, 4 * batch), batch, *args, 0, 1e-3, drop, 1, shuffle_key=7, layers=layers, hidden=hidden)
        t0 = time.perf_counter()
        for e in range(epochs):
            loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, batch, *args, step, 1e-3, drop, 1, shuffle_key=8 + e, layers=layers, hidden=hidden)
        dt = time.perf_counter() - t0
        nsteps = epochs * ((rows + batch - 1) // batch)
        print(json.dumps(dict(rows=rows, batch=batch, hidden=hidden, layers=layers, dropout=drop, steps=nsteps, us_per_step=1e6 * dt / nsteps,
                              rows_per_s=epochs * rows / dt, tflops=epochs * rows * flop_row / dt / 1e12, loss=loss)),
              flush=True)
