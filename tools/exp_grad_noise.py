#!/usr/bin/env python3
"""Which side carries the error when a trainer kernel's gradient and PyTorch's float32 autograd differ by more than
2e-5 of the largest component?  Both against autograd in float64 under the same masks.
usage: exp_grad_noise.py hidden layers rows p [step seed]"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from options_model_amd import nn_regressor as nnr
from oracle import dropout as dr
import test_gpu_dropout as td

hidden, layers, rows, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = torch.device("cuda", 0)
ctx = nnr._ctx_on_torch_stream(0)
variant = ctx.lib.omc_mlp_train_variant(hidden, layers, rows)
for step, seed in [(1, 5), (999, 77), (21_999, 123456789), (0, 2 ** 61 + 5), (10 ** 6, 42), (7, 7)]:
    torch.manual_seed(3)
    net = nnr.make_net(7, hidden, layers, p).to(dev)
    data = td._data(torch, dev, rows, 11)
    masks = dr.train_masks(variant, hidden, layers, np.arange(rows), step + 1, seed, p)
    net.zero_grad(set_to_none=True)
    td._masked_loss(torch, net, data, masks, p).backward()
    g32 = td._flat_grads(torch, nnr, net).cpu().numpy().astype(np.float64)
    net64 = copy.deepcopy(net).double()
    net64.zero_grad(set_to_none=True)
    lin = [m for m in net64.net if isinstance(m, torch.nn.Linear)]
    h = data[:, :7].double()
    inv = float(dr.inv_keep_of(p))
    for j, l_ in enumerate(lin[:-1]):
        h = torch.relu(l_(h)) * (torch.from_numpy(masks[j]).to(dev).double() * inv)
    ((lin[-1](h) - data[:, 7:].double()) ** 2).sum().div(rows).backward()
    g64 = td._flat_grads(torch, nnr, net64).cpu().numpy()
    p0 = nnr.flatten_params(net)
    pp, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    torch.cuda.synchronize()
    ctx.mlp_train_epoch(data.data_ptr(), rows, rows, pp.data_ptr(), m.data_ptr(), v.data_ptr(), step, 1e-3, p, seed,
                        weight_decay=0.0, hidden=hidden, layers=layers)
    g = (m / 0.1).cpu().numpy().astype(np.float64)
    sc = np.abs(g64).max()
    print(f"variant {variant} {hidden}x{layers} rows {rows} p {p} step {step}: kernel vs f64 {np.abs(g - g64).max() / sc:.2e}   "
          f"torch f32 vs f64 {np.abs(g32 - g64).max() / sc:.2e}   kernel vs torch f32 {np.abs(g - g32).max() / sc:.2e}", flush=True)
