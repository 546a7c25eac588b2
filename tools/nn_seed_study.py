#!/usr/bin/env python3
"""Config 1 through the default pricer, seed by seed: price with dropout at inference (the reference's mode) and in eval
mode, epochs run, best loss -- for the library's trainer (16-row kernel; OMC_MLP_Q16=0 in the environment gives the 32-row
kernel) and, with --torch, PyTorch autograd on the same paths / rows / initial weights.
usage: nn_seed_study.py [--torch] seed [seed ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import RNGManager, nn_regressor as nnr

args = sys.argv[1:]
trainer = "torch" if args[:1] == ["--torch"] else "hip"
seeds = [int(s) for s in (args[1:] if trainer == "torch" else args)] or [42, 1, 2, 3, 4, 5, 6, 7]
dev = torch.device("cuda", 0)
for seed in seeds:
    rm = RNGManager(seed)
    path_seed, torch_seed = rm.get_child_seed(), rm.get_child_seed()  # options_model_3.py:454-455
    ctx = nnr._ctx_on_torch_stream(0)
    S = torch.empty((51, 10000), dtype=torch.float32, device=dev)
    nnr.generate_paths(ctx, S, dict(model="gbm"), 100.0, 0.05, 0.2, 1.0, path_seed)
    out = nnr.price_with_paths(S, 100.0, 0.05, 1.0, True, torch_seed, nn_hidden=128, nn_layers=3, nn_dropout=0.1, nn_epochs=25,
                               nn_lr=1e-3, trainer=trainer)
    ym = torch.tensor(out["Y_mean"], dtype=torch.float64, device=dev)
    ysd = torch.tensor(out["Y_std"], dtype=torch.float64, device=dev)
    a = (S, 100.0, 0.05, 1.0, True, out["net"], out["feat_mean"], out["feat_std"], ym, ysd)
    ev = nnr.pass2_fused(*a, dropout_on=False)
    on = [nnr.pass2_fused(*a, dropout_on=True, seed=1000 + k)["price"] for k in range(4)]
    print(json.dumps(dict(seed=seed, trainer=out["trainer"], price=round(out["price"], 4), eval_price=round(ev["price"], 4),
                          other_masks=[round(x, 4) for x in on], epochs=out["epochs_run"], steps=out["optimizer_steps"],
                          best_loss=round(out["best_loss"], 5), best_epoch=out.get("best_epoch"), final_lr=out.get("final_lr"),
                          european=round(float(((100.0 - S[50].double()).clamp(min=0)).mean() * torch.exp(torch.tensor(-0.05))), 4))),
          flush=True)
