#!/usr/bin/env python3
"""Fixtures for SURVEY row f-4: local-vol paths through the IV network
(options_model_3.py:263-333 + NN_training_stock_iv.py:109-155), captured by running the real
simulate_local_vol_paths_antithetic with a seeded random-init ImprovedIVNetwork (eval mode) and a
recording RNG.  A trained net would need market data from the network; parity of the SIMULATOR only
needs *a* net.  Build container only; writes tests/golden/localvol.npz (numbers only)."""
import os
import sys
import types

import numpy as np
import torch

sys.modules.setdefault("yfinance", types.ModuleType("yfinance"))
sys.path.insert(0, "/root/reference/options_model_3")
import NN_training_stock_iv as iv  # noqa: E402
import options_model_3 as om  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "localvol.npz")


class Rec:
    def __init__(self, gen):
        self.gen, self.log = gen, []

    def standard_normal(self, *a, **k):
        z = self.gen.standard_normal(*a, **k)
        self.log.append(z.copy())
        return z


def main():
    torch.manual_seed(123)
    cfg = iv.TrainingConfig()
    net = iv.ImprovedIVNetwork(cfg)
    # make the surface interesting: bias the output so sigma ~ 0.2 +- smile instead of the 1e-4 floor
    with torch.no_grad():
        net.output.bias.fill_(0.22)
        net.output.weight.mul_(0.35)
    sc = iv.DataScaler()
    sc.m_scale, sc.tau_scale, sc.m_mean, sc.tau_mean, sc.S0 = 0.15, 0.4, 0.01, 0.5, 100.0
    net.scaler = sc
    model = om.IVModel(net)
    out = {}
    rng = Rec(np.random.default_rng(5))
    S0, r, T, K, M, N = 100.0, 0.05, 0.75, 105.0, 256, 24
    S = om.simulate_local_vol_paths_antithetic(S0, r, T, M, N, model, K, rng)
    out["z_half"] = rng.log[0]
    out["S"] = S
    out["params"] = np.array([S0, r, T, K, sc.m_scale, sc.tau_scale, cfg.epsilon])
    sig = model.get_volatility_batch(K, np.linspace(60, 160, 41), 0.3)
    out["vol_S"], out["vol_out"] = np.linspace(60, 160, 41), sig
    for k, v in net.state_dict().items():
        out["sd_" + k] = v.numpy()
    out["arch"] = np.array([cfg.hidden_dim, cfg.num_hidden_layers])
    # American put on those paths through the reference's default route is NN-LSM (slow, stochastic);
    # the LSM side is pinned elsewhere.  Record the European value for an end-to-end anchor.
    out["european_put"] = np.float64(np.exp(-r * T) * np.maximum(K - S[-1], 0).mean())
    np.savez_compressed(OUT, **out)
    print("sigma range", sig.min(), sig.max(), "S_T mean", S[-1].mean(), os.path.getsize(OUT))


if __name__ == "__main__":
    main()
