#!/usr/bin/env python3
"""Fixtures for SURVEY row f-3: the calibrator's own European Heston Monte-Carlo
(options_model_3/heston_calibration.py:197-312), captured by running the real HestonPricer with a
recording RNG.  Build container only; writes tests/golden/calibrator.npz (numbers only)."""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("yfinance", types.ModuleType("yfinance"))
sys.path.insert(0, "/root/reference/options_model_3")
import heston_calibration as hc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "calibrator.npz")


class Rec:
    def __init__(self, gen):
        self.gen, self.log = gen, []

    def standard_normal(self, *a, **k):
        z = self.gen.standard_normal(*a, **k)
        self.log.append(z.copy())
        return z


def main():
    out = {}
    cfg = hc.CalibrationConfig(n_mc_paths=512, n_time_steps=40, use_antithetic=True, seed=42,
                               verbose=False, plot_results=False)
    for tag, sig in (("feller", 0.3), ("floor", 1.2)):
        prm = hc.HestonParams(kappa=2.0, theta=0.04, sigma=sig, rho=-0.7, v0=0.04)
        pr = hc.HestonPricer(cfg)
        pr.rng = Rec(np.random.default_rng(7))
        S, V = pr.simulate_paths(prm, 100.0, 0.75, 0.03)
        out[f"{tag}_z1"], out[f"{tag}_z2i"] = pr.rng.log
        out[f"{tag}_S"], out[f"{tag}_V"] = S, V
        out[f"{tag}_params"] = np.array([100.0, 0.03, 0.75, prm.v0, prm.kappa, prm.theta, prm.sigma, prm.rho])
        # one expiry, many strikes, on freshly drawn (recorded) normals
        pr.rng = Rec(np.random.default_rng(11))
        K = np.array([80.0, 90.0, 100.0, 110.0, 125.0])
        prices = pr.price_options_batch(prm, 100.0, K, np.full(5, 0.75), 0.03)
        out[f"{tag}_batch_z1"], out[f"{tag}_batch_z2i"] = pr.rng.log
        out[f"{tag}_batch_K"], out[f"{tag}_batch_prices"] = K, prices
        pr.rng = Rec(np.random.default_rng(11))
        out[f"{tag}_put100"] = np.float64(pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03, "put"))
        print(tag, prices, out[f"{tag}_put100"], "min V", V.min(), "min S", S.min())
    np.savez_compressed(OUT, **out)
    print(os.path.getsize(OUT))


if __name__ == "__main__":
    main()
