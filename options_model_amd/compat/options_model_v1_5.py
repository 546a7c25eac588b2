"""v1.5 class surface (reference `options_model_v1.5.py:91-227, 301-304`) on the GPU hot path.

Same per-step flow and regressor as v1 (fresh ContNet(1-32-32-1) per step, 10 Adam steps; `regressor="poly"` /
OMC_REGRESSOR=poly for the polynomial).  What is particular to this file and kept:
  * ONE generator per pricer, never reseeded (`self.rng = np.random.default_rng(seed)`, :99-100): consecutive
    pricings of a pricer consume consecutive draws -> here pricing number k of a pricer uses Philox stream k of
    `seed` (and net seed `seed + k`), so a curve's points are independent and a fresh pricer repeats them;
  * the curve's step rule `steps = max(2, min(500, ceil(d * intervals_per_day)))` (:220);
  * `compute_curve_worker` lets exceptions propagate (:301-304).
The reference prints a summary per pricing (:191-210); here only when `verbose=True`.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional

from .. import _ffi
from ._regressor import resolve

NN_HIDDEN, NN_EPOCHS, NN_LR = 32, 10, 1e-3  # options_model_v1.5.py:60,158,166


class OptionPricer:
    def __init__(self, K, r, sigma, option_type="call", lsm_poly_degree=2, seed=42, *,
                 regressor: Optional[str] = None, verbose: bool = False):
        self.K, self.r, self.sigma, self.option_type = K, r, sigma, option_type
        self.lsm_poly_degree, self.seed = lsm_poly_degree, seed
        self.regressor, self.verbose = resolve(regressor), verbose
        self._pricings = 0
        self.last_result: Optional[dict] = None

    def _check(self, S0, T, num_simulations, num_time_steps):
        if S0 <= 0 or self.K <= 0 or T <= 0 or self.sigma <= 0:
            raise ValueError("S0, K, T, and sigma must be positive.")
        if self.r < 0:
            raise ValueError("r must be non-negative.")
        if num_simulations <= 0 or num_time_steps <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        if not isinstance(self.lsm_poly_degree, int) or self.lsm_poly_degree < 0:
            raise ValueError("lsm_poly_degree must be a non-negative integer.")
        if self.option_type not in ("call", "put"):
            raise ValueError("option_type must be 'call' or 'put'.")
        if int(num_simulations) // 2 * 2 == 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")

    def _next(self, S0, T, num_simulations, num_time_steps):
        k = self._pricings
        self._pricings += 1
        p = _ffi.make_params(model="gbm", is_put=(self.option_type == "put"), semantics="reference",
                             n_paths=int(num_simulations) // 2 * 2, n_steps=int(num_time_steps), S0=S0, K=self.K,
                             r=self.r, sigma=self.sigma, T=T, seed=int(self.seed), stream=k)
        return p, int(self.seed) + k

    def _run(self, ctx, job):
        p, net_seed = job
        if self.regressor == "nn":
            return ctx.price_american_contnet(p, NN_HIDDEN, NN_EPOCHS, NN_LR, net_seed)
        return ctx.price_american(p)

    def price_american_option(self, S0, T, num_simulations=10000, num_time_steps=50, plot_paths=False):
        self._check(S0, T, num_simulations, num_time_steps)
        out = self._run(_ffi.default_context(), self._next(S0, T, num_simulations, num_time_steps))
        self.last_result = out
        if self.verbose:
            print(f"Probability option expires worthless: {out['zero_prob']:.2%}")
            print(f"Estimated American {self.option_type} price: ${out['price']:.4f} "
                  f"(S0={S0}, K={self.K}, T={T}, r={self.r}, sigma={self.sigma}, "
                  f"simulations={num_simulations}, steps={num_time_steps})")
            print(f"Mean: ${out['price']:.4f}")
            print(f"Std Dev: ${out['std']:.4f}")
        return out["price"]

    def compute_curve_for_S0(self, S0, intervals_per_day, total_points, num_simulations, plot_paths) -> List[Dict[str, Any]]:
        """:214-227.  Streams are handed out in the order the sequential loop would consume its generator, so the
        records equal point-by-point calls on a pricer in the same state."""
        points, jobs = [], []
        for i in range(total_points, 0, -1):
            d = i / intervals_per_day
            T = d / 365.0
            steps = max(2, min(500, int(math.ceil(d * intervals_per_day))))
            self._check(S0, T, num_simulations, steps)
            points.append(d)
            jobs.append(self._next(S0, T, num_simulations, steps))
        if not jobs:
            return []
        if self.regressor == "nn":
            outs = _ffi.default_context().price_american_contnet_batch([p for p, _ in jobs], NN_HIDDEN, NN_EPOCHS, NN_LR,
                                                                       [s for _, s in jobs])
        else:
            outs = _ffi.default_context().price_american_batch([p for p, _ in jobs])
        self.last_result = outs[-1]
        return [{"S0": float(S0), "Days to Expiry": d, "Option Value": o["price"]} for d, o in zip(points, outs)]


def compute_curve_worker(S0, K, r, sigma, option_type, lsm_poly_degree, seed, intervals_per_day, total_points,
                         num_simulations, plot_paths):
    pricer = OptionPricer(K, r, sigma, option_type, lsm_poly_degree, seed)
    return pricer.compute_curve_for_S0(S0, intervals_per_day, total_points, num_simulations, plot_paths)
