"""v2 class surface (reference options_model_2.py:176-355, :443-457) on the GPU hot path:
`OptionPricer` and the `compute_curve_worker` that options_model_2_ui.py imports.

Same per-step sticky flow as v1; Heston paths are antithetic here (v2's simulator is not,
options_model_2.py:150-170) -- a variance reduction, not a change of distribution.

Regressor: by default the reference's own -- a fresh ContNet(1 -> nn_hidden -> nn_hidden -> 1) per time
step, nn_epochs full-batch Adam(lr = nn_lr) steps from torch's default initialisation
(options_model_2.py:291-301) -- as HIP kernels (omc_contnet.hip); nn_hidden / nn_epochs / nn_lr are
honoured.  The reference seeds torch once per pricing with `seed` (:241); our nets are keyed by the same
`seed` (a different stream, the same distribution): same arguments, same price.  `regressor="poly"`
(keyword, or OMC_REGRESSOR=poly in the environment) solves OLS on [1, u, u^2] per step instead
(BASELINE.json's "polynomial LSM"; the reference accepts lsm_poly_degree and ignores it, :178-179) and
then ignores the nn_* arguments.  The reference's mask / discount / strict-'>' / statistics logic is
pinned to recorded runs of this very class (tests/golden/per_step_ref.npz).
"""
from __future__ import annotations

import logging
import math
from typing import Any, Dict, List, Optional

from .. import _ffi
from ..api import heston_defaults
from ._regressor import resolve


class OptionPricer:
    def __init__(self, K: float, r: float, sigma: Optional[float], option_type: str = "call",
                 lsm_poly_degree: int = 2, seed: int = 42, use_heston: bool = False,
                 heston_params: Optional[Dict[str, Any]] = None, nn_hidden: int = 32,
                 nn_epochs: int = 10, nn_lr: float = 1e-3, verbose: bool = False, *,
                 regressor: Optional[str] = None):
        self.K, self.r, self.sigma, self.option_type = K, r, sigma, option_type
        self.lsm_poly_degree, self.seed = lsm_poly_degree, seed
        self.use_heston, self.heston_params = use_heston, heston_params
        self.nn_hidden, self.nn_epochs, self.nn_lr, self.verbose = nn_hidden, nn_epochs, nn_lr, verbose
        self.regressor = resolve(regressor)
        self.last_result: Optional[dict] = None

    def price_american_option(self, S0: float, T: float, num_simulations: int = 10000,
                              num_time_steps: int = 50, plot_paths: bool = False) -> float:
        if S0 <= 0 or self.K <= 0 or T <= 0 or (self.sigma is None and not self.use_heston):
            raise ValueError("S0, K, T, and sigma must be positive.")
        if self.r < 0:
            raise ValueError("r must be non-negative.")
        if num_simulations <= 0 or num_time_steps <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        if not isinstance(self.lsm_poly_degree, int) or self.lsm_poly_degree < 0:
            raise ValueError("lsm_poly_degree must be a non-negative integer.")
        if self.option_type not in ("call", "put"):
            raise ValueError("option_type must be 'call' or 'put'.")
        M = int(num_simulations) // 2 * 2
        if M == 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        out = self._price(self._params(S0, T, M, num_time_steps))
        self.last_result = out
        if self.verbose:
            logging.info(f"Probability option expires worthless: {out['zero_prob']:.2%}")
            logging.info(f"Estimated American {self.option_type} price: ${out['price']:.4f} "
                         f"(S0={S0}, K={self.K}, T={T}, r={self.r}, sigma={self.sigma}, "
                         f"simulations={num_simulations}, steps={num_time_steps}, heston={self.use_heston})")
        return out["price"]

    def _price(self, params, ctx=None):
        ctx = ctx or _ffi.default_context()
        if self.regressor == "nn":
            if not (1 <= int(self.nn_hidden) <= 128):
                raise ValueError("nn_hidden must be in 1 .. 128.")
            return ctx.price_american_contnet(params, int(self.nn_hidden), int(self.nn_epochs), float(self.nn_lr),
                                              int(self.seed))
        return ctx.price_american(params)

    def _params(self, S0, T, M, steps):
        if self.use_heston and self.heston_params is not None:
            kw = dict(model="heston", **heston_defaults(self.sigma, self.heston_params))
        else:
            kw = dict(model="gbm")
        return _ffi.make_params(is_put=(self.option_type == "put"), semantics="reference", n_paths=M,
                                n_steps=int(steps), S0=S0, K=self.K, r=self.r,
                                sigma=self.sigma or 0.0, T=T, seed=int(self.seed), stream=0, **kw)

    def compute_curve_for_S0(self, S0: float, intervals_per_day: int, total_points: int,
                             num_simulations: int, plot_paths: bool) -> List[Dict[str, Any]]:
        """options_model_2.py:336-355: independent points, every one reseeded with self.seed -> one batched
        set of launches instead of a loop of pricings (polynomial regressor and per-step network alike)."""
        points = []
        for i in range(total_points, 0, -1):
            d = i / intervals_per_day
            points.append((d, d / 365, max(10, min(130, int(math.ceil(d))))))
        if not points:
            return []
        if S0 <= 0 or self.K <= 0 or (self.sigma is None and not self.use_heston):
            raise ValueError("S0, K, T, and sigma must be positive.")
        if self.r < 0:
            raise ValueError("r must be non-negative.")
        if num_simulations <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        if not isinstance(self.lsm_poly_degree, int) or self.lsm_poly_degree < 0:
            raise ValueError("lsm_poly_degree must be a non-negative integer.")
        if self.option_type not in ("call", "put"):
            raise ValueError("option_type must be 'call' or 'put'.")
        M = int(num_simulations) // 2 * 2
        if M == 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        params = [self._params(S0, T, M, st) for _, T, st in points]
        if self.regressor == "nn":
            if not (1 <= int(self.nn_hidden) <= 128):
                raise ValueError("nn_hidden must be in 1 .. 128.")
            outs = _ffi.default_context().price_american_contnet_batch(
                params, int(self.nn_hidden), int(self.nn_epochs), float(self.nn_lr), int(self.seed))
        else:
            outs = _ffi.default_context().price_american_batch(params)
        self.last_result = outs[-1]
        return [{"S0": S0, "Days to Expiry": d, "Option Value": o["price"]} for (d, _, _), o in zip(points, outs)]


def compute_curve_worker(S0, K, r, sigma, option_type, lsm_poly_degree, seed, intervals_per_day,
                         total_points, num_simulations, plot_paths, use_heston, heston_params,
                         nn_hidden=32, nn_epochs=10, nn_lr=1e-3, verbose=False):
    """Never raises (returns []), like the reference worker (options_model_2.py:455-457)."""
    try:
        pricer = OptionPricer(K, r, sigma, option_type, lsm_poly_degree, seed, use_heston,
                              heston_params, nn_hidden=nn_hidden, nn_epochs=nn_epochs, nn_lr=nn_lr,
                              verbose=verbose)
        return pricer.compute_curve_for_S0(S0, intervals_per_day, total_points, num_simulations,
                                           plot_paths)
    except Exception as e:  # noqa: BLE001
        logging.error(f"Error in worker for S0={S0}: {e}")
        return []
