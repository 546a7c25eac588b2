"""v1 functional surface (reference Options_model.py:44-157, :190-211) on the GPU hot path.

Flow kept: per-time-step regress-and-decide with the sticky `exercised` mask, valued at t=dt
(Options_model.py:108-157).  Regressor: by default the reference's own -- a fresh ContNet(1-32-32-1) per
step, 10 full-batch Adam steps at lr 1e-3 (Options_model.py:14-25,127-139) -- as HIP kernels;
`regressor="poly"` (keyword, or OMC_REGRESSOR=poly in the environment) solves OLS on [1,u,u^2] per step
instead.  The reference never seeds torch here, so its nets differ from run to run; ours are keyed by
`seed` (same arguments, same price).  As in the reference every pricing reseeds with `seed`, so all
points of a curve share their normals (common random numbers across maturities, Options_model.py:74,197).
"""
from __future__ import annotations

import math

from .. import _ffi
from ._regressor import resolve

NN_HIDDEN, NN_EPOCHS, NN_LR = 32, 10, 1e-3  # Options_model.py:15,129,132


def price_american_option(S0, K, T, r, sigma, num_simulations=10000, num_time_steps=50,
                          option_type="call", lsm_poly_degree=2, plot_paths=False, seed=42, *, regressor=None):
    """-> (mean, std, probability the option ends worthless)"""
    if S0 <= 0 or K <= 0 or T <= 0 or sigma <= 0:
        raise ValueError("S0, K, T, and sigma must be positive.")
    if r < 0:
        raise ValueError("r must be non-negative.")
    if num_simulations <= 0 or num_time_steps <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    if not isinstance(lsm_poly_degree, int) or lsm_poly_degree < 0:
        raise ValueError("lsm_poly_degree must be a non-negative integer.")
    if option_type not in ("call", "put"):
        raise ValueError("option_type must be 'call' or 'put'.")
    M = int(num_simulations) // 2 * 2
    if M == 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    p = _ffi.make_params(model="gbm", is_put=(option_type == "put"), semantics="reference",
                         n_paths=M, n_steps=int(num_time_steps), S0=S0, K=K, r=r, sigma=sigma, T=T,
                         seed=int(seed), stream=0)
    if resolve(regressor) == "nn":
        out = _ffi.default_context().price_american_contnet(p, NN_HIDDEN, NN_EPOCHS, NN_LR, int(seed))
    else:
        out = _ffi.default_context().price_american(p)
    return out["price"], out["std"], out["zero_prob"]


def compute_curve_for_S0(S0, K, r, sigma, num_simulations, intervals_per_day, total_points,
                         option_type, lsm_poly_degree, plot_paths, seed, *, regressor=None):
    """Options_model.py:190-211.  All points share `seed` (the reference reseeds per pricing) and are
    independent: they run as one batched set of launches (omc_price_american_batch for the polynomial regressor,
    omc_price_american_contnet_batch for the per-step network); every point equals its own pricing call."""
    regressor = resolve(regressor)
    points = []
    for i in range(total_points, 0, -1):
        d = i / intervals_per_day
        points.append((d, d / 365, max(10, min(130, int(math.ceil(d))))))
    if not points:
        return []
    # same validation (and messages) as the per-point call
    if S0 <= 0 or K <= 0 or sigma <= 0:
        raise ValueError("S0, K, T, and sigma must be positive.")
    if r < 0:
        raise ValueError("r must be non-negative.")
    if num_simulations <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    if not isinstance(lsm_poly_degree, int) or lsm_poly_degree < 0:
        raise ValueError("lsm_poly_degree must be a non-negative integer.")
    if option_type not in ("call", "put"):
        raise ValueError("option_type must be 'call' or 'put'.")
    M = int(num_simulations) // 2 * 2
    if M == 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    params = [_ffi.make_params(model="gbm", is_put=(option_type == "put"), semantics="reference",
                               n_paths=M, n_steps=steps, S0=S0, K=K, r=r, sigma=sigma, T=T,
                               seed=int(seed), stream=0) for _, T, steps in points]
    if regressor == "nn":
        # one batched set of launches for all points: problem index on the grid, set sizes stay on the device
        outs = _ffi.default_context().price_american_contnet_batch(params, NN_HIDDEN, NN_EPOCHS, NN_LR, int(seed))
    else:
        outs = _ffi.default_context().price_american_batch(params)
    return [{"S0": S0, "Days to Expiry": d, "Option Value": o["price"], "Std Dev": o["std"],
             "Zero Prob": o["zero_prob"]} for (d, _, _), o in zip(points, outs)]
