"""Drop-in for the call surface of options_model_3/options_model_3.py (the reference's CPU
pricer): same names, argument order, defaults, error messages and return types, with the hot
path (paths -> backward induction -> mean) on the MI355X through libomc.so.

Mirrored symbols (reference file:line):
    welford_batch_update, monte_carlo_price_streaming   options_model_3.py:33-63
    RNGManager                                          options_model_3.py:69-79
    BlackScholesGreeks                                  options_model_3.py:127-159
    AdvancedOptionPricer                                options_model_3.py:339-713
    compute_curve_worker_enhanced                       options_model_3.py:719-739

What differs, by construction (DESIGN.md "Parity"):
  * normals come from Philox4x32-10 keyed by the child seed RNGManager hands out, not from
    numpy's PCG64 -- the master-seed draw sequence (two draws per LSM pricing, one per
    500-path European chunk) is kept so curves consume seeds exactly as the reference does;
  * the continuation-value regressor is the reference's: ONE SingleLSMNet trained on the pass-1 rows
    (regressor="nn", the default; nn_regressor.py + the MFMA kernels of csrc/omc_mlp.hip);
    regressor="poly" (OLS on [1,u,u^2] per time step, same control flow) is the explicit fast option;
  * iv_model: pass an `options_model_amd.local_vol.IVModel` (same constructor contract as the
    reference's); paths are then simulated on the GPU through the IV network (local_vol.py).
"""
from __future__ import annotations

import logging
import math
import os
import warnings
from typing import Any, Dict, List, Optional

import numpy as np

from . import _ffi
from .api import heston_defaults

log = logging.getLogger(__name__)


# -- streaming statistics: Chan's parallel mean/M2 merge ------------------------------------
def welford_batch_update(mean, m2, n, batch):
    batch = np.asarray(batch, dtype=np.float64)
    k = batch.size
    if k == 0:
        return mean, m2, n
    bm = batch.mean()
    bm2 = float(((batch - bm) ** 2).sum())
    d = bm - mean
    tot = n + k
    return mean + d * (k / tot), m2 + bm2 + d * d * n * k / tot, tot


def monte_carlo_price_streaming(simulator_func, total_paths, chunk_size, *sim_args, **sim_kwargs):
    done, mean, m2 = 0, 0.0, 0.0
    while done < total_paths:
        b = min(chunk_size, total_paths - done)
        mean, m2, done = welford_batch_update(mean, m2, done, simulator_func(b, *sim_args, **sim_kwargs))
    var = m2 / (done - 1) if done > 1 else 0.0
    return mean, (math.sqrt(var / done) if done > 0 else 0.0), done


class RNGManager:
    """Master PCG64 handing out child seeds in [0, 2**31-1) -- identical draw sequence to the
    reference, so `RNGManager(42)` yields the same child seeds (tests/golden/scalars.json)."""

    def __init__(self, master_seed: int = 42):
        self.master_rng = np.random.default_rng(master_seed)
        self.master_seed = master_seed

    def get_child_rng(self) -> np.random.Generator:
        return np.random.default_rng(self.master_rng.integers(0, 2**31 - 1))

    def get_child_seed(self) -> int:
        return int(self.master_rng.integers(0, 2**31 - 1))


def _ncdf(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def _npdf(x):
    return math.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


class BlackScholesGreeks:
    @staticmethod
    def _d12(S, K, T, r, sigma):
        d1 = (math.log(S / K) + (r + 0.5 * sigma**2) * T) / (sigma * math.sqrt(T))
        return d1, d1 - sigma * math.sqrt(T)

    @staticmethod
    def greeks(S, K, T, r, sigma, option_type="call"):
        d1, d2 = BlackScholesGreeks._d12(S, K, T, r, sigma)
        common = -S * _npdf(d1) * sigma / (2 * math.sqrt(T))
        if option_type == "call":
            delta = _ncdf(d1)
            theta = common - r * K * math.exp(-r * T) * _ncdf(d2)
            rho = K * T * math.exp(-r * T) * _ncdf(d2)
        else:
            delta = -_ncdf(-d1)
            theta = common + r * K * math.exp(-r * T) * _ncdf(-d2)
            rho = -K * T * math.exp(-r * T) * _ncdf(-d2)
        return {"Delta": delta, "Gamma": _npdf(d1) / (S * sigma * math.sqrt(T)),
                "Vega": S * _npdf(d1) * math.sqrt(T) / 100, "Theta": theta / 365, "Rho": rho / 100}

    @staticmethod
    def black_scholes_price(S, K, T, r, sigma, option_type="call"):
        d1, d2 = BlackScholesGreeks._d12(S, K, T, r, sigma)
        if option_type == "call":
            return S * _ncdf(d1) - K * math.exp(-r * T) * _ncdf(d2)
        return K * math.exp(-r * T) * _ncdf(-d2) - S * _ncdf(-d1)


class AdvancedOptionPricer:
    def __init__(self, K: float, r: float, sigma: Optional[float], option_type: str = "call",
                 rng_manager: Optional[RNGManager] = None, use_heston: bool = False,
                 heston_params: Optional[Dict[str, Any]] = None, nn_hidden: int = 128,
                 nn_epochs: int = 25, nn_lr: float = 1e-3, verbose: bool = False, iv_model=None,
                 use_streaming: bool = True, chunk_size: int = 500,
                 european_approximation: bool = False, use_control_variate: bool = True,
                 # -- extensions (keyword-only in spirit; the GPU file adds nn_layers/nn_dropout
                 #    the same way, option_model_3_gpu.py:557-561)
                 nn_layers: int = 3, nn_dropout: float = 0.10, regressor: Optional[str] = None,
                 semantics: str = "two_pass", device: Optional[int] = None, n_gpus: Optional[int] = None,
                 devices: Optional[list] = None):
        """Called with the reference's own arguments only, this prices the way the reference does:
        ONE SingleLSMNet(7, nn_hidden, nn_layers) with dropout, trained on the pass-1 rows and applied
        in the sticky pass 2 (options_model_3.py:482-651) -- regressor "nn", the default -- on the GPU
        (hand-written MFMA trainer / pass-2 kernels for 64|128 units x 2|3 layers).  regressor="poly"
        is the explicit fast option: OLS on [1,u,u^2] per time step in the same two-pass control flow
        (the reference accepts lsm_poly_degree and never uses it).  The environment
        variable OMC_REGRESSOR=poly|nn changes the default for callers that cannot pass the argument
        (e.g. the Streamlit UI going through compute_curve_worker_enhanced).
        n_gpus > 1 (or OMC_N_GPUS for the same callers): the American pricing shards its paths over that many
        GPUs, one rank process per GPU -- started from here when this is a plain process (api.py, launcher.py);
        `devices` lists one HIP device per rank (default: rank r -> device r; OMC_DEVICES="0,1,.." likewise).
        Curves (compute_curve_for_S0) are thousands of SMALL pricings: they run as one batch on one GPU whatever
        n_gpus says."""
        if regressor is None:
            regressor = os.environ.get("OMC_REGRESSOR", "nn").lower()
        if regressor not in ("poly", "nn"):
            raise ValueError("regressor must be 'poly' or 'nn'.")
        if regressor == "poly" and (nn_hidden, nn_epochs, nn_lr, nn_layers, nn_dropout) != (128, 25, 1e-3, 3, 0.10):
            warnings.warn("AdvancedOptionPricer(regressor='poly') ignores nn_hidden / nn_epochs / nn_lr / nn_layers / "
                          "nn_dropout: the polynomial regressor has no network", stacklevel=2)
        self.K, self.r, self.sigma, self.option_type = K, r, sigma, option_type
        self.rng_manager = rng_manager or RNGManager()
        self.use_heston, self.heston_params = use_heston, heston_params
        self.nn_hidden, self.nn_epochs, self.nn_lr = nn_hidden, nn_epochs, nn_lr
        self.nn_layers, self.nn_dropout = nn_layers, nn_dropout
        self.verbose, self.iv_model = verbose, iv_model
        self.use_streaming, self.chunk_size = use_streaming, chunk_size
        self.european_approximation = european_approximation
        self.use_control_variate = use_control_variate
        self.regressor, self.semantics, self.device = regressor, semantics, _ffi.resolve_device(device)
        self.n_gpus = int(n_gpus if n_gpus is not None else os.environ.get("OMC_N_GPUS", "1"))
        if self.n_gpus < 1:
            raise ValueError("n_gpus must be a positive integer.")
        if devices is None and os.environ.get("OMC_DEVICES"):
            devices = [int(d) for d in os.environ["OMC_DEVICES"].split(",")]
        self.devices = devices
        self._calls = 0
        self.last_result: Optional[dict] = None

    # -- helpers
    def _ctx(self):
        return _ffi.default_context(self.device)

    def _payoff(self, S: np.ndarray) -> np.ndarray:
        return np.maximum(S - self.K, 0) if self.option_type == "call" else np.maximum(self.K - S, 0)

    def _model_kw(self):
        if self.use_heston and self.heston_params is not None:
            return dict(model="heston", **heston_defaults(self.sigma, self.heston_params))
        if self.sigma is None:
            raise ValueError("sigma is None: provide sigma, iv_model, or heston configuration")
        return dict(model="gbm")

    def _params(self, S0, T, M, N, seed, semantics):
        sem = {"two_pass": "two_pass", "per_step": "reference", "textbook": "textbook"}[semantics]
        return _ffi.make_params(is_put=(self.option_type == "put"), semantics=sem, n_paths=M,
                                n_steps=N, S0=S0, K=self.K, r=self.r, sigma=self.sigma or 0.0, T=T,
                                seed=seed, stream=0, **self._model_kw())

    # -- European (options_model_3.py:382-437)
    def price_european_streaming(self, S0: float, T: float, num_simulations: int = 10000,
                                 num_time_steps: int = 50) -> float:
        # the reference draws one child RNG per chunk_size-path chunk; consume the same number
        # of master draws and key the single fused launch with the first of them
        n_chunks = max(1, -(-int(num_simulations) // int(self.chunk_size)))
        seeds = [self.rng_manager.get_child_seed() for _ in range(n_chunks)]
        M = int(num_simulations) // 2 * 2
        if M <= 0:
            return 0.0
        if self.iv_model is not None:  # local-vol paths through the IV network (:394-395)
            from . import local_vol
            S = local_vol.simulate_local_vol_paths(S0, self.r, T, M, int(num_time_steps), self.iv_model,
                                                   self.K, seeds[0], stream=1)
            pay = (self.K - S[-1].double()).clamp(min=0) if self.option_type == "put" else \
                (S[-1].double() - self.K).clamp(min=0)
            return float(pay.mean()) * math.exp(-self.r * T)
        kw = self._model_kw()
        p = _ffi.make_params(is_put=(self.option_type == "put"), n_paths=M, n_steps=int(num_time_steps),
                             S0=S0, K=self.K, r=self.r, sigma=self.sigma or 0.0, T=T, seed=seeds[0],
                             stream=1, **kw)
        out = self._ctx().price_european(p)
        if self.verbose:
            var = max(out["sumsq"] / M - out["price"] ** 2, 0.0)
            print(f"European streaming MC: {out['price']:.4f} ± {math.sqrt(var / M):.4f} (n={M})")
        return out["price"]

    # -- American LSM (options_model_3.py:439-651)
    def price_american_enhanced_lsm(self, S0: float, T: float, num_simulations: int = 10000,
                                    num_time_steps: int = 50) -> float:
        if S0 <= 0 or self.K <= 0 or T <= 0:
            raise ValueError("S0, K, T must be positive.")
        if self.r < 0:
            raise ValueError("r must be non-negative.")
        if num_simulations <= 0 or num_time_steps <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        path_seed = self.rng_manager.get_child_seed()   # reference: rng = get_child_rng()
        torch_seed = self.rng_manager.get_child_seed()  # reference: torch.manual_seed(...)
        M = int(num_simulations) // 2 * 2
        if M == 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        self._calls += 1
        if self.iv_model is not None:  # :461-462: local-vol paths, then the same backward induction
            from . import local_vol
            S = local_vol.simulate_local_vol_paths(S0, self.r, T, M, int(num_time_steps), self.iv_model,
                                                   self.K, path_seed)
            if self.regressor == "nn":
                from . import nn_regressor
                res = nn_regressor.price_with_paths(S, self.K, self.r, T, self.option_type == "put",
                                                    torch_seed, nn_hidden=self.nn_hidden,
                                                    nn_layers=self.nn_layers, nn_dropout=self.nn_dropout,
                                                    nn_epochs=self.nn_epochs, nn_lr=self.nn_lr)
                res.pop("net", None)
            else:
                res = local_vol.price_american_on_paths(self._ctx(), S, self.K, self.r, T,
                                                        self.option_type == "put", self.semantics)
            self.last_result = res
            return res["price"]
        if self.regressor not in ("poly", "nn"):
            raise ValueError("regressor must be 'poly' or 'nn'.")
        if self.n_gpus > 1:  # paths sharded over rank processes (api.price_american_option, n_gpus)
            from . import api
            kw = self._model_kw()
            res = api.price_american_option(
                S0, self.K, self.r, self.sigma, T, M, int(num_time_steps), model=kw.pop("model"),
                option_type=self.option_type, regressor=self.regressor, semantics=self.semantics,
                heston_params=kw or None, seed=path_seed, stream=0, n_gpus=self.n_gpus, device=self.devices,
                nn_options=dict(nn_hidden=self.nn_hidden, nn_layers=self.nn_layers, nn_dropout=self.nn_dropout,
                                nn_epochs=self.nn_epochs, nn_lr=self.nn_lr, torch_seed=torch_seed))
            self.last_result = dict(price=res.price, std=res.std, stderr=res.stderr, n_paths=res.n_paths,
                                    n_exercised=res.n_exercised, sum_nitm=res.sum_nitm, zero_prob=res.zero_prob,
                                    R=res.sum_nitm, **res.info)
            return res.price
        if self.regressor == "nn":
            from . import nn_regressor
            res = nn_regressor.price_two_pass_nn(self, S0, T, M, int(num_time_steps), path_seed, torch_seed)
            self.last_result = res
            return res["price"]
        out = self._ctx().price_american(self._params(S0, T, M, int(num_time_steps), path_seed,
                                                      self.semantics))
        self.last_result = out
        return out["price"]

    # -- default wrapper (options_model_3.py:653-677); see SURVEY.md F10 for what it really is
    def price_american_with_control_variate(self, S0: float, T: float, num_simulations: int = 10000,
                                            num_time_steps: int = 50) -> float:
        american = self.price_american_enhanced_lsm(S0, T, num_simulations, num_time_steps)
        if not self.use_control_variate or self.sigma is None:
            return american
        european_mc = self.price_european_streaming(S0, T, num_simulations, num_time_steps)
        european_bs = BlackScholesGreeks.black_scholes_price(S0, self.K, T, self.r, self.sigma,
                                                            self.option_type)
        cv = american + 1.0 * (european_bs - european_mc)
        if self.verbose:
            print(f"American: {american:.4f}, European MC: {european_mc:.4f}, "
                  f"European Analytical: {european_bs:.4f}, CV Adjusted: {cv:.4f}")
        return cv

    def price_american_option(self, S0: float, T: float, num_simulations: int = 10000,
                              num_time_steps: int = 50, plot_paths: bool = False) -> float:
        if self.use_streaming and self.european_approximation:
            if self.verbose:
                print("WARNING: Using European approximation for American option (streaming mode)")
            return self.price_european_streaming(S0, T, num_simulations, num_time_steps)
        if self.use_control_variate and self.sigma is not None:
            return self.price_american_with_control_variate(S0, T, num_simulations, num_time_steps)
        return self.price_american_enhanced_lsm(S0, T, num_simulations, num_time_steps)

    def compute_curve_for_S0(self, S0: float, intervals_per_day: int, total_points: int,
                             num_simulations: int, plot_paths: bool) -> List[Dict[str, Any]]:
        """options_model_3.py:697-713.  The points of a curve are independent pricings, so they
        run as ONE batched set of launches (polynomial regressor: omc_price_american_batch; network
        regressor, the default: one net per point, trained side by side -- omc_mlp_train_epoch_batch);
        child seeds are drawn in exactly the order the sequential loop would draw them, so the result
        equals calling price_american_option point by point."""
        points = []
        for i in range(total_points, 0, -1):
            d = i / intervals_per_day
            points.append((d, d / 365, max(10, min(130, int(math.ceil(d))))))
        batchable = (self.iv_model is None and not (self.use_streaming and self.european_approximation)
                     and (self.regressor == "poly" or os.environ.get("OMC_NN_CURVE_BATCH", "1") != "0"))
        if not batchable:
            return [{"S0": S0, "Days to Expiry": d,
                     "Option Value": self.price_american_option(S0, T, num_simulations, steps, plot_paths)}
                    for d, T, steps in points]
        if S0 <= 0 or self.K <= 0:
            raise ValueError("S0, K, T must be positive.")
        if self.r < 0:
            raise ValueError("r must be non-negative.")
        if num_simulations <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        with_cv = self.use_control_variate and self.sigma is not None
        kw = self._model_kw()
        M = int(num_simulations) // 2 * 2
        if M == 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        n_chunks = max(1, -(-int(num_simulations) // int(self.chunk_size)))
        lsm_params, eur_params, nn_seeds = [], [], []
        for d, T, steps in points:
            path_seed = self.rng_manager.get_child_seed()
            torch_seed = self.rng_manager.get_child_seed()  # the reference's torch.manual_seed draw
            nn_seeds.append((path_seed, torch_seed))
            lsm_params.append(self._params(S0, T, M, steps, path_seed, self.semantics))
            if with_cv:
                eseed = [self.rng_manager.get_child_seed() for _ in range(n_chunks)][0]
                eur_params.append(_ffi.make_params(is_put=(self.option_type == "put"), n_paths=M,
                                                   n_steps=steps, S0=S0, K=self.K, r=self.r,
                                                   sigma=self.sigma or 0.0, T=T, seed=eseed, stream=1, **kw))
        ctx = self._ctx()
        if self.regressor == "nn":
            # the default regressor: one SingleLSMNet per point, all of them trained side by side
            # (nn_regressor.price_curve_nn); every point equals its own price_american_enhanced_lsm call
            from . import nn_regressor
            lsm = nn_regressor.price_curve_nn(self, [(S0, T, M, steps, ps, ts) for (d, T, steps), (ps, ts)
                                                     in zip(points, nn_seeds)])
        else:
            lsm = ctx.price_american_batch(lsm_params)
        eur = ctx.price_european_batch(eur_params) if with_cv else None
        self._calls += len(points)
        records = []
        for k, (d, T, steps) in enumerate(points):
            price = lsm[k]["price"]
            if with_cv:
                bs = BlackScholesGreeks.black_scholes_price(S0, self.K, T, self.r, self.sigma, self.option_type)
                price = price + 1.0 * (bs - eur[k]["price"])
            records.append({"S0": S0, "Days to Expiry": d, "Option Value": price})
        self.last_result = lsm[-1]
        return records


def compute_curve_worker_enhanced(S0, K, r, sigma, option_type, worker_seed, intervals_per_day,
                                  total_points, num_simulations, plot_paths, use_heston,
                                  heston_params, nn_hidden=128, nn_epochs=25, nn_lr=1e-3,
                                  verbose=False, european_approximation=False,
                                  use_control_variate=True):
    """Never raises: logs and returns [] like the reference worker (options_model_3.py:737-739)."""
    try:
        pricer = AdvancedOptionPricer(K, r, sigma, option_type, RNGManager(worker_seed), use_heston,
                                      heston_params, nn_hidden=nn_hidden, nn_epochs=nn_epochs,
                                      nn_lr=nn_lr, verbose=verbose,
                                      european_approximation=european_approximation,
                                      use_control_variate=use_control_variate)
        return pricer.compute_curve_for_S0(S0, intervals_per_day, total_points, num_simulations,
                                           plot_paths)
    except Exception as e:  # noqa: BLE001
        logging.error(f"Error in enhanced worker for S0={S0}: {e}")
        return []
