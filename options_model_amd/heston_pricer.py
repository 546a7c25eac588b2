"""Drop-in for the Monte-Carlo pricer inside the reference's Heston calibrator
(options_model_3/heston_calibration.py): `HestonPricer.price_european_option` / `price_options_batch`
(:259-312) -- the objective function's inner loop: 100k paths x 100 steps per expiry, dozens of
strikes, thousands of optimizer iterations.  The optimizer itself (`HestonCalibrator`) stays the
reference's scipy code and simply calls these methods.

The parameter and configuration objects are the CALLER's: the reference's own `HestonParams`
(heston_calibration.py:34-73) and `CalibrationConfig` (:75-90) instances are accepted as they are --
this module only reads attributes (`params.kappa/theta/sigma/rho/v0`, `config.n_mc_paths/n_time_steps/
seed`) and re-declares neither schema.  Any object with those attributes works (a SimpleNamespace, a
dataclass of your own, a dict through `as_params` / `as_config`).

Scheme = the calibrator's own (NOT the pricer's one in options_model_3.py): variance floored at
1e-8 before use and at store, arithmetic Euler for S, rho-mix before the antithetic stacking
(:223-255).  On the GPU the terminal spots go to a 4-byte-per-path buffer and every strike is one
workgroup's reduction over it; no [path][step] matrix is ever built.

Like the reference, consecutive calls use fresh normals (the reference's `self.rng` keeps
advancing; here every call takes the next Philox sub-stream of `config.seed`).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import _ffi

_PARAM_FIELDS = ("kappa", "theta", "sigma", "rho", "v0")


def as_params(obj=None, **kw):
    """Heston parameters by duck typing: an object with kappa/theta/sigma/rho/v0 attributes (e.g. the
    reference's HestonParams), a mapping with those keys, or keyword arguments."""
    if obj is None:
        obj = kw
    if isinstance(obj, dict):
        obj = SimpleNamespace(**obj)
    missing = [f for f in _PARAM_FIELDS if not hasattr(obj, f)]
    if missing:
        raise ValueError(f"Heston parameters lack {missing}")
    return obj


def as_config(obj=None, **kw):
    """Pricer configuration by duck typing: n_mc_paths (default 100000), n_time_steps (100), seed (42)
    are all this module reads; everything else on the object is the calibrator's business."""
    if obj is None:
        obj = kw
    if isinstance(obj, dict):
        obj = SimpleNamespace(**obj)
    return obj


class HestonPricer:
    def __init__(self, config=None, device: int = 0):
        self.config = as_config(config)
        self.device = device
        self._stream = 0  # next Philox sub-stream (plays the role of the advancing self.rng)

    def _strikes(self, params, S0, strikes, T, r, is_put):
        params = as_params(params)
        cfg = self.config
        n_paths = int(getattr(cfg, "n_mc_paths", 100000)) // 2 * 2
        self._stream += 1
        prices, errs = _ffi.default_context(self.device).heston_price_strikes(
            n_paths, int(getattr(cfg, "n_time_steps", 100)), float(S0), float(r), float(T), float(params.v0),
            float(params.kappa), float(params.theta), float(params.sigma), float(params.rho), strikes,
            is_put=is_put, seed=int(getattr(cfg, "seed", 42)), stream=self._stream,
            scheme=_ffi.HESTON_SCHEMES["calibrator"])
        return prices, errs

    def price_european_option(self, params, S0: float, K: float, T: float,
                              r: float = 0.05, option_type: str = "call") -> float:
        try:
            kind = option_type.lower()
            if kind not in ("call", "put"):
                raise ValueError(f"Unknown option type: {option_type}")
            prices, _ = self._strikes(params, S0, [float(K)], T, r, kind == "put")
            return float(prices[0])
        except Exception as e:  # noqa: BLE001  (the reference swallows and returns nan, :279-281)
            print(f"Warning: Pricing failed for K={K}, T={T}: {e}")
            return float("nan")

    def price_options_batch(self, params, S0: float, K_array, T_array,
                            r: float = 0.05) -> np.ndarray:
        """Calls only, grouped by expiry: one simulation per distinct T (:289-306) -- here ALL of them in one launch set
        (omc_heston_price_surface: the expiry on grid.y, every quote one workgroup, one wait), each expiry on the Philox
        sub-stream the reference's loop `for T in unique_T` would have reached it on.  Quote for quote the bits of the
        per-expiry calls (`per_expiry=True` keeps that form: one synchronous call per expiry, as the reference's loop)."""
        return self._batch(params, S0, K_array, T_array, r, per_expiry=False)

    def price_options_batch_per_expiry(self, params, S0: float, K_array, T_array, r: float = 0.05) -> np.ndarray:
        """The reference's own loop shape (:291-310): one simulation + one host wait per distinct expiry."""
        return self._batch(params, S0, K_array, T_array, r, per_expiry=True)

    def _batch(self, params, S0, K_array, T_array, r, per_expiry):
        K_array = np.asarray(K_array, np.float64)
        T_array = np.asarray(T_array, np.float64)
        prices = np.zeros(len(K_array))
        if len(K_array) == 0:
            return prices
        uniq, expiry_of = np.unique(T_array, return_inverse=True)
        if not per_expiry:
            try:
                prm = as_params(params)
                cfg = self.config
                n_paths = int(getattr(cfg, "n_mc_paths", 100000)) // 2 * 2
                streams = self._stream + 1 + np.arange(len(uniq), dtype=np.uint64)
                got, _ = _ffi.default_context(self.device).heston_price_surface(
                    n_paths, int(getattr(cfg, "n_time_steps", 100)), float(S0), float(r), float(prm.v0), float(prm.kappa),
                    float(prm.theta), float(prm.sigma), float(prm.rho), uniq, streams, K_array, expiry_of, is_put=False,
                    seed=int(getattr(cfg, "seed", 42)), scheme=_ffi.HESTON_SCHEMES["calibrator"])
                self._stream += len(uniq)
                return got
            except Exception as e:  # noqa: BLE001  (the reference prints and returns nan for what failed, :308-310)
                print(f"Warning: Batch pricing failed: {e}")
                self._stream += len(uniq)
                return np.full(len(K_array), np.nan)
        for T in uniq:
            mask = T_array == T
            try:
                p, _ = self._strikes(params, S0, K_array[mask], float(T), r, False)
                prices[mask] = p
            except Exception as e:  # noqa: BLE001
                print(f"Warning: Batch pricing failed for T={T}: {e}")
                prices[mask] = np.nan
        return prices
