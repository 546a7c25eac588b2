"""Drop-in for the Monte-Carlo pricer inside the reference's Heston calibrator
(options_model_3/heston_calibration.py): `HestonParams` (:34-73), `CalibrationConfig` (:75-90, the
fields the pricer reads) and `HestonPricer.price_european_option` / `price_options_batch`
(:259-312) -- the objective function's inner loop: 100k paths x 100 steps per expiry, dozens of
strikes, thousands of optimizer iterations.  The optimizer itself (`HestonCalibrator`) stays the
reference's scipy code and simply calls these methods.

Scheme = the calibrator's own (NOT the pricer's one in options_model_3.py): variance floored at
1e-8 before use and at store, arithmetic Euler for S, rho-mix before the antithetic stacking
(:223-255).  On the GPU the terminal spots go to a 4-byte-per-path buffer and every strike is one
workgroup's reduction over it; no [path][step] matrix is ever built.

Like the reference, consecutive calls use fresh normals (the reference's `self.rng` keeps
advancing; here every call takes the next Philox sub-stream of `config.seed`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _ffi


@dataclass
class HestonParams:
    kappa: float
    theta: float
    sigma: float
    rho: float
    v0: float

    def __post_init__(self):
        if not (0 < self.kappa < 20):
            raise ValueError(f"kappa={self.kappa} must be in (0, 20)")
        if not (0 < self.theta < 2):
            raise ValueError(f"theta={self.theta} must be in (0, 2)")
        if not (0 < self.sigma < 3):
            raise ValueError(f"sigma={self.sigma} must be in (0, 3)")
        if not (-1 < self.rho < 1):
            raise ValueError(f"rho={self.rho} must be in (-1, 1)")
        if not (0 < self.v0 < 2):
            raise ValueError(f"v0={self.v0} must be in (0, 2)")

    def to_array(self) -> np.ndarray:
        return np.array([self.kappa, self.theta, self.sigma, self.rho, self.v0])

    @classmethod
    def from_array(cls, x) -> "HestonParams":
        return cls(kappa=x[0], theta=x[1], sigma=x[2], rho=x[3], v0=x[4])

    def feller_condition(self) -> bool:
        return 2 * self.kappa * self.theta >= self.sigma**2


@dataclass
class CalibrationConfig:
    use_vega_weighting: bool = True
    min_vega_weight: float = 0.01
    max_iterations: int = 2000
    tolerance: float = 1e-8
    n_mc_paths: int = 100000
    n_time_steps: int = 100
    use_antithetic: bool = True
    seed: int = 42
    verbose: bool = True
    plot_results: bool = True
    optimization_methods: List[str] = field(
        default_factory=lambda: ["L-BFGS-B", "differential_evolution", "dual_annealing"])
    fallback_enabled: bool = True
    regime_detection: bool = True


class HestonPricer:
    def __init__(self, config: CalibrationConfig, device: int = 0):
        self.config = config
        self.device = device
        self._stream = 0  # next Philox sub-stream (plays the role of the advancing self.rng)

    def _strikes(self, params: HestonParams, S0, strikes, T, r, is_put):
        n_paths = int(self.config.n_mc_paths) // 2 * 2
        self._stream += 1
        prices, errs = _ffi.default_context(self.device).heston_price_strikes(
            n_paths, int(self.config.n_time_steps), float(S0), float(r), float(T), params.v0, params.kappa,
            params.theta, params.sigma, params.rho, strikes, is_put=is_put, seed=int(self.config.seed),
            stream=self._stream, scheme=_ffi.HESTON_SCHEMES["calibrator"])
        return prices, errs

    def price_european_option(self, params: HestonParams, S0: float, K: float, T: float,
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

    def price_options_batch(self, params: HestonParams, S0: float, K_array, T_array,
                            r: float = 0.05) -> np.ndarray:
        """Calls only, grouped by expiry: one simulation per distinct T (:289-306)."""
        K_array = np.asarray(K_array, np.float64)
        T_array = np.asarray(T_array, np.float64)
        prices = np.zeros(len(K_array))
        for T in np.unique(T_array):
            mask = T_array == T
            if not mask.any():
                continue
            try:
                p, _ = self._strikes(params, S0, K_array[mask], float(T), r, False)
                prices[mask] = p
            except Exception as e:  # noqa: BLE001
                print(f"Warning: Batch pricing failed for T={T}: {e}")
                prices[mask] = np.nan
        return prices
