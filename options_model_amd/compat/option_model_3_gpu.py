"""The call surface of the reference's "GPU" file (options_model_3/option_model_3_gpu.py) on the HIP hot path.

That file cannot be imported as committed (class-body indentation error at :594-596); it is the
author's design for a GPU variant, and this module is what its entry points look like when they work:

  simulate_bs_paths_torch / _bandwidth_optimized / simulate_heston_paths_torch   (:117-248)
        -> float32 torch tensors [num_time_steps+1, num_simulations] on `device`, filled by gbm_paths_kernel /
           heston_paths_kernel (one launch instead of a Python loop of ~6 torch kernels per step).  The
           normals come from Philox keyed by a seed drawn from torch's global generator, so
           torch.manual_seed(...) makes them reproducible, as it does for the reference's torch.randn.
  AdvancedOptionPricer with the GPU file's constructor order (nn_layers, nn_dropout before verbose, :548-569)
        and its method names: price_european_gpu, price_american_enhanced_lsm_gpu (with the file's short-
        maturity step rule :664-667 and its 50,000-path cap :675), price_american_with_control_variate,
        price_american_option, compute_curve_for_S0                                (:605-908)
  compute_curve_worker_gpu, compute_multiple_S0_gpu_batch                          (:910-956)

The backward induction is the one of options_model_amd.pricer.AdvancedOptionPricer (the v3 two-pass flow,
network regressor by default, regressor="poly" / OMC_REGRESSOR=poly for the polynomial).
torch is only needed by the simulate_* functions (they return torch tensors); everything else is ctypes.
"""
from __future__ import annotations

import logging
import math
from typing import Any, Dict, List, Optional

from .. import _ffi
from ..pricer import (AdvancedOptionPricer as _Pricer, BlackScholesGreeks, RNGManager,  # noqa: F401  (re-exported)
                      monte_carlo_price_streaming, welford_batch_update)

MAX_PATHS_PER_PRICING = 50000  # option_model_3_gpu.py:675


def get_device():
    """option_model_3_gpu.py:35-44: the accelerator if there is one."""
    import torch
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def clear_gpu_memory():
    """:46-52.  The library keeps its workspaces for reuse; only torch's cache is dropped."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def check_gpu_memory():
    import torch
    if torch.cuda.is_available():
        free, total = torch.cuda.mem_get_info()
        print(f"GPU memory: {(total - free) / 1e9:.2f} GB used of {total / 1e9:.2f} GB")


class _TensorView:
    """What _ffi.Context's path generators need of an output matrix: its address and its leading dimension."""

    def __init__(self, ptr: int, shape):
        self.ptr, self.shape = ptr, shape


def _device_index(device) -> int:
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("options_model_amd generates paths on the GPU: pass a cuda device "
                           "(there is no CPU fallback).")
    return device.index or 0


def _draw_seed() -> int:
    import torch
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _paths(device, num_simulations: int, num_time_steps: int, antithetic: bool, fill):
    """fill(ctx, view, n_paths, seed, stream, antithetic) for the even antithetic block (or all paths), then for
    the single extra path the reference appends when num_simulations is odd (:140-146, :227-246)."""
    import torch
    dev = _device_index(device)
    n, N = int(num_simulations), int(num_time_steps)
    if n <= 0 or N <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    ctx = _ffi.default_context(dev)
    S = torch.empty((N + 1, n), dtype=torch.float32, device=torch.device("cuda", dev))
    torch.cuda.synchronize(dev)  # the allocation is torch's; the fill runs on the library's stream
    seed = _draw_seed()
    main = n if not antithetic else n // 2 * 2
    if main:
        fill(ctx, _TensorView(S.data_ptr(), (N + 1, n)), main, seed, 0, antithetic)
    if main < n:
        fill(ctx, _TensorView(S.data_ptr() + 4 * main, (N + 1, n)), 1, seed, 1, False)
    ctx.sync()
    return S


def simulate_bs_paths_torch(S0: float, r: float, T: float, sigma: float, num_simulations: int,
                            num_time_steps: int, device):
    """:117-148: antithetic pairs (j, j + M/2), plus one plain path when num_simulations is odd."""
    return _paths(device, num_simulations, num_time_steps, True,
                  lambda ctx, out, m, seed, stream, anti: ctx.gbm_paths(m, int(num_time_steps), S0, r, sigma, T, seed,
                                                                       stream=stream, antithetic=anti, out=out))


def simulate_bs_paths_torch_bandwidth_optimized(S0: float, r: float, T: float, sigma: float, num_simulations: int,
                                                num_time_steps: int, device):
    """:150-185: independent (NOT antithetic) paths.  The reference batches to ~1.5 GB and sums in log space to
    save bandwidth; here the matrix is written exactly once, by one kernel."""
    return _paths(device, num_simulations, num_time_steps, False,
                  lambda ctx, out, m, seed, stream, anti: ctx.gbm_paths(m, int(num_time_steps), S0, r, sigma, T, seed,
                                                                       stream=stream, antithetic=False, out=out))


def simulate_heston_paths_torch(S0: float, r: float, T: float, v0: float, kappa: float, theta: float, xi: float,
                                rho: float, num_simulations: int, num_time_steps: int, device):
    """:187-248: Euler with the variance clamped where it is stored (scheme "reference"), antithetic in both
    normals; an odd num_simulations gets its last path from the antithetic pair of a separate stream."""
    import torch
    n, N = int(num_simulations), int(num_time_steps)
    dev = _device_index(device)
    if n <= 0 or N <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    ctx = _ffi.default_context(dev)
    seed = _draw_seed()
    even = n // 2 * 2
    S = torch.empty((N + 1, n), dtype=torch.float32, device=torch.device("cuda", dev))
    torch.cuda.synchronize(dev)
    if even:
        ctx.heston_paths(even, N, S0, r, T, v0, kappa, theta, xi, rho, seed, stream=0, scheme=0,
                         out=_TensorView(S.data_ptr(), (N + 1, n)))
    if even < n:  # Heston paths come in antithetic pairs: make one pair aside, keep its first path
        pair = ctx.heston_paths(2, N, S0, r, T, v0, kappa, theta, xi, rho, seed, stream=1, scheme=0)
        ctx.sync()
        S[:, n - 1] = torch.from_numpy(pair.to_host()[:, 0]).to(S.device)
        pair.free()
    ctx.sync()
    return S


class AdvancedOptionPricer(_Pricer):
    def __init__(self, K: float, r: float, sigma: Optional[float], option_type: str = "call",
                 rng_manager: Optional[RNGManager] = None, use_heston: bool = False,
                 heston_params: Optional[Dict[str, Any]] = None, nn_hidden: int = 128, nn_epochs: int = 25,
                 nn_lr: float = 1e-3, nn_layers: int = 3, nn_dropout: float = 0.10, verbose: bool = False,
                 iv_model=None, use_streaming: bool = True, chunk_size: int = 500,
                 european_approximation: bool = False, use_control_variate: bool = True, *,
                 regressor: Optional[str] = None, device: int = 0):
        super().__init__(K, r, sigma, option_type, rng_manager, use_heston, heston_params, nn_hidden=nn_hidden,
                         nn_epochs=nn_epochs, nn_lr=nn_lr, verbose=verbose, iv_model=iv_model,
                         use_streaming=use_streaming, chunk_size=chunk_size,
                         european_approximation=european_approximation, use_control_variate=use_control_variate,
                         nn_layers=nn_layers, nn_dropout=nn_dropout, regressor=regressor, device=device)

    def price_european_gpu(self, S0: float, T: float, num_simulations: int = 10000, num_time_steps: int = 50) -> float:
        """:605-653"""
        return self.price_european_streaming(S0, T, num_simulations, num_time_steps)

    def price_american_enhanced_lsm_gpu(self, S0: float, T: float, num_simulations: int = 10000,
                                        num_time_steps: int = 50) -> float:
        """:655-839: short maturities get 2 steps per day (10..25), at most 50,000 paths per pricing."""
        if S0 <= 0 or self.K <= 0 or T <= 0:
            raise ValueError("S0, K, T must be positive.")
        days = max(1, int(round(T * 365)))
        if days < 10:
            num_time_steps = max(10, min(25, days * 2))
        return self.price_american_enhanced_lsm(S0, T, min(int(num_simulations), MAX_PATHS_PER_PRICING),
                                                num_time_steps)

    def price_american_with_control_variate(self, S0: float, T: float, num_simulations: int = 10000,
                                            num_time_steps: int = 50) -> float:
        """:841-866"""
        american = self.price_american_enhanced_lsm_gpu(S0, T, num_simulations, num_time_steps)
        if not self.use_control_variate or self.sigma is None:
            return american
        european_mc = self.price_european_gpu(S0, T, num_simulations, num_time_steps)
        european_bs = BlackScholesGreeks.black_scholes_price(S0, self.K, T, self.r, self.sigma, self.option_type)
        cv = american + 1.0 * (european_bs - european_mc)
        if self.verbose:
            print(f"American: {american:.4f}, European MC: {european_mc:.4f}, "
                  f"European Analytical: {european_bs:.4f}, CV Adjusted: {cv:.4f}")
        return cv

    def price_american_option(self, S0: float, T: float, num_simulations: int = 10000, num_time_steps: int = 50,
                              plot_paths: bool = False) -> float:
        """:868-885 (the European shortcut does not depend on use_streaming here)"""
        if self.european_approximation:
            if self.verbose:
                print("WARNING: Using European approximation for American option")
            return self.price_european_gpu(S0, T, num_simulations, num_time_steps)
        if self.use_control_variate and self.sigma is not None:
            return self.price_american_with_control_variate(S0, T, num_simulations, num_time_steps)
        return self.price_american_enhanced_lsm_gpu(S0, T, num_simulations, num_time_steps)

    def compute_curve_for_S0(self, S0: float, intervals_per_day: int, total_points: int, num_simulations: int,
                             plot_paths: bool) -> List[Dict[str, Any]]:
        """:887-904.  Point by point: the step rule of price_american_enhanced_lsm_gpu depends on each point's T."""
        records = []
        for i in range(total_points, 0, -1):
            d = i / intervals_per_day
            steps = max(10, min(130, int(math.ceil(d))))
            records.append({"S0": S0, "Days to Expiry": d,
                            "Option Value": self.price_american_option(S0, d / 365, num_simulations, steps, plot_paths)})
        return records


def compute_curve_worker_gpu(S0, K, r, sigma, option_type, worker_seed, intervals_per_day, total_points,
                             num_simulations, plot_paths, use_heston, heston_params, nn_hidden=128, nn_epochs=25,
                             nn_lr=1e-3, nn_layers=3, nn_dropout=0.10, verbose=False, european_approximation=False,
                             use_control_variate=True, iv_model=None):
    """:910-932.  Never raises: logs and returns []."""
    try:
        pricer = AdvancedOptionPricer(K, r, sigma, option_type, RNGManager(worker_seed), use_heston, heston_params,
                                      nn_hidden=nn_hidden, nn_epochs=nn_epochs, nn_lr=nn_lr, nn_layers=nn_layers,
                                      nn_dropout=nn_dropout, verbose=verbose, iv_model=iv_model,
                                      european_approximation=european_approximation,
                                      use_control_variate=use_control_variate)
        return pricer.compute_curve_for_S0(S0, intervals_per_day, total_points, num_simulations, plot_paths)
    except Exception as e:  # noqa: BLE001
        logging.error(f"Error in GPU worker for S0={S0}: {e}")
        return []


def compute_multiple_S0_gpu_batch(s0_list: List[float], K: float, r: float, sigma: Optional[float], option_type: str,
                                  intervals_per_day: int, total_points: int, num_simulations: int,
                                  use_heston: bool = False, heston_params: Optional[Dict[str, Any]] = None,
                                  nn_hidden: int = 128, nn_epochs: int = 25, nn_lr: float = 1e-3, nn_layers: int = 3,
                                  nn_dropout: float = 0.10, verbose: bool = False, european_approximation: bool = False,
                                  use_control_variate: bool = True, iv_model=None, seed: int = 42):
    """:934-956: one pricer (one RNGManager) across all S0 values, records concatenated."""
    pricer = AdvancedOptionPricer(K, r, sigma, option_type, RNGManager(seed), use_heston, heston_params,
                                  nn_hidden=nn_hidden, nn_epochs=nn_epochs, nn_lr=nn_lr, nn_layers=nn_layers,
                                  nn_dropout=nn_dropout, verbose=verbose, iv_model=iv_model,
                                  european_approximation=european_approximation,
                                  use_control_variate=use_control_variate)
    all_records: List[Dict[str, Any]] = []
    for S0 in s0_list:
        all_records.extend(pricer.compute_curve_for_S0(S0, intervals_per_day, total_points, num_simulations, False))
    return all_records
