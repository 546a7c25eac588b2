"""North-star facade (BASELINE.json):

    price_american_option(S0, K, r, sigma, T, n_paths, n_steps, model='GBM'|'Heston', ...)

A thin host-side wrapper over the C ABI (include/omc.h).  The reference has no function with
this exact signature; its three real call shapes are mirrored in
`options_model_amd.pricer` (options_model_3.py) and `options_model_amd.compat`
(Options_model.py, options_model_2.py).

semantics (the reference's exercise rule is not textbook LSM, and it is kept):
  "two_pass"  (default) control flow of options_model_3/options_model_3.py:482-651: pass 1
              regresses discounted TERMINAL payoffs with no decisions, pass 2 applies the
              sticky rule; valued at t = dt.  Moments are decision-independent -> one
              all-reduce across GPUs.
  "per_step"  control flow of Options_model.py:108-157 / options_model_2.py:278-313:
              regress-and-decide per time step with the sticky `exercised` mask.
  "textbook"  classic Longstaff-Schwartz (overwrite on earlier exercise, discounted to t=0).
regressor:
  "poly"      OLS on [1,u,u^2], u=S/K-1, one fit per time step (the reference validates
              lsm_poly_degree and then ignores it: Options_model.py:53,69-70)
n_gpus:
  1           this process, one GPU (`device`).
  N > 1       the paths shard by antithetic pair over N GPUs of one node, ONE PROCESS PER GPU; each rank prices its
              shard through the library's own RCCL communicator (dist.RcclPricer: regression moments and result sums
              all-reduced over xGMI) and the call returns the global result.  Two ways to get the ranks:
              * called from a PLAIN process (a script, a notebook, the Streamlit UI -- the reference's callers are
                single-process programs, options_model_2_ui.py:87-133): the ranks are started here as child
                processes on first use (launcher.RankPool: fresh interpreters, the parent never re-execs and makes
                no GPU call for them), keep their communicator across calls and are closed at exit.  `device` may
                list one HIP device per rank (default: rank r -> device r);
              * called by every rank of an N-rank job (`python -m torch.distributed.run --nproc-per-node N
                script.py`, or any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT): each rank is
                its own process already and every rank returns the same global result.
              It never silently prices on fewer GPUs: a rank that cannot start, or a job of another size, raises.
              regressor="nn" shards as well (nn_dist: rows per rank, statistics and gradient partials all-reduced).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

from . import _ffi

_SEM = {"two_pass": "two_pass", "v3": "two_pass", "reference_v3": "two_pass",
        "per_step": "reference", "v1": "reference", "reference_v1": "reference",
        "textbook": "textbook"}


@dataclass
class PriceResult:
    price: float
    stderr: float
    std: float
    zero_prob: float
    n_paths: int
    n_exercised: int
    sum_nitm: int
    model: str
    semantics: str
    option_type: str
    timings_ms: dict = field(default_factory=dict)
    info: dict = field(default_factory=dict)  # regressor='nn': trainer, batch, epochs_run, optimizer_steps ...

    def __float__(self):
        return float(self.price)


def _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=True):
    # same checks and messages as options_model_3.py:447-452,471-472 / Options_model.py:63-72
    if S0 <= 0 or K <= 0 or T <= 0:
        raise ValueError("S0, K, T must be positive.")
    if r < 0:
        raise ValueError("r must be non-negative.")
    if n_paths <= 0 or n_steps <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    if option_type not in ("call", "put"):
        raise ValueError("option_type must be 'call' or 'put'.")
    if need_sigma and (sigma is None or sigma <= 0):
        raise ValueError("sigma is None: provide sigma, iv_model, or heston configuration"
                         if sigma is None else "S0, K, T, and sigma must be positive.")


def heston_defaults(sigma, heston_params=None):
    """kappa=2, xi=0.3, rho=-0.7, v0=theta=sigma^2: the reference's hard-coded defaults
    (options_model_3.py:948-950,995-996; options_model_2_ui.py:74-80)."""
    hp = dict(v0=(sigma or 0.2) ** 2, kappa=2.0, theta=(sigma or 0.2) ** 2, xi=0.3, rho=-0.7)
    if heston_params:
        hp.update({k: float(v) for k, v in heston_params.items() if k in hp})
    return hp


def price_american_option(S0, K, r, sigma, T, n_paths, n_steps, model="GBM", option_type="put",
                          regressor="poly", semantics="two_pass", heston_params=None,
                          heston_scheme="reference", antithetic=True, seed=42, stream=0,
                          device=None, ctx=None, n_gpus=1, nn_options=None) -> PriceResult:
    """regressor: "poly" (OLS on [1, u, u^2] per time step: the flows of `semantics`), "nn" (the reference's network, two-pass
    flow), "ols7" (one least-squares fit on the reference's 7 features, two-pass flow).
    nn_options (regressor="nn" only): dict with any of nn_hidden, nn_layers, nn_dropout, nn_epochs, nn_lr, nn_batch,
    inference_dropout, torch_seed -- the network named by BASELINE config 5 (2 x 64) by default."""
    model_l = str(model).lower()
    n_gpus = int(n_gpus)
    if n_gpus < 1:
        raise ValueError("n_gpus must be a positive integer.")
    if device is None and n_gpus == 1:
        device = _ffi.resolve_device(None)  # 0, or what OMC_DEVICE says ("auto": spread processes over the GPUs)
    if model_l not in ("gbm", "heston"):
        raise ValueError("model must be 'GBM' or 'Heston'.")
    if semantics not in _SEM:
        raise ValueError(f"semantics must be one of {sorted(set(_SEM))}.")
    if regressor not in ("poly", "nn", "ols7"):
        raise ValueError("regressor must be 'poly', 'nn' or 'ols7'.")
    if n_gpus > 1 and not _in_job(n_gpus):
        # a plain process: validate here (the reference's messages), then let the rank pool do the collective call
        if ctx is not None:
            raise ValueError("n_gpus > 1 uses one context per rank process; do not pass ctx.")
        _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
        from . import launcher
        if isinstance(device, int) and not os.environ.get("OMC_RCCL_LIB"):
            # (real RCCL refuses two ranks on one card; only the shared-memory stand-in of the tests runs that way)
            raise ValueError("n_gpus > 1 needs one device per rank: pass device=[d0, d1, ...] or leave it unset.")
        devices = None if device is None else ([int(device)] * n_gpus if isinstance(device, int) else list(device))
        if devices is not None and len(devices) != n_gpus:
            raise ValueError(f"device lists {len(devices)} cards for n_gpus={n_gpus}.")
        kw = dict(S0=S0, K=K, r=r, sigma=sigma, T=T, n_paths=int(n_paths), n_steps=int(n_steps), model=model,
                  option_type=option_type, heston_params=heston_params, seed=int(seed), stream=int(stream))
        if regressor == "nn":
            kw.update(nn_options or {})
            d = launcher.pool(n_gpus, devices).call("price_american_option_nn", kw, timeout_s=3600.0)
        else:
            kw.update(regressor=regressor, semantics=semantics, heston_scheme=heston_scheme, antithetic=bool(antithetic))
            d = launcher.pool(n_gpus, devices).call("price_american_option", kw)
        d["info"] = dict(d.get("info", {}), launched_ranks=n_gpus)
        return PriceResult(**d)
    if regressor == "ols7":
        # ONE global least-squares fit on the reference's seven features (options_model_3.py:105-121) in its two-pass flow
        # (omc_price_american_ols7): the linear regressor between the per-step polynomial and the network.  Across GPUs the
        # paths shard by antithetic pair and the ranks' co-moments are merged by two small all-reduces: the fit is the job's.
        _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
        M = int(n_paths) // 2 * 2 if antithetic else int(n_paths)
        if M <= 0:
            raise ValueError("num_simulations and num_time_steps must be positive integers.")
        kwp = dict(model=model_l, is_put=(option_type == "put"), semantics="two_pass", antithetic=antithetic,
                   heston_scheme=heston_scheme, n_steps=int(n_steps), S0=S0, K=K, r=r, sigma=sigma or 0.0, T=T, seed=seed,
                   stream=stream, **heston_defaults(sigma, heston_params))
        info = dict(regressor="ols7")
        if n_gpus > 1:
            if ctx is not None:
                raise ValueError("n_gpus > 1 uses the job's own per-rank context; do not pass ctx.")
            sp = _job_pricer(n_gpus, device)
            out = sp.price_american_ols7(M, **kwp)
            info.update(n_gpus=n_gpus, rank=sp.rank, transport=sp.transport)
        else:
            c = ctx or _ffi.default_context(_ffi.resolve_device(None) if device is None else device)
            out = c.price_american_ols7(_ffi.make_params(n_paths=M, **kwp))
        Mg = out["n_paths"]
        var = max(out["sumsq"] / Mg - out["price"] ** 2, 0.0)
        info.update(weights=[float(v) for v in out["weights"]], y_mean=out["y_mean"], y_std=out["y_std"])
        return PriceResult(price=out["price"], stderr=math.sqrt(var / Mg), std=out["std"], zero_prob=out["zero_prob"], n_paths=Mg,
                           n_exercised=out["n_exercised"], sum_nitm=out["sum_nitm"], model=model_l, semantics="two_pass",
                           option_type=option_type, info=info)
    if regressor == "nn":
        if n_gpus > 1:
            from . import nn_dist
            return nn_dist.price_american_option_nn_sharded(
                _job_pricer(n_gpus, device), S0, K, r, sigma, T, n_paths, n_steps, model=model,
                option_type=option_type, heston_params=heston_params, seed=seed, stream=stream, **(nn_options or {}))
        from . import nn_regressor
        return nn_regressor.price_american_option_nn(
            S0, K, r, sigma, T, n_paths, n_steps, model=model, option_type=option_type,
            heston_params=heston_params, seed=seed, stream=stream, device=device, **(nn_options or {}))
    _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
    M = int(n_paths) // 2 * 2 if antithetic else int(n_paths)  # options_model_3.py:458
    if M <= 0:
        raise ValueError("num_simulations and num_time_steps must be positive integers.")
    hp = heston_defaults(sigma, heston_params)
    if n_gpus > 1:
        if ctx is not None:
            raise ValueError("n_gpus > 1 uses the job's own per-rank context; do not pass ctx.")
        sp = _job_pricer(n_gpus, device)
        out = sp.price_american(M, model=model_l, is_put=(option_type == "put"), semantics=_SEM[semantics],
                                antithetic=antithetic, heston_scheme=heston_scheme, n_steps=int(n_steps), S0=S0,
                                K=K, r=r, sigma=sigma or 0.0, T=T, seed=seed, stream=stream, **hp)
        loc = out["local"]
        return PriceResult(price=out["price"], stderr=out["stderr"], std=out["std"], zero_prob=out["zero_prob"],
                           n_paths=out["n_paths"], n_exercised=out["n_exercised"], sum_nitm=out["sum_nitm"],
                           model=model_l, semantics=semantics, option_type=option_type,
                           timings_ms=dict(paths=loc["ms_paths"], lsm=loc["ms_lsm"], total=loc["ms_total"]),
                           info=dict(n_gpus=n_gpus, rank=sp.rank, transport=sp.transport))
    c = ctx or _ffi.default_context(device)
    p = _ffi.make_params(model=model_l, is_put=(option_type == "put"), semantics=_SEM[semantics],
                         antithetic=antithetic, heston_scheme=heston_scheme, n_paths=M,
                         n_steps=int(n_steps), S0=S0, K=K, r=r, sigma=sigma or 0.0, T=T,
                         seed=seed, stream=stream, **hp)
    out = c.price_american(p)
    var = max(out["sumsq"] / M - out["price"] ** 2, 0.0)
    return PriceResult(price=out["price"], stderr=math.sqrt(var / M), std=out["std"],
                       zero_prob=out["zero_prob"], n_paths=M, n_exercised=out["n_exercised"],
                       sum_nitm=out["sum_nitm"], model=model_l, semantics=semantics,
                       option_type=option_type,
                       timings_ms=dict(paths=out["ms_paths"], lsm=out["ms_lsm"], total=out["ms_total"]))


_job = {}


def _in_job(n_gpus: int) -> bool:
    """Is this process one rank of an n_gpus-rank job (started by a launcher that set the rank environment)?
    A job of ANOTHER size is an error, not a reason to start ranks from inside a rank."""
    if "RANK" not in os.environ:
        return False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == n_gpus:
        return True
    if world > 1:
        raise RuntimeError(
            f"price_american_option(n_gpus={n_gpus}) was called inside a {world}-rank job (RANK="
            f"{os.environ['RANK']}): n_gpus must equal the job's WORLD_SIZE.  It does not fall back to one GPU.")
    return False


def _job_pricer(n_gpus: int, device=None):
    """The per-process RcclPricer of an n_gpus-rank job (created on first use, closed at exit)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n_gpus or "RANK" not in os.environ:
        raise RuntimeError(
            f"this process is not a rank of a {n_gpus}-rank job (WORLD_SIZE={os.environ.get('WORLD_SIZE', 'unset')}, "
            f"RANK={os.environ.get('RANK', 'unset')}); it does not fall back to one GPU.")
    key = (os.getpid(), n_gpus)
    sp = _job.get(key)
    if sp is None:
        import atexit

        from .dist import RcclPricer
        rank = int(os.environ["RANK"])
        if isinstance(device, (list, tuple)):  # one device per rank
            device = device[rank]
        local = int(os.environ.get("LOCAL_RANK", rank)) if device is None else int(device)
        sp = RcclPricer(local, rank, world)
        _job[key] = sp
        atexit.register(sp.close)
    return sp


def price_european_option(S0, K, r, sigma, T, n_paths, n_steps=1, model="GBM", option_type="put",
                          heston_params=None, heston_scheme="reference", antithetic=True, seed=42,
                          stream=0, device=0, ctx=None) -> PriceResult:
    """Discounted terminal payoff mean; no path matrix is stored (options_model_3.py:382-437)."""
    model_l = str(model).lower()
    _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
    M = int(n_paths) // 2 * 2 if antithetic else int(n_paths)
    hp = heston_defaults(sigma, heston_params)
    c = ctx or _ffi.default_context(device)
    p = _ffi.make_params(model=model_l, is_put=(option_type == "put"), antithetic=antithetic,
                         heston_scheme=heston_scheme, n_paths=M, n_steps=int(n_steps), S0=S0, K=K,
                         r=r, sigma=sigma or 0.0, T=T, seed=seed, stream=stream, **hp)
    out = c.price_european(p)
    var = max(out["sumsq"] / M - out["price"] ** 2, 0.0)
    return PriceResult(price=out["price"], stderr=math.sqrt(var / max(M - 1, 1)), std=out["std"],
                       zero_prob=out["zero_prob"], n_paths=M, n_exercised=0, sum_nitm=0,
                       model=model_l, semantics="european", option_type=option_type,
                       timings_ms=dict(total=out["ms_total"]))
