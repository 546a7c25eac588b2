#!/usr/bin/env python3
"""Capture golden fixtures for the PER-STEP control flow from the reference's own v1 / v2 pricers.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's
`Options_model.py` (v1, functional API) and `options_model_2.py` (v2, OptionPricer) with an empty
`yfinance` stub, runs the REAL `price_american_option` of each under an observation hook, and writes
numeric fixtures only (tests/golden/per_step_ref.npz): no reference source text is stored.

Hook (observation only, the reference code runs unmodified): `ContNet` is replaced by a subclass
whose forward hook records every no-grad forward -- that is the `continuation` vector of
Options_model.py:141-142 / options_model_2.py:300-301, one per time step that had a non-empty
regression set.

For every captured run the per-step loop (Options_model.py:108-157, options_model_2.py:278-313) is
then REPLAYED here with the recorded continuation values on the reference's own paths, and the
replay is asserted to reproduce what the reference returned bit for bit -- (mean, std, zero_prob)
for v1, mean for v2.  What is stored per run:
  S          float64 [N+1][M]   the reference's path matrix (regenerated from its global-RNG seed)
  cont       float32 [N+1][M]   recorded continuation value of path j at step t (NaN: not evaluated)
  cf, ex     final cash-flows (valued at t = dt) and sticky exercise flags
  stats      (mean, std(ddof=0), P(cf == 0))
This pins sticky mask, discount order, strict '>' and the returned statistics of the per-step flow
to a run of the reference itself (tests/test_oracle_golden.py, tests/test_gpu_parity.py).

Also captured: a Heston *put* run of the v3 pricer (frozen-network fixture, same recipe as
tools/capture_golden.py's G4) -> tests/golden/v3_frozen_nn_heston_put.npz, so that Heston exercise
decisions are pinned too (an American call on a non-dividend asset never exercises).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_per_step.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
REF_ROOT = "/root/reference"


def import_v1_v2():
    sys.modules.setdefault("yfinance", types.ModuleType("yfinance"))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    import Options_model as v1  # noqa
    import options_model_2 as v2  # noqa
    return v1, v2


def payoff(S, K, is_put):
    return np.maximum(K - S, 0) if is_put else np.maximum(S - K, 0)


def spy_contnet(mod, log):
    """Replace mod.ContNet by a recording subclass; returns the original class."""
    import torch
    Orig = mod.ContNet

    class Spy(Orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)

            def hook(m, inp, outp):
                if not torch.is_grad_enabled():
                    log.append(outp.detach().clone().cpu().numpy().flatten())

            self.register_forward_hook(hook)

    mod.ContNet = Spy
    return Orig


def replay(S, conts, K, r, T, is_put):
    """The per-step loop with recorded continuation values (same statements, same order, as
    Options_model.py:108-157).  Returns cf, ex, dense continuation matrix."""
    N, M = S.shape[0] - 1, S.shape[1]
    dt = T / N
    cashflows = payoff(S[-1], K, is_put)
    exercised = np.zeros(M, dtype=bool)
    discount = np.exp(-r * dt)
    dense = np.full((N + 1, M), np.nan, np.float32)
    ci = 0
    for t in range(N - 1, 0, -1):
        cashflows *= discount
        itm = (payoff(S[t], K, is_put) > 0) & (~exercised)
        if not np.any(itm):
            continue
        X = S[t, itm]
        continuation = conts[ci]
        ci += 1
        assert continuation.shape == X.shape and continuation.dtype == np.float32
        dense[t, itm] = continuation
        immediate = payoff(X, K, is_put)
        to_exercise = immediate > continuation
        idx_itm = np.where(itm)[0]
        ex_idx = idx_itm[to_exercise]
        cashflows[ex_idx] = immediate[to_exercise]
        exercised[ex_idx] = True
    assert ci == len(conts), (ci, len(conts))
    return cashflows, exercised, dense


def v1_paths(S0, r, sigma, T, M, N, seed):
    """Options_model.py:74-88 replayed (global numpy RNG, antithetic halves)."""
    np.random.seed(seed)
    dt = T / N
    M = M // 2 * 2
    drift = (r - 0.5 * sigma ** 2) * dt
    diffusion = sigma * np.sqrt(dt)
    Z = np.random.standard_normal((N, M // 2))
    Z = np.concatenate([Z, -Z], axis=1)
    stock = np.zeros((N + 1, M))
    stock[0] = S0
    for t in range(1, N + 1):
        stock[t] = stock[t - 1] * np.exp(drift + diffusion * Z[t - 1])
    return stock


def capture_v1(v1, out, tag, option_type, M, N, S0=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42):
    log = []
    orig = spy_contnet(v1, log)
    try:
        mean, std, zero = v1.price_american_option(S0, K, T, r, sigma, M, N, option_type, 2, False, seed)
    finally:
        v1.ContNet = orig
    is_put = option_type == "put"
    S = v1_paths(S0, r, sigma, T, M, N, seed)
    cf, ex, dense = replay(S, log, K, r, T, is_put)
    got = (cf.mean(), cf.std(), np.mean(cf == 0))
    assert got == (mean, std, zero), (got, (mean, std, zero))
    store(out, tag, S, dense, cf, ex, got, [S0, K, T, r, sigma, float(is_put), float(seed)])
    print(f"[v1 {tag}] mean={mean!r} std={std!r} zero_prob={zero!r} steps with a fit={len(log)} "
          f"exercised={ex.mean():.3f}")


def capture_v2(v2, out, tag, option_type, M, N, use_heston, S0=100.0, K=100.0, T=1.0, r=0.05, sigma=0.2, seed=42,
               nn_hidden=32, nn_epochs=10):
    hp = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7) if use_heston else None
    log = []
    orig = spy_contnet(v2, log)
    try:
        pricer = v2.OptionPricer(K, r, sigma, option_type, 2, seed, use_heston, hp, nn_hidden, nn_epochs)
        mean = pricer.price_american_option(S0, T, M, N)
    finally:
        v2.ContNet = orig
    is_put = option_type == "put"
    M = M // 2 * 2
    if use_heston:
        S = v2.simulate_heston_paths(S0, r, T, hp["v0"], hp["kappa"], hp["theta"], hp["xi"], hp["rho"], M, N, seed)
    else:
        S = v1_paths(S0, r, sigma, T, M, N, seed)  # options_model_2.py:257-264 is the same recurrence
    cf, ex, dense = replay(S, log, K, r, T, is_put)
    assert cf.mean() == mean, (cf.mean(), mean)
    store(out, tag, S, dense, cf, ex, (cf.mean(), cf.std(), np.mean(cf == 0)),
          [S0, K, T, r, sigma, float(is_put), float(seed)])
    print(f"[v2 {tag}] mean={mean!r} heston={use_heston} steps with a fit={len(log)} exercised={ex.mean():.3f}")


def store(out, tag, S, dense, cf, ex, stats, params):
    out[f"{tag}_S"] = S
    out[f"{tag}_cont"] = dense
    out[f"{tag}_cf"] = cf
    out[f"{tag}_ex"] = ex
    out[f"{tag}_stats"] = np.array(stats, np.float64)
    out[f"{tag}_params"] = np.array(params, np.float64)  # S0 K T r sigma is_put seed


def main():
    import torch
    torch.set_num_threads(8)
    v1, v2 = import_v1_v2()
    os.makedirs(OUT, exist_ok=True)
    out = {}
    capture_v1(v1, out, "v1_put", "put", 2048, 20)
    capture_v1(v1, out, "v1_call", "call", 1024, 12, S0=100.0, K=95.0)
    capture_v1(v1, out, "v1_put_odd", "put", 515, 9, S0=95.0, K=100.0, T=0.5, seed=7)  # odd M: one path dropped
    capture_v2(v2, out, "v2_put", "put", 1024, 16, False, nn_hidden=16, nn_epochs=5)
    capture_v2(v2, out, "v2_heston_put", "put", 2048, 20, True)
    np.savez_compressed(os.path.join(OUT, "per_step_ref.npz"), **out)

    # Heston PUT through the v3 pricer (frozen-network fixture): real Heston exercise decisions
    sys.path.insert(0, HERE)
    import capture_golden as cg
    om = cg.import_reference()
    nn = {}
    cg.capture_frozen_nn(om, nn, "put", True, "heston_put", M=1024, N=25, hidden=64, epochs=10)
    np.savez_compressed(os.path.join(OUT, "v3_frozen_nn_heston_put.npz"), **nn)
    for f in ("per_step_ref.npz", "v3_frozen_nn_heston_put.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
