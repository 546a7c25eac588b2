#!/usr/bin/env python3
"""Capture golden fixtures from the reference's importable CPU pricer.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's `options_model_3/options_model_3.py` (with an empty `yfinance` stub:
the module is only touched by network fetchers that are never called), drives the
real `AdvancedOptionPricer.price_american_enhanced_lsm` under observation hooks and
writes *numeric* fixtures (inputs + expected outputs) to tests/golden/.  No
reference source text is stored.

Hooks used (observation only, reference code runs unmodified):
  * RNGManager subclass whose child generator records every standard_normal draw
    -> the exact normals Z the reference consumed (options_model_3.py:475, :223-224)
  * SingleLSMNet subclass that remembers the instance + a forward hook that records
    every no-grad forward (= pass-2 continuation calls, options_model_3.py:637-640)

Fixture groups: G1 GBM paths from Z, G2 Heston paths from Z,
G3 regression features, G4 frozen-regressor pass-2 decisions, G5 polynomial flows
(independent numpy lstsq restatement, cross-checked against SURVEY anchors), G6
end-to-end scalars, G7 Welford merge, G8 Black-Scholes closed form.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden.py [--slow]
        (--slow also re-runs the 10k x 50 end-to-end reference pricings, ~12 min)
"""
import argparse
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/options_model_3"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.modules.setdefault("yfinance", types.ModuleType("yfinance"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import options_model_3 as om  # noqa
    return om


class RecordingGenerator:
    """Proxy around numpy Generator that logs standard_normal draws."""

    def __init__(self, gen, log):
        self._gen = gen
        self._log = log

    def standard_normal(self, *a, **k):
        z = self._gen.standard_normal(*a, **k)
        self._log.append(np.array(z, copy=True))
        return z

    def __getattr__(self, name):
        return getattr(self._gen, name)


def make_recording_rng_manager(om, seed, log, seeds_log):
    class RecMgr(om.RNGManager):
        def get_child_rng(self):
            child_seed = self.master_rng.integers(0, 2**31 - 1)
            seeds_log.append(int(child_seed))
            return RecordingGenerator(np.random.default_rng(child_seed), log)

        def get_child_seed(self):
            s = self.master_rng.integers(0, 2**31 - 1)
            seeds_log.append(int(s))
            return s

    return RecMgr(seed)


# ----------------------------------------------------------------------------
# Independent numpy restatement of the flows with a linear regressor (lstsq).
# This is NOT the repo's oracle: it exists so the oracle has something other than
# itself to be checked against (fixture group G5).
# ----------------------------------------------------------------------------
def payoff(S, K, is_put):
    return np.maximum(K - S, 0.0) if is_put else np.maximum(S - K, 0.0)


def poly_fit_eval(x, y):
    """OLS of y on [1,u,u^2], u=x-1, degree reduced to n-1 for n<3; in-sample fit."""
    n = x.size
    u = x - 1.0
    deg = min(2, n - 1)
    A = np.stack([u**k for k in range(deg + 1)], axis=1)
    beta, *_ = np.linalg.lstsq(A, y, rcond=None)
    b = np.zeros(3)
    b[: deg + 1] = beta
    return b, A @ beta


def flow_per_step(S, K, r, T, is_put, textbook):
    """v1/v2 per-step control flow (Options_model.py:108-157) with the net replaced by
    OLS on [1,u,u^2].  textbook=True: classic Longstaff-Schwartz (no sticky mask,
    overwrite on earlier exercise, discount to t=0)."""
    N = S.shape[0] - 1
    M = S.shape[1]
    dt = T / N
    disc = np.exp(-r * dt)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    ex = np.zeros(M, bool)
    betas = np.zeros((N + 1, 3))
    nitm = np.zeros(N + 1, np.int64)
    for t in range(N - 1, 0, -1):
        cf *= disc
        pay = payoff(S[t], K, is_put)
        itm = pay > 0
        if not textbook:
            itm &= ~ex
        if not itm.any():
            continue
        x = S[t, itm] / K
        b, cont = poly_fit_eval(x, cf[itm])
        betas[t] = b
        nitm[t] = itm.sum()
        imm = pay[itm]
        doex = imm > cont
        idx = np.where(itm)[0][doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    if textbook:
        cf = cf * disc
    return cf, ex, betas, nitm


def flow_two_pass_poly(S, K, r, T, is_put):
    """v3 two-pass control flow (options_model_3.py:482-651): pass 1 collects
    (x, discounted TERMINAL payoff) for every ITM (t, path) with no decisions; the
    regressor here is one OLS on [1,u,u^2] per time step; pass 2 applies the sticky
    rule with those fits."""
    N = S.shape[0] - 1
    M = S.shape[1]
    dt = T / N
    disc = np.exp(-r * dt)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    betas = np.zeros((N + 1, 3))
    nitm = np.zeros(N + 1, np.int64)
    for t in range(N - 1, 0, -1):
        cf *= disc
        itm = payoff(S[t], K, is_put) > 0
        if not itm.any():
            continue
        b, _ = poly_fit_eval(S[t, itm] / K, cf[itm])
        betas[t] = b
        nitm[t] = itm.sum()
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    ex = np.zeros(M, bool)
    for t in range(N - 1, 0, -1):
        cf *= disc
        pay = payoff(S[t], K, is_put)
        itm = (pay > 0) & ~ex
        if not itm.any() or nitm[t] == 0:
            continue
        u = S[t, itm] / K - 1.0
        cont = betas[t, 0] + betas[t, 1] * u + betas[t, 2] * u * u
        imm = pay[itm]
        doex = imm > cont
        idx = np.where(itm)[0][doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    return cf, ex, betas, nitm


def flow_two_pass_ols7(om, S, K, r, T, is_put):
    """v3 flow with ONE global OLS on the reference's 7 features (normalised exactly as
    options_model_3.py:550-563, min-norm lstsq so the zeroed constant column gets 0)."""
    N = S.shape[0] - 1
    M = S.shape[1]
    dt = T / N
    disc = np.exp(-r * dt)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    F, Y = [], []
    for t in range(N - 1, 0, -1):
        cf *= disc
        itm = payoff(S[t], K, is_put) > 0
        if not itm.any():
            continue
        F.append(om.create_regression_features(S[t, itm], K, r, T, t * dt))
        Y.append(cf[itm].reshape(-1, 1))
    X_all = np.vstack(F)
    Y_all = np.vstack(Y)
    Y_mean, Y_std = Y_all.mean(), Y_all.std()
    fm, fs = X_all.mean(axis=0), X_all.std(axis=0)
    fs[fs == 0] = 1
    w, *_ = np.linalg.lstsq((X_all - fm) / fs, (Y_all - Y_mean) / Y_std, rcond=None)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    ex = np.zeros(M, bool)
    for t in range(N - 1, 0, -1):
        cf *= disc
        pay = payoff(S[t], K, is_put)
        itm = (pay > 0) & ~ex
        if not itm.any():
            continue
        f = om.create_regression_features(S[t, itm], K, r, T, t * dt)
        cont = (((f - fm) / fs) @ w).ravel() * Y_std + Y_mean
        imm = pay[itm]
        doex = imm > cont
        idx = np.where(itm)[0][doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    return cf, ex, dict(w=w.ravel(), feat_mean=fm, feat_std=fs, Y_mean=Y_mean, Y_std=Y_std,
                        R=X_all.shape[0])


def gbm_from_zhalf(z_half, S0, r, sigma, T):
    """Replay of options_model_3.py:473-480 on recorded normals."""
    N, P = z_half.shape
    dt = T / N
    drift = (r - 0.5 * sigma**2) * dt
    diffusion = sigma * np.sqrt(dt)
    Z = np.concatenate([z_half, -z_half], axis=1)
    S = np.zeros((N + 1, 2 * P))
    S[0] = S0
    for t in range(1, N + 1):
        S[t] = S[t - 1] * np.exp(drift + diffusion * Z[t - 1])
    return S


# ----------------------------------------------------------------------------
def capture_gbm_paths(om, out):
    """G1: exact Z the reference draws for RNGManager(42) and the S it builds."""
    for tag, (M, N) in {"small": (16, 8), "mid": (1024, 50)}.items():
        mgr = om.RNGManager(42)
        rng = mgr.get_child_rng()
        mgr.get_child_seed()
        z_half = rng.standard_normal((N, M // 2))
        S = gbm_from_zhalf(z_half, 100.0, 0.05, 0.2, 1.0)
        out[f"gbm_{tag}_zhalf"] = z_half
        out[f"gbm_{tag}_S"] = S
    out["gbm_params"] = np.array([100.0, 0.05, 0.2, 1.0])  # S0 r sigma T


def capture_heston_paths(om, out):
    """G2: simulate_heston_paths_antithetic (options_model_3.py:211-251) run for real,
    with the per-step z1_half/z2_half it drew."""
    sets = {
        "feller": dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7),
        "clamp": dict(v0=0.04, kappa=2.0, theta=0.04, xi=1.0, rho=-0.7),  # violates Feller
    }
    for pname, hp in sets.items():
        for tag, (M, N) in {"small": (16, 8), "mid": (1024, 50)}.items():
            log = []
            rng = RecordingGenerator(np.random.default_rng(1234), log)
            S = om.simulate_heston_paths_antithetic(100.0, 0.05, 1.0, hp["v0"], hp["kappa"],
                                                    hp["theta"], hp["xi"], hp["rho"], M, N, rng)
            z1 = np.stack(log[0::2])
            z2 = np.stack(log[1::2])
            assert z1.shape == (N, M // 2)
            out[f"heston_{pname}_{tag}_z1"] = z1
            out[f"heston_{pname}_{tag}_z2"] = z2
            out[f"heston_{pname}_{tag}_S"] = S
        out[f"heston_{pname}_params"] = np.array([100.0, 0.05, 1.0, hp["v0"], hp["kappa"],
                                                  hp["theta"], hp["xi"], hp["rho"]])


def capture_features(om, out):
    """G3: create_regression_features (options_model_3.py:105-121)."""
    S = np.linspace(60.0, 140.0, 32)
    K, r, T = 100.0, 0.05, 1.0
    tcur = np.array([0.02, 0.5, 1.0 - 1e-7])  # last one hits the 1e-6 floor
    out["feat_S"] = S
    out["feat_KrT"] = np.array([K, r, T])
    out["feat_tcur"] = tcur
    out["feat_out"] = np.stack([om.create_regression_features(S, K, r, T, t) for t in tcur])


def capture_frozen_nn(om, out, option_type, use_heston, tag, M=1024, N=50, hidden=128, epochs=25):
    """G4: run the REAL price_american_enhanced_lsm with observation hooks."""
    import torch

    zlog, seeds = [], []
    mgr = make_recording_rng_manager(om, 42, zlog, seeds)
    nets, fwd = [], []

    Orig = om.SingleLSMNet

    class Spy(Orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            nets.append(self)

            def hook(mod, inp, outp):
                if not torch.is_grad_enabled():
                    if not fwd:  # weights as they are when pass 2 starts
                        fwd.append({k_: v.detach().clone() for k_, v in mod.state_dict().items()})
                    fwd.append((inp[0].detach().clone().numpy(), outp.detach().clone().numpy()))

            self.register_forward_hook(hook)

    om.SingleLSMNet = Spy
    hp = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7) if use_heston else None
    K, r, sigma, S0, T = 100.0, 0.05, 0.2, 100.0, 1.0
    try:
        pricer = om.AdvancedOptionPricer(K=K, r=r, sigma=sigma, option_type=option_type,
                                         rng_manager=mgr, use_heston=use_heston, heston_params=hp,
                                         nn_hidden=hidden, nn_epochs=epochs,
                                         use_control_variate=False)
        price = pricer.price_american_option(S0, T, M, N)
    finally:
        om.SingleLSMNet = Orig
    is_put = option_type == "put"

    # paths the reference used
    if use_heston:
        z1 = np.stack(zlog[0::2])
        z2 = np.stack(zlog[1::2])
        S = om.simulate_heston_paths_antithetic(S0, r, T, hp["v0"], hp["kappa"], hp["theta"],
                                                hp["xi"], hp["rho"], M, N,
                                                types.SimpleNamespace(
                                                    standard_normal=iter_normals([z for z in zlog])))
        out[f"{tag}_z1"] = z1
        out[f"{tag}_z2"] = z2
    else:
        z_half = zlog[0]
        S = gbm_from_zhalf(z_half, S0, r, sigma, T)
        out[f"{tag}_zhalf"] = z_half

    state = fwd[0]
    calls = fwd[1:]
    net = nets[0]

    # restate pass 1 normalisers and check them against what the net was actually fed
    dt = T / N
    disc = np.exp(-r * dt)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    F, Y = [], []
    for t in range(N - 1, 0, -1):
        cf *= disc
        itm = payoff(S[t], K, is_put) > 0
        if not itm.any():
            continue
        F.append(om.create_regression_features(S[t, itm], K, r, T, t * dt))
        Y.append(cf[itm].reshape(-1, 1))
    X_all, Y_all = np.vstack(F), np.vstack(Y)
    Y_mean, Y_std = Y_all.mean(), Y_all.std()
    fm, fs = X_all.mean(axis=0), X_all.std(axis=0)
    fs[fs == 0] = 1

    # replay pass 2 with the recorded (dropout-on) outputs; must reproduce `price` exactly
    net.eval()
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    cf_eval = cf.copy()
    ex = np.zeros(M, bool)
    ex_eval = np.zeros(M, bool)
    ci = 0
    cont_eval_steps = np.full((N + 1, M), np.nan, np.float32)
    cont_drop_steps = np.full((N + 1, M), np.nan, np.float32)
    for t in range(N - 1, 0, -1):
        cf *= disc
        cf_eval *= disc
        pay = payoff(S[t], K, is_put)
        itm = (pay > 0) & ~ex
        if itm.any():
            f = om.create_regression_features(S[t, itm], K, r, T, t * dt)
            fn = ((f - fm) / fs).astype(np.float32)
            xin, yout = calls[ci]
            ci += 1
            assert np.array_equal(fn, xin), f"feature restatement mismatch at t={t}"
            cont = yout.flatten() * Y_std + Y_mean
            cont_drop_steps[t, itm] = yout.flatten()
            imm = pay[itm]
            doex = imm > cont
            idx = np.where(itm)[0][doex]
            cf[idx] = imm[doex]
            ex[idx] = True
        itm_e = (pay > 0) & ~ex_eval
        if itm_e.any():
            f = om.create_regression_features(S[t, itm_e], K, r, T, t * dt)
            fn = ((f - fm) / fs).astype(np.float32)
            with torch.no_grad():
                yo = net(torch.from_numpy(fn)).numpy().flatten()
            cont_eval_steps[t, itm_e] = yo
            cont = yo * Y_std + Y_mean
            imm = pay[itm_e]
            doex = imm > cont
            idx = np.where(itm_e)[0][doex]
            cf_eval[idx] = imm[doex]
            ex_eval[idx] = True
    assert ci == len(calls)
    assert cf.mean() == price, (cf.mean(), price)

    out[f"{tag}_S"] = S
    out[f"{tag}_price_ref"] = np.float64(price)
    out[f"{tag}_cf_ref"] = cf
    out[f"{tag}_ex_ref"] = ex
    out[f"{tag}_cont_scaled_dropout"] = cont_drop_steps
    out[f"{tag}_price_eval"] = np.float64(cf_eval.mean())
    out[f"{tag}_cf_eval"] = cf_eval
    out[f"{tag}_ex_eval"] = ex_eval
    out[f"{tag}_cont_scaled_eval"] = cont_eval_steps
    out[f"{tag}_feat_mean"] = fm
    out[f"{tag}_feat_std"] = fs
    out[f"{tag}_Y_mean_std"] = np.array([Y_mean, Y_std])
    out[f"{tag}_R"] = np.int64(X_all.shape[0])
    out[f"{tag}_params"] = np.array([S0, K, r, sigma, T, float(is_put), float(hidden)])
    for k_, v in state.items():
        out[f"{tag}_sd_{k_}"] = v.numpy()
    print(f"[G4 {tag}] price_ref={price!r} price_eval={cf_eval.mean()!r} R={X_all.shape[0]}")


def iter_normals(arrs):
    it = iter(arrs)

    def f(*a, **k):
        return next(it)

    return f


def capture_poly_flows(om, out, scalars):
    """G5: polynomial flows on the reference's own seed-42 paths."""
    K, r, T = 100.0, 0.05, 1.0
    for tag, (M, N) in {"small": (16, 8), "mid": (1024, 50)}.items():
        mgr = om.RNGManager(42)
        z_half = mgr.get_child_rng().standard_normal((N, M // 2))
        S = gbm_from_zhalf(z_half, 100.0, 0.05, 0.2, 1.0)
        for is_put in (True, False):
            pc = "put" if is_put else "call"
            for name, fn in (("ref", lambda: flow_per_step(S, K, r, T, is_put, False)),
                             ("textbook", lambda: flow_per_step(S, K, r, T, is_put, True)),
                             ("twopass", lambda: flow_two_pass_poly(S, K, r, T, is_put))):
                cf, ex, betas, nitm = fn()
                out[f"poly_{tag}_{pc}_{name}_cf"] = cf
                out[f"poly_{tag}_{pc}_{name}_ex"] = ex
                out[f"poly_{tag}_{pc}_{name}_betas"] = betas
                out[f"poly_{tag}_{pc}_{name}_nitm"] = nitm
        if tag == "mid":
            cf, ex, info = flow_two_pass_ols7(om, S, K, r, T, True)
            out["ols7_mid_put_cf"] = cf
            out["ols7_mid_put_ex"] = ex
            for k_, v in info.items():
                out[f"ols7_mid_put_{k_}"] = np.asarray(v)

    # C1-size anchors: regenerate the 10k x 50 seed-42 paths
    M, N = 10000, 50
    mgr = om.RNGManager(42)
    child_seed_probe = om.RNGManager(42).master_rng.integers(0, 2**31 - 1)
    z_half = mgr.get_child_rng().standard_normal((N, M // 2))
    S = gbm_from_zhalf(z_half, 100.0, 0.05, 0.2, 1.0)
    c1 = {}
    cf, ex, betas, nitm = flow_per_step(S, K, r, T, True, False)
    c1["poly_ref_price"] = float(cf.mean())
    c1["poly_ref_sum_nitm"] = int(nitm.sum())
    c1["poly_ref_exercised_frac"] = float(ex.mean())
    cf, ex, betas, nitm = flow_per_step(S, K, r, T, True, True)
    c1["poly_textbook_price"] = float(cf.mean())
    c1["poly_textbook_stderr"] = float(cf.std(ddof=1) / np.sqrt(M))
    cf, ex, betas, nitm = flow_two_pass_poly(S, K, r, T, True)
    c1["poly_twopass_price"] = float(cf.mean())
    c1["poly_twopass_sum_nitm_pass1"] = int(nitm.sum())
    cf, ex, info = flow_two_pass_ols7(om, S, K, r, T, True)
    c1["ols7_twopass_price"] = float(cf.mean())
    c1["R"] = int(info["R"])
    c1["Y_mean"] = float(info["Y_mean"])
    c1["Y_std"] = float(info["Y_std"])
    c1["european_on_paths"] = float((payoff(S[-1], K, True) * np.exp(-r * T)).mean())
    c1["child_seed"] = int(child_seed_probe)
    c1["zhalf_sum"] = float(z_half.sum())
    c1["zhalf_sumsq"] = float((z_half**2).sum())
    c1["zhalf_first4"] = [float(v) for v in z_half.ravel()[:4]]
    c1["S_T_sum"] = float(S[-1].sum())
    scalars["c1_seed42_gbm_put"] = c1
    print("[G5 C1 anchors]", json.dumps(c1, indent=1))


def capture_welford(om, out):
    """G7: welford_batch_update / monte_carlo_price_streaming (options_model_3.py:33-63)."""
    rng = np.random.default_rng(7)
    data = rng.standard_normal(5000) * 3.0 + 1.5
    sizes = [1, 7, 500, 500, 1234, 2758]
    mean, m2, n = 0.0, 0.0, 0
    trace = []
    o = 0
    for s in sizes:
        mean, m2, n = om.welford_batch_update(mean, m2, n, data[o:o + s])
        o += s
        trace.append([mean, m2, n])
    it = iter(np.split(data, np.cumsum([500] * 9)))
    res = om.monte_carlo_price_streaming(lambda b: next(it), 5000, 500)
    out["welford_data"] = data
    out["welford_sizes"] = np.array(sizes)
    out["welford_trace"] = np.array(trace)
    out["welford_streaming_result"] = np.array(res, dtype=np.float64)


def capture_scalars(om, scalars, slow):
    bs = om.BlackScholesGreeks.black_scholes_price
    scalars["black_scholes"] = {
        "put_100_100_1_0.05_0.2": float(bs(100.0, 100.0, 1.0, 0.05, 0.2, "put")),
        "call_100_100_1_0.05_0.2": float(bs(100.0, 100.0, 1.0, 0.05, 0.2, "call")),
        "put_90_100_0.25_0.03_0.35": float(bs(90.0, 100.0, 0.25, 0.03, 0.35, "put")),
        "call_120_100_2_0.01_0.15": float(bs(120.0, 100.0, 2.0, 0.01, 0.15, "call")),
    }
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors; ctr[4], key[2] -> out[4])
    scalars["philox4x32_10_kat"] = [
        {"ctr": ["00000000"] * 4, "key": ["00000000"] * 2,
         "out": ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]},
        {"ctr": ["ffffffff"] * 4, "key": ["ffffffff"] * 2,
         "out": ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]},
        {"ctr": ["243f6a88", "85a308d3", "13198a2e", "03707344"], "key": ["a4093822", "299f31d0"],
         "out": ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]},
    ]
    # RNGManager child-seed sequence (options_model_3.py:69-79)
    m = om.RNGManager(42)
    scalars["rng_manager_42_child_seeds"] = [int(m.get_child_seed()) for _ in range(6)]
    if slow:
        import time
        e2e = {}
        for name, kw, args in [
            ("gbm_put_cv_off", dict(option_type="put", use_control_variate=False), {}),
            ("gbm_put_cv_on", dict(option_type="put", use_control_variate=True), {}),
            ("heston_call_cv_off", dict(option_type="call", use_control_variate=False,
                                        use_heston=True,
                                        heston_params=dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3,
                                                           rho=-0.7)), {}),
        ]:
            t0 = time.time()
            p = om.AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, rng_manager=om.RNGManager(42), **kw)
            e2e[name] = float(p.price_american_option(100.0, 1.0, 10000, 50))
            e2e[name + "_seconds"] = time.time() - t0
            print("[G6]", name, e2e[name], f"{time.time()-t0:.0f}s", flush=True)
        scalars["end_to_end_10k_x_50_seed42"] = e2e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slow", action="store_true")
    a = ap.parse_args()
    import torch
    torch.set_num_threads(8)
    om = import_reference()
    os.makedirs(OUT, exist_ok=True)
    scalars = {}
    spath = os.path.join(OUT, "scalars.json")
    if os.path.exists(spath):
        scalars = json.load(open(spath))

    paths = {}
    capture_gbm_paths(om, paths)
    capture_heston_paths(om, paths)
    capture_features(om, paths)
    capture_welford(om, paths)
    np.savez_compressed(os.path.join(OUT, "paths_features.npz"), **paths)

    nn = {}
    capture_frozen_nn(om, nn, "put", False, "gbm_put")
    capture_frozen_nn(om, nn, "call", True, "heston_call", M=512, N=20, hidden=64, epochs=10)
    np.savez_compressed(os.path.join(OUT, "v3_frozen_nn.npz"), **nn)

    poly = {}
    capture_poly_flows(om, poly, scalars)
    np.savez_compressed(os.path.join(OUT, "poly_flows.npz"), **poly)

    capture_scalars(om, scalars, a.slow)
    json.dump(scalars, open(spath, "w"), indent=1, sort_keys=True)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
