"""numpy float64 restatement of the reference CPU flows.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Pinned against fixtures captured from the importable reference
(tools/capture_golden.py -> tests/golden/*.npz): bit-for-bit for the path recurrences,
features, normalisers and Welford merge; decision-for-decision for the sweeps.
"""
from __future__ import annotations

import math

import numpy as np


# -- RNG management (options_model_3/options_model_3.py:69-79) -------------------------
class RNGManager:
    """master PCG64; every child is default_rng(master.integers(0, 2**31-1))."""

    def __init__(self, master_seed: int = 42):
        self.master_rng = np.random.default_rng(master_seed)
        self.master_seed = master_seed

    def get_child_seed(self) -> int:
        return int(self.master_rng.integers(0, 2**31 - 1))

    def get_child_rng(self) -> np.random.Generator:
        return np.random.default_rng(self.get_child_seed())


# -- paths -------------------------------------------------------------------------------
def gbm_paths_from_normals(z_half, S0, r, sigma, T):
    """options_model_3.py:473-480: Z=[Z_half,-Z_half]; S[t]=S[t-1]*exp(drift+diff*Z[t-1])."""
    z_half = np.asarray(z_half, np.float64)
    N, P = z_half.shape
    dt = T / N
    drift = (r - 0.5 * sigma**2) * dt
    diffusion = sigma * np.sqrt(dt)
    Z = np.concatenate([z_half, -z_half], axis=1)
    S = np.zeros((N + 1, 2 * P), dtype=np.float64)
    S[0] = S0
    for t in range(1, N + 1):
        S[t] = S[t - 1] * np.exp(drift + diffusion * Z[t - 1])
    return S


def heston_paths_from_normals(z1_half, z2_half, S0, r, T, v0, kappa, theta, xi, rho, scheme=0):
    """options_model_3.py:211-233 (scheme 0: clamp before use AND at store).
    scheme 1: Lord et al. full truncation (the variance is carried unclamped)."""
    z1_half = np.asarray(z1_half, np.float64)
    z2_half = np.asarray(z2_half, np.float64)
    N, P = z1_half.shape
    dt = T / N
    S = np.zeros((N + 1, 2 * P))
    v = np.zeros(2 * P) + v0
    S[0] = S0
    for t in range(1, N + 1):
        z1 = np.concatenate([z1_half[t - 1], -z1_half[t - 1]])
        z2 = np.concatenate([z2_half[t - 1], -z2_half[t - 1]])
        w1 = z1
        w2 = rho * z1 + np.sqrt(1 - rho**2) * z2
        v_prev = np.maximum(v, 0)
        base = v_prev if scheme == 0 else v
        v_new = base + kappa * (theta - v_prev) * dt + xi * np.sqrt(v_prev * dt) * w2
        S[t] = S[t - 1] * np.exp((r - 0.5 * v_prev) * dt + np.sqrt(v_prev * dt) * w1)
        v = np.maximum(v_new, 0) if scheme == 0 else v_new
    return S


def heston_calibrator_paths_from_normals(z1, z2_indep, S0, r, T, v0, kappa, theta, xi, rho):
    """options_model_3/heston_calibration.py:204-257 on recorded normals: z1, z2_indep are
    [n_sim][n_steps]; antithetic stacking AFTER the rho-mix (:230-234); variance floored at 1e-8
    before use and at store; arithmetic Euler for S.  Returns S, V as [n_paths][n_steps+1]."""
    z1 = np.asarray(z1, np.float64)
    z2 = rho * z1 + np.sqrt(1 - rho**2) * np.asarray(z2_indep, np.float64)
    Z1 = np.vstack([z1, -z1])
    Z2 = np.vstack([z2, -z2])
    n_paths, N = Z1.shape
    dt = T / N
    S = np.zeros((n_paths, N + 1))
    V = np.zeros((n_paths, N + 1))
    S[:, 0], V[:, 0] = S0, v0
    sqrt_dt = np.sqrt(dt)
    for t in range(N):
        V_pos = np.maximum(V[:, t], 1e-8)
        sqrt_V = np.sqrt(V_pos)
        dV = kappa * (theta - V_pos) * dt + xi * sqrt_V * sqrt_dt * Z2[:, t]
        V[:, t + 1] = np.maximum(V_pos + dV, 1e-8)
        dS = r * S[:, t] * dt + sqrt_V * S[:, t] * sqrt_dt * Z1[:, t]
        S[:, t + 1] = S[:, t] + dS
    return S, V


def strike_prices(S_T, strikes, r, T, is_put=False):
    """heston_calibration.py:301-306: discounted mean payoff per strike."""
    S_T = np.asarray(S_T, np.float64)
    df = np.exp(-r * T)
    return np.array([df * np.mean(np.maximum(K - S_T, 0) if is_put else np.maximum(S_T - K, 0))
                     for K in strikes])


# -- payoff / features ------------------------------------------------------------------
def payoff(S, K, is_put):
    """options_model_3.py:376-380."""
    return np.maximum(K - S, 0) if is_put else np.maximum(S - K, 0)


def regression_features(S, K, T, t_current):
    """options_model_3.py:105-121: [1, x, x^2, x^3, max(x-1,0), s, x*s], x=S/K,
    s=sqrt(max(T-t,1e-6)); the reference's `r` argument is unused."""
    x = np.asarray(S, np.float64) / K
    s = float(np.sqrt(max(T - t_current, 1e-6)))
    sc = np.full_like(x, s)
    return np.column_stack([np.ones_like(x), x, x**2, x**3, np.maximum(x - 1, 0), sc, x * sc])


# -- Welford / Chan merge (options_model_3.py:33-63) --------------------------------------
def welford_batch_update(mean, m2, n, batch):
    batch = np.asarray(batch, np.float64)
    b_n = batch.size
    if b_n == 0:
        return mean, m2, n
    b_mean = batch.mean()
    b_m2 = ((batch - b_mean) ** 2).sum()
    delta = b_mean - mean
    new_n = n + b_n
    return mean + delta * (b_n / new_n), m2 + b_m2 + delta**2 * n * b_n / new_n, new_n


def streaming_stats(chunks):
    mean, m2, n = 0.0, 0.0, 0
    for c in chunks:
        mean, m2, n = welford_batch_update(mean, m2, n, c)
    var = m2 / (n - 1) if n > 1 else 0.0
    return mean, (math.sqrt(var / n) if n > 0 else 0.0), n


# -- Black-Scholes closed form (options_model_3.py:150-159) --------------------------------
def _ncdf(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def black_scholes_price(S, K, T, r, sigma, option_type="call"):
    d1 = (math.log(S / K) + (r + 0.5 * sigma**2) * T) / (sigma * math.sqrt(T))
    d2 = d1 - sigma * math.sqrt(T)
    if option_type == "call":
        return S * _ncdf(d1) - K * math.exp(-r * T) * _ncdf(d2)
    return K * math.exp(-r * T) * _ncdf(-d2) - S * _ncdf(-d1)


# -- backward sweeps -----------------------------------------------------------------------
def _fit_poly2(x, y):
    """OLS on [1,u,u^2], u=x-1, degree min(2,n-1); returns (beta[3], in-sample fit)."""
    n = x.size
    u = x - 1.0
    deg = min(2, n - 1)
    A = np.stack([u**k for k in range(deg + 1)], axis=1)
    beta, *_ = np.linalg.lstsq(A, y, rcond=None)
    b = np.zeros(3)
    b[: deg + 1] = beta
    return b, A @ beta


def lsm_per_step(S, K, r, T, is_put, textbook=False, fit=_fit_poly2, cont_values=None):
    """Per-step control flow of Options_model.py:108-157 / options_model_2.py:278-313 with
    the per-step net swapped for `fit`.  textbook=True is classic Longstaff-Schwartz.
    cont_values: dense [N+1][M] matrix of continuation values to use instead of a fit (the recorded
    ContNet outputs of a run of the reference: tests/golden/per_step_ref.npz)."""
    N, M = S.shape[0] - 1, S.shape[1]
    disc = np.exp(-r * T / N)
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
        if cont_values is not None:
            cont = cont_values[t, itm]
            nitm[t] = itm.sum()
        else:
            b, cont = fit(S[t, itm] / K, cf[itm])
            betas[t], nitm[t] = b, itm.sum()
        imm = pay[itm]
        doex = imm > cont
        idx = np.where(itm)[0][doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    if textbook:
        cf = cf * disc
    return cf, ex, betas, nitm


def lsm_per_step_contnet(S, K, r, T, is_put, hidden=32, epochs=10, lr=1e-3, init=None):
    """The per-step loop of Options_model.py:108-157 / options_model_2.py:278-313 WITH its regressor:
    a fresh ContNet(1 -> hidden -> hidden -> 1) per step (Options_model.py:14-25), inputs standardised by
    the set's own mean / population std (:124), raw cash-flows as targets, `epochs` full-batch Adam steps
    on the MSE (:129-139), continuation = net(inputs) in float32 (:141-142), strict > (:145).
    init(t) -> dict(w0 [h,1], b0 [h], w1 [h,h], b1 [h], w2 [1,h], b2 [1]) replaces torch's own
    initialisation (None: torch default, as the reference).  torch on the CPU, float32, like the
    reference without a GPU.  Returns cash-flows, exercised mask, set sizes, dense continuation values."""
    import torch
    import torch.nn as nn

    N, M = S.shape[0] - 1, S.shape[1]
    disc = np.exp(-r * T / N)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    ex = np.zeros(M, bool)
    nitm = np.zeros(N + 1, np.int64)
    cont_all = np.zeros((N + 1, M), np.float32)
    for t in range(N - 1, 0, -1):
        cf *= disc
        pay = payoff(S[t], K, is_put)
        itm = (pay > 0) & ~ex
        if not itm.any():
            continue
        X = S[t, itm].astype(np.float64)
        Y = cf[itm]
        Xs = (X - X.mean()) / X.std() if X.std() > 0 else X - X.mean()
        net = nn.Sequential(nn.Linear(1, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                            nn.Linear(hidden, 1))
        if init is not None:
            w = init(t)
            with torch.no_grad():
                for lin, (kw, kb) in zip((net[0], net[2], net[4]), (("w0", "b0"), ("w1", "b1"), ("w2", "b2"))):
                    lin.weight.copy_(torch.from_numpy(np.asarray(w[kw], np.float32)))
                    lin.bias.copy_(torch.from_numpy(np.asarray(w[kb], np.float32)))
        opt = torch.optim.Adam(net.parameters(), lr=lr)
        xt = torch.from_numpy(Xs.reshape(-1, 1)).float()
        yt = torch.from_numpy(Y.reshape(-1, 1)).float()
        for _ in range(epochs):
            loss = nn.MSELoss()(net(xt), yt)
            opt.zero_grad()
            loss.backward()
            opt.step()
        with torch.no_grad():
            cont = net(xt).numpy().flatten()
        nitm[t] = itm.sum()
        cont_all[t, itm] = cont
        imm = pay[itm]
        doex = imm > cont
        idx = np.where(itm)[0][doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    return cf, ex, nitm, cont_all


def lsm_two_pass(S, K, r, T, is_put, regress, predict):
    """v3 control flow, options_model_3.py:482-516 (pass 1: no decisions, targets are the
    discounted terminal payoff) and :615-651 (pass 2: sticky mask, strict >, valued at
    t=dt).  `regress(rows)` gets [(t, S_itm, Y_itm)] and returns a model; `predict(model,
    t, S_itm)` returns continuation values."""
    N, M = S.shape[0] - 1, S.shape[1]
    disc = np.exp(-r * T / N)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    rows = []
    for t in range(N - 1, 0, -1):
        cf *= disc
        itm = payoff(S[t], K, is_put) > 0
        if itm.any():
            rows.append((t, S[t, itm].copy(), cf[itm].copy()))
    if not rows:
        return cf, np.zeros(M, bool), None
    model = regress(rows)
    cf = payoff(S[-1], K, is_put).astype(np.float64)
    ex = np.zeros(M, bool)
    for t in range(N - 1, 0, -1):
        cf *= disc
        pay = payoff(S[t], K, is_put)
        itm = (pay > 0) & ~ex
        if not itm.any():
            continue
        cols = np.where(itm)[0]
        cont = predict(model, t, S[t, itm], cols) if getattr(predict, "wants_cols", False) else predict(model, t, S[t, itm])
        if cont is None:
            continue
        imm = pay[itm]
        doex = imm > cont
        idx = cols[doex]
        cf[idx] = imm[doex]
        ex[idx] = True
    return cf, ex, model


def two_pass_poly_regressor(K):
    def regress(rows):
        return {t: _fit_poly2(s / K, y)[0] for t, s, y in rows}

    def predict(model, t, s):
        if t not in model:
            return None
        b = model[t]
        u = s / K - 1.0
        return b[0] + b[1] * u + b[2] * u * u

    return regress, predict


def normalisers(rows, K, T, dt):
    """options_model_3.py:542-563: population std, zero std -> 1."""
    X_all = np.vstack([regression_features(s, K, T, t * dt) for t, s, _ in rows])
    Y_all = np.concatenate([y for _, _, y in rows]).reshape(-1, 1)
    Y_mean, Y_std = Y_all.mean(), Y_all.std()
    if not Y_std > 0:
        Y_std = 1.0
    fm, fs = X_all.mean(axis=0), X_all.std(axis=0)
    fs[fs == 0] = 1
    return X_all, Y_all, fm, fs, Y_mean, Y_std


def two_pass_ols7_regressor(K, T, N):
    dt = T / N

    def regress(rows):
        X_all, Y_all, fm, fs, Y_mean, Y_std = normalisers(rows, K, T, dt)
        w, *_ = np.linalg.lstsq((X_all - fm) / fs, (Y_all - Y_mean) / Y_std, rcond=None)
        return dict(w=w.ravel(), fm=fm, fs=fs, Y_mean=Y_mean, Y_std=Y_std, R=X_all.shape[0])

    def predict(m, t, s):
        f = regression_features(s, K, T, t * dt)
        return ((f - m["fm"]) / m["fs"]) @ m["w"] * m["Y_std"] + m["Y_mean"]

    return regress, predict


def mlp_forward(state, x):
    """SingleLSMNet in eval mode (options_model_3.py:85-103): Linear/ReLU stacks, float32."""
    h = np.asarray(x, np.float32)
    idx = sorted({int(k.split(".")[1]) for k in state if k.endswith("weight")})
    for n, li in enumerate(idx):
        W = state[f"net.{li}.weight"].astype(np.float32)
        b = state[f"net.{li}.bias"].astype(np.float32)
        h = h @ W.T + b
        if n + 1 < len(idx):
            h = np.maximum(h, 0)
    return h


def two_pass_frozen_mlp_regressor(K, T, N, state, fm, fs, Y_mean, Y_std, dropout=None):
    """Pass 2 with given (already trained) weights, float32 cast as options_model_3.py:638.
    dropout=None: the net in eval mode.  dropout=dict(p=, seed=, hidden=, layers=[, col_of=]): the net as the reference
    runs it (:637-640, no .eval(): nn.Dropout active), with the build's mask definition (oracle/dropout.py) keyed by
    (path column, time step); col_of maps a column of S to its column in the unsharded matrix (default: identity)."""
    dt = T / N

    def regress(rows):
        return None

    def predict(_, t, s, cols=None):
        f = regression_features(s, K, T, t * dt)
        fn = ((f - fm) / fs).astype(np.float32)
        if dropout is None:
            return mlp_forward(state, fn).ravel() * Y_std + Y_mean
        from . import dropout as dr
        key = cols if dropout.get("col_of") is None else dropout["col_of"](cols)
        masks = dr.apply_masks(dropout["hidden"], dropout["layers"], key, t, dropout["seed"], dropout["p"])
        return dr.mlp_forward_masked(state, fn, masks, dropout["p"]).ravel() * Y_std + Y_mean

    predict.wants_cols = dropout is not None
    return regress, predict
