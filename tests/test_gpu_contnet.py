"""The per-step ContNet regressor (omc_contnet.hip) -- what the reference's v1 / v2 pricers actually fit --
against (a) a torch-on-CPU restatement started from the SAME initial nets on the SAME paths and (b) the
recorded runs of the reference itself (tests/golden/per_step_ref.npz: its paths, its prices)."""
import numpy as np
import pytest

from oracle import reference_flow as rf

pytestmark = pytest.mark.gpu

TAGS = ["v1_put", "v1_call", "v1_put_odd", "v2_put", "v2_heston_put"]


def _case(golden, tag):
    g = golden["per_step"]
    S0, K, T, r, sig, is_put, seed = g[f"{tag}_params"]
    return g[f"{tag}_S"].astype(np.float32), float(K), float(r), float(T), bool(is_put), g[f"{tag}_stats"]


@pytest.mark.parametrize("tag", TAGS)
def test_contnet_flow_matches_the_restatement_from_the_same_initial_nets(ctx, golden, tag):
    S32, K, r, T, is_put, _ = _case(golden, tag)
    N, M = S32.shape[0] - 1, S32.shape[1]
    seed = 1234
    cf_o, ex_o, nitm_o, cont_o = rf.lsm_per_step_contnet(
        S32.astype(np.float64), K, r, T, is_put, hidden=32, epochs=10, lr=1e-3,
        init=lambda t: ctx.contnet_init_params(32, t, seed))
    Sd = ctx.to_device(S32)
    out = ctx.lsm_contnet(Sd, K, r, T, is_put, 32, 10, 1e-3, seed)
    Sd.free()
    # float32 training with a different summation order: only paths whose payoff sits within rounding of the
    # (tiny) continuation value may decide differently
    ex = out["tex"] < N
    assert int((ex != ex_o).sum()) <= max(2, M // 500)
    assert out["price"] == pytest.approx(cf_o.mean(), rel=2e-3)
    assert out["std"] == pytest.approx(cf_o.std(), rel=2e-3)
    assert abs(out["sum_nitm"] - nitm_o.sum()) <= max(4, nitm_o.sum() // 200)


@pytest.mark.parametrize("tag", TAGS)
def test_contnet_flow_prices_the_references_own_paths_like_the_reference(ctx, golden, tag):
    """Same paths as a recorded run of Options_model.price_american_option / OptionPricer, own nets: the
    reference's price moves by ~0.5 % with the (unseeded) torch initialisation; ours must sit in that band."""
    S32, K, r, T, is_put, stats = _case(golden, tag)
    Sd = ctx.to_device(S32)
    prices = [ctx.lsm_contnet(Sd, K, r, T, is_put, 32, 10, 1e-3, s, want_state=False)["price"] for s in range(4)]
    Sd.free()
    assert abs(np.mean(prices) - stats[0]) <= 0.01 * stats[0]
    assert max(abs(p - stats[0]) for p in prices) <= 0.02 * stats[0]


def test_contnet_initialisation_is_torchs_linear_default(ctx):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights AND biases (torch.nn.Linear.reset_parameters), padding zero."""
    h = 32
    ws = [ctx.contnet_init_params(h, t, 7) for t in range(1, 200)]
    w0 = np.concatenate([w["w0"].ravel() for w in ws] + [w["b0"] for w in ws])
    w1 = np.concatenate([w["w1"].ravel() for w in ws] + [w["b1"] for w in ws] +
                        [w["w2"].ravel() for w in ws] + [w["b2"] for w in ws])
    b = 1 / np.sqrt(h)
    assert np.abs(w0).max() <= 1.0 and np.abs(w0).max() > 0.99
    assert np.abs(w1).max() <= b and np.abs(w1).max() > 0.99 * b
    assert abs(w0.mean()) < 0.03 and w0.std() == pytest.approx(1 / np.sqrt(3), rel=0.03)
    assert abs(w1.mean()) < 0.003 and w1.std() == pytest.approx(b / np.sqrt(3), rel=0.02)
    # narrower net in the 32-unit trainer: units >= h are exactly zero, the rest bounded by 1/sqrt(h)
    w = ctx.contnet_init_params(16, 3, 7)
    H = 32
    flat = w["flat"]
    l0 = flat[:8 * H].reshape(H, 8)
    assert not l0[16:].any() and not l0[:, 1:7].any()
    w1p = flat[8 * H:8 * H + H * H].reshape(H, H)
    assert not w1p[16:].any() and not w1p[:, 16:].any()
    assert np.abs(w["w1"]).max() <= 0.25
    # different steps and seeds draw different nets; same (seed, t) the same
    a, b2, c = ctx.contnet_init_params(h, 5, 7), ctx.contnet_init_params(h, 6, 7), ctx.contnet_init_params(h, 5, 8)
    assert not np.array_equal(a["flat"], b2["flat"]) and not np.array_equal(a["flat"], c["flat"])
    assert np.array_equal(a["flat"], ctx.contnet_init_params(h, 5, 7)["flat"])


@pytest.mark.parametrize("hidden,epochs,lr", [(16, 5, 1e-2), (64, 3, 1e-3), (100, 2, 5e-3)])
def test_contnet_honours_hidden_epochs_and_learning_rate(ctx, golden, hidden, epochs, lr):
    """options_model_2.OptionPricer(nn_hidden, nn_epochs, nn_lr): other widths go through the 64- and 128-unit
    trainers with the same zero padding."""
    S32, K, r, T, is_put, _ = _case(golden, "v2_put")
    N, M = S32.shape[0] - 1, S32.shape[1]
    cf_o, ex_o, nitm_o, _ = rf.lsm_per_step_contnet(
        S32.astype(np.float64), K, r, T, is_put, hidden=hidden, epochs=epochs, lr=lr,
        init=lambda t: ctx.contnet_init_params(hidden, t, 99))
    Sd = ctx.to_device(S32)
    out = ctx.lsm_contnet(Sd, K, r, T, is_put, hidden, epochs, lr, 99)
    Sd.free()
    assert int(((out["tex"] < N) != ex_o).sum()) <= max(2, M // 250)
    assert out["price"] == pytest.approx(cf_o.mean(), rel=3e-3)


def test_contnet_is_reproducible_and_epochs_zero_uses_the_initial_net(ctx, golden):
    S32, K, r, T, is_put, _ = _case(golden, "v1_put")
    Sd = ctx.to_device(S32)
    a = ctx.lsm_contnet(Sd, K, r, T, is_put, 32, 10, 1e-3, 5)
    b = ctx.lsm_contnet(Sd, K, r, T, is_put, 32, 10, 1e-3, 5)
    z = ctx.lsm_contnet(Sd, K, r, T, is_put, 32, 0, 1e-3, 5)
    Sd.free()
    assert a["price"] == b["price"] and np.array_equal(a["tex"], b["tex"])
    cf_o, ex_o, _, _ = rf.lsm_per_step_contnet(S32.astype(np.float64), K, r, T, is_put, 32, 0, 1e-3,
                                               init=lambda t: ctx.contnet_init_params(32, t, 5))
    assert int(((z["tex"] < S32.shape[0] - 1) != ex_o).sum()) <= 2
    assert z["price"] == pytest.approx(cf_o.mean(), rel=1e-3)


def test_contnet_fused_pricing_and_degenerate_sets(ctx):
    """omc_price_american_contnet generates the paths itself; deep out-of-the-money contracts have empty or
    one-member sets (std == 0 -> centred only, Options_model.py:124) and must not fault."""
    from options_model_amd import _ffi
    p = _ffi.make_params(is_put=True, n_paths=20000, n_steps=30, seed=11)
    out = ctx.price_american_contnet(p, 32, 10, 1e-3, 11)
    poly = ctx.price_american(p)
    # barely trained nets -> exercise almost whenever in the money: a different (worse) policy than the
    # regression's, same ballpark
    assert out["price"] != poly["price"] and 0.88 * poly["price"] < out["price"] < 1.12 * poly["price"]
    assert out["ms_lsm"] > 0 and out["sum_nitm"] > 0
    far = _ffi.make_params(is_put=True, S0=100.0, K=62.0, n_paths=4096, n_steps=12, seed=3)
    o = ctx.price_american_contnet(far, 32, 10, 1e-3, 0)
    assert np.isfinite(o["price"]) and o["price"] >= 0.0
    with pytest.raises(ValueError):
        ctx.price_american_contnet(_ffi.make_params(n_paths=1000, n_steps=5, semantics="two_pass"), 32, 10, 1e-3, 0)
    with pytest.raises(ValueError):
        ctx.price_american_contnet(p, 500, 10, 1e-3, 0)


def test_curves_with_the_network_run_concurrently_and_equal_the_point_by_point_prices(ctx, monkeypatch):
    """compute_curve_for_S0 (v1) / OptionPricer.compute_curve_for_S0 (v2) in their default (network) mode issue
    the points from several host threads, one context each: same numbers as one pricing after the other."""
    import math
    import time
    from options_model_amd.compat import Options_model as v1
    from options_model_amd.compat.options_model_2 import OptionPricer
    monkeypatch.delenv("OMC_REGRESSOR", raising=False)
    monkeypatch.setenv("OMC_CURVE_STREAMS", "1")
    t0 = time.perf_counter()
    seq = v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 10000, 2, 40, "put", 2, False, 2025)
    t_seq = time.perf_counter() - t0
    monkeypatch.setenv("OMC_CURVE_STREAMS", "8")
    v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 10000, 2, 8, "put", 2, False, 2025)  # contexts created
    t0 = time.perf_counter()
    par = v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 10000, 2, 40, "put", 2, False, 2025)
    t_par = time.perf_counter() - t0
    assert par == seq
    print(f"\n40-point network curve: {t_seq * 1e3:.0f} ms sequential, {t_par * 1e3:.0f} ms on 8 contexts")
    d = par[0]["Days to Expiry"]
    m, s, z = v1.price_american_option(100.0, 100.0, d / 365, 0.05, 0.2, 10000, max(10, min(130, int(math.ceil(d)))),
                                       "put", 2, False, 2025)
    assert (par[0]["Option Value"], par[0]["Std Dev"], par[0]["Zero Prob"]) == (m, s, z)
    p = OptionPricer(100.0, 0.05, None, "put", 2, 7, True, dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7),
                     nn_hidden=16, nn_epochs=5)
    recs = p.compute_curve_for_S0(95.0, 1, 6, 4000, False)
    for r in recs:
        d = r["Days to Expiry"]
        assert r["Option Value"] == p.price_american_option(95.0, d / 365, 4000, max(10, min(130, int(math.ceil(d)))))


@pytest.mark.parametrize("M,N,hidden", [(2, 1, 32), (2, 2, 1), (66, 3, 5), (1002, 7, 32), (4098, 4, 64)])
def test_contnet_edge_geometries_follow_the_restatement(ctx, M, N, hidden):
    """Degenerate shapes of the reference loop: a single step (no regression at all), two steps (one fit), sets
    of a handful of paths, a one-unit net, path counts that break every vector width."""
    rng = np.random.default_rng(M * 131 + N)
    z = rng.standard_normal((N, M // 2))
    S = rf.gbm_paths_from_normals(z, 100.0, 0.05, 0.3, 0.5).astype(np.float32)
    for is_put in (True, False):
        cf_o, ex_o, nitm_o, _ = rf.lsm_per_step_contnet(S.astype(np.float64), 100.0, 0.05, 0.5, is_put, hidden=hidden,
                                                        epochs=3, lr=1e-3,
                                                        init=lambda t: ctx.contnet_init_params(hidden, t, 17))
        Sd = ctx.to_device(S)
        out = ctx.lsm_contnet(Sd, 100.0, 0.05, 0.5, is_put, hidden, 3, 1e-3, 17)
        Sd.free()
        assert int(((out["tex"] < N) != ex_o).sum()) <= max(1, M // 300)
        assert out["sum_nitm"] == pytest.approx(nitm_o.sum(), abs=max(1, M // 300))
        assert out["price"] == pytest.approx(cf_o.mean(), rel=5e-3, abs=1e-9)
