"""CPU suite, part 1: the oracle (oracle/) against the fixtures captured from the reference
(tests/golden/, tools/capture_golden.py).  No GPU.  This is what pins the checker itself.
"""
import numpy as np
import pytest

from oracle import cpu as orc
from oracle import reference_flow as rf

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0


def test_philox_known_answers(golden):
    for k in golden["scalars"]["philox4x32_10_kat"]:
        o = orc.philox4x32_10([int(x, 16) for x in k["ctr"]], [int(x, 16) for x in k["key"]])
        assert [f"{v:08x}" for v in o] == k["out"]


def test_rng_manager_child_seeds(golden):
    m = rf.RNGManager(42)
    assert [m.get_child_seed() for _ in range(6)] == golden["scalars"]["rng_manager_42_child_seeds"]
    assert golden["scalars"]["c1_seed42_gbm_put"]["child_seed"] == rf.RNGManager(42).get_child_seed()


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_gbm_recurrence_bit_exact(golden, tag):
    g = golden["paths"]
    S = rf.gbm_paths_from_normals(g[f"gbm_{tag}_zhalf"], *g["gbm_params"])
    assert np.array_equal(S, g[f"gbm_{tag}_S"])
    S32 = orc.gbm_paths_from_normals(g[f"gbm_{tag}_zhalf"], *g["gbm_params"])
    assert np.abs(S32 / g[f"gbm_{tag}_S"] - 1).max() < 1e-5


@pytest.mark.parametrize("pname", ["feller", "clamp"])
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_heston_recurrence_bit_exact(golden, pname, tag):
    g = golden["paths"]
    z1, z2 = g[f"heston_{pname}_{tag}_z1"], g[f"heston_{pname}_{tag}_z2"]
    hp = g[f"heston_{pname}_params"]
    assert np.array_equal(rf.heston_paths_from_normals(z1, z2, *hp), g[f"heston_{pname}_{tag}_S"])
    S32 = orc.heston_paths_from_normals(z1, z2, *hp)
    assert np.abs(S32 / g[f"heston_{pname}_{tag}_S"] - 1).max() < 2e-5


def test_features_bit_exact(golden):
    g = golden["paths"]
    Kf, _, Tf = g["feat_KrT"]
    for t, ref in zip(g["feat_tcur"], g["feat_out"]):
        assert np.array_equal(rf.regression_features(g["feat_S"], Kf, Tf, t), ref)


def test_welford_merge(golden):
    g = golden["paths"]
    data, sizes = g["welford_data"], g["welford_sizes"]
    mean, m2, n, o = 0.0, 0.0, 0, 0
    for s, ref in zip(sizes, g["welford_trace"]):
        mean, m2, n = rf.welford_batch_update(mean, m2, n, data[o:o + s])
        o += s
        assert (mean, m2, n) == tuple(ref[:2]) + (int(ref[2]),)
    res = rf.streaming_stats(np.split(data, np.cumsum([500] * 9)))
    assert np.allclose(res, g["welford_streaming_result"], rtol=0, atol=0)
    assert abs(res[0] - data.mean()) < 1e-12 and res[2] == data.size


def test_black_scholes_closed_form(golden):
    bs = golden["scalars"]["black_scholes"]
    assert abs(rf.black_scholes_price(100, 100, 1, 0.05, 0.2, "put") - bs["put_100_100_1_0.05_0.2"]) < 1e-12
    assert abs(rf.black_scholes_price(100, 100, 1, 0.05, 0.2, "call") - bs["call_100_100_1_0.05_0.2"]) < 1e-12
    assert abs(rf.black_scholes_price(90, 100, 0.25, 0.03, 0.35, "put") - bs["put_90_100_0.25_0.03_0.35"]) < 1e-12
    assert abs(rf.black_scholes_price(120, 100, 2, 0.01, 0.15, "call") - bs["call_120_100_2_0.01_0.15"]) < 1e-12


SEM = {"ref": "reference", "textbook": "textbook", "twopass": "two_pass"}


@pytest.mark.parametrize("pc", ["put", "call"])
@pytest.mark.parametrize("name", ["ref", "textbook", "twopass"])
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_c_oracle_flows_against_golden(golden, tag, name, pc):
    g, pf = golden["paths"], golden["poly"]
    Sref = g[f"gbm_{tag}_S"]
    N = Sref.shape[0] - 1
    is_put = pc == "put"
    o = orc.lsm_poly(Sref.astype(np.float32), K, R, T, is_put, SEM[name])
    assert np.array_equal(o["nitm"][1:N], pf[f"poly_{tag}_{pc}_{name}_nitm"][1:N])
    if name != "textbook":
        assert np.array_equal(o["tex"] < N, pf[f"poly_{tag}_{pc}_{name}_ex"])
    cf_g = pf[f"poly_{tag}_{pc}_{name}_cf"]
    pay = np.maximum(K - o["sx"].astype(np.float64), 0) if is_put else np.maximum(o["sx"].astype(np.float64) - K, 0)
    cf = pay * np.exp(-R * T / N * (o["tex"] - (0 if name == "textbook" else 1)))
    assert np.allclose(cf, cf_g, rtol=2e-6, atol=1e-5)  # atol: float32 rounding of S (ulp(100) = 7.6e-6)
    assert abs(o["price"] - cf_g.mean()) <= 2e-6 * cf_g.mean()


@pytest.mark.parametrize("name", ["ref", "textbook", "twopass"])
def test_numpy_flows_reproduce_golden_exactly(golden, name):
    g, pf = golden["paths"], golden["poly"]
    S = g["gbm_mid_S"]
    if name == "twopass":
        reg, pred = rf.two_pass_poly_regressor(K)
        cf, ex, _ = rf.lsm_two_pass(S, K, R, T, True, reg, pred)
    else:
        cf, ex, _, _ = rf.lsm_per_step(S, K, R, T, True, textbook=(name == "textbook"))
    assert np.array_equal(ex, pf[f"poly_mid_put_{name}_ex"])
    assert np.allclose(cf, pf[f"poly_mid_put_{name}_cf"], rtol=1e-12, atol=0)


def test_ols7_flow_reproduces_the_fixture_built_with_the_references_features(golden):
    """`ols7_mid_put_*` (tools/capture_golden.py flow_two_pass_ols7): the v3 control flow with ONE least-squares fit on the
    design matrix the REFERENCE's create_regression_features builds (options_model_3.py:105-121), normalised as
    :550-563.  The oracle's two_pass_ols7_regressor must give its weights, normalisers, decisions and cash-flows."""
    g, pf = golden["paths"], golden["poly"]
    S = g["gbm_mid_S"]
    reg, pred = rf.two_pass_ols7_regressor(K, T, S.shape[0] - 1)
    cf, ex, m = rf.lsm_two_pass(S, K, R, T, True, reg, pred)
    assert m["R"] == int(pf["ols7_mid_put_R"])
    assert np.allclose(m["fm"], pf["ols7_mid_put_feat_mean"], rtol=1e-13, atol=0)
    assert np.allclose(m["fs"], pf["ols7_mid_put_feat_std"], rtol=1e-12, atol=0)
    assert float(m["Y_mean"]) == pytest.approx(float(pf["ols7_mid_put_Y_mean"]), rel=1e-13)
    assert float(m["Y_std"]) == pytest.approx(float(pf["ols7_mid_put_Y_std"]), rel=1e-13)
    assert np.allclose(m["w"], pf["ols7_mid_put_w"], rtol=1e-7, atol=1e-9)  # x, x^2, x^3 are nearly collinear: cond ~ 1e6
    assert np.array_equal(ex, pf["ols7_mid_put_ex"]) and np.allclose(cf, pf["ols7_mid_put_cf"], rtol=1e-12, atol=0)


def _c1_paths(golden):
    c1 = golden["scalars"]["c1_seed42_gbm_put"]
    z_half = rf.RNGManager(42).get_child_rng().standard_normal((50, 5000))
    assert z_half.sum() == c1["zhalf_sum"] and list(z_half.ravel()[:4]) == c1["zhalf_first4"]
    return c1, rf.gbm_paths_from_normals(z_half, 100.0, R, SIG, T)


def test_c1_anchors_seed42(golden):
    """10k x 50 ATM put on the reference's seed-42 paths (SURVEY section 6 / BASELINE.md)."""
    c1, S = _c1_paths(golden)
    assert S[-1].sum() == c1["S_T_sum"]
    cf, ex, _, nitm = rf.lsm_per_step(S, K, R, T, True)
    assert cf.mean() == pytest.approx(c1["poly_ref_price"], rel=1e-12)
    assert int(nitm.sum()) == c1["poly_ref_sum_nitm"] and ex.mean() == c1["poly_ref_exercised_frac"]
    cf, *_ = rf.lsm_per_step(S, K, R, T, True, textbook=True)
    assert cf.mean() == pytest.approx(c1["poly_textbook_price"], rel=1e-12)
    reg, pred = rf.two_pass_ols7_regressor(K, T, 50)
    cf, ex, model = rf.lsm_two_pass(S, K, R, T, True, reg, pred)
    assert model["R"] == c1["R"]
    assert model["Y_mean"] == pytest.approx(c1["Y_mean"], rel=1e-13)
    assert model["Y_std"] == pytest.approx(c1["Y_std"], rel=1e-13)
    assert cf.mean() == pytest.approx(c1["ols7_twopass_price"], rel=1e-9)
    # float32-path C oracle on the same paths: same decisions up to a handful of paths
    S32 = S.astype(np.float32)
    for sem, key in (("reference", "poly_ref_price"), ("textbook", "poly_textbook_price"),
                     ("two_pass", "poly_twopass_price")):
        o = orc.lsm_poly(S32, K, R, T, True, sem)
        assert o["price"] == pytest.approx(c1[key], rel=2e-5)
    assert o["sum_nitm"] == c1["poly_twopass_sum_nitm_pass1"] == c1["R"]


def test_frozen_mlp_pass2_reproduces_reference_decisions(golden):
    """G4: the reference's own trained net (eval mode) replayed by the numpy restatement."""
    nn = golden["nn"]
    for tag, hes in (("gbm_put", False), ("heston_call", True)):
        S0, Kp, r, sig, Tm, is_put, hidden = nn[f"{tag}_params"]
        S = nn[f"{tag}_S"]
        N = S.shape[0] - 1
        state = {k[len(tag) + 4:]: nn[k] for k in nn.files if k.startswith(f"{tag}_sd_")}
        Ym, Ys = nn[f"{tag}_Y_mean_std"]
        reg, pred = rf.two_pass_frozen_mlp_regressor(Kp, Tm, N, state, nn[f"{tag}_feat_mean"],
                                                     nn[f"{tag}_feat_std"], Ym, Ys)
        cf, ex, _ = rf.lsm_two_pass(S, Kp, r, Tm, bool(is_put), reg, pred)
        flips = int((ex != nn[f"{tag}_ex_eval"]).sum())
        assert flips <= 2, flips  # float32 matmul order (numpy vs torch) can flip a boundary path
        assert abs(cf.mean() - float(nn[f"{tag}_price_eval"])) < 2e-3 * float(nn[f"{tag}_price_eval"])
        # normalisers restated from the paths alone
        rows = []
        disc = np.exp(-r * Tm / N)
        c = rf.payoff(S[-1], Kp, bool(is_put)).astype(np.float64)
        for t in range(N - 1, 0, -1):
            c *= disc
            itm = rf.payoff(S[t], Kp, bool(is_put)) > 0
            if itm.any():
                rows.append((t, S[t, itm], c[itm]))
        _, _, fm, fs, ym, ys = rf.normalisers(rows, Kp, Tm, Tm / N)
        assert np.array_equal(fm, nn[f"{tag}_feat_mean"]) and np.array_equal(fs, nn[f"{tag}_feat_std"])
        assert (ym, ys) == (Ym, Ys)


def test_reference_end_to_end_scalars_recorded(golden):
    e2e = golden["scalars"]["end_to_end_10k_x_50_seed42"]
    assert e2e["gbm_put_cv_off"] == 6.812542119814994
    assert e2e["gbm_put_cv_on"] == 6.803039032349685
    assert e2e["heston_call_cv_off"] == 10.346080827233468


def test_philox_paths_statistics():
    S = orc.gbm_paths(200_000, 16, 100.0, R, SIG, T, 99)
    x = S[-1].astype(np.float64)
    assert abs(x.mean() - 100 * np.exp(R * T)) < 5 * x.std() / np.sqrt(x.size)
    s, q = orc.european_from_paths(S, K, R, T, True)
    bs = rf.black_scholes_price(100, 100, 1, R, SIG, "put")
    n = S.shape[1]
    assert abs(s / n - bs) < 5 * np.sqrt((q / n - (s / n) ** 2) / n)
    # shard invariance of the oracle's counter layout
    a = orc.gbm_paths(1000, 8, 100.0, R, SIG, T, 1, 0, 0)
    b = orc.gbm_paths(500, 8, 100.0, R, SIG, T, 1, 0, 250)
    assert np.array_equal(b[:, :250], a[:, 250:500]) and np.array_equal(b[:, 250:], a[:, 750:])


@pytest.mark.parametrize("tag", ["feller", "floor"])
def test_calibrator_heston_scheme_bit_exact(tag):
    """Row f-3: heston_calibration.py:204-312 restated; fixtures from the real HestonPricer."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calibrator.npz"))
    prm = g[f"{tag}_params"]
    S, V = rf.heston_calibrator_paths_from_normals(g[f"{tag}_z1"], g[f"{tag}_z2i"], *prm)
    assert np.array_equal(S, g[f"{tag}_S"]) and np.array_equal(V, g[f"{tag}_V"])
    Sb, _ = rf.heston_calibrator_paths_from_normals(g[f"{tag}_batch_z1"], g[f"{tag}_batch_z2i"], *prm)
    assert np.array_equal(rf.strike_prices(Sb[:, -1], g[f"{tag}_batch_K"], prm[1], prm[2]), g[f"{tag}_batch_prices"])
    assert rf.strike_prices(Sb[:, -1], [100.0], prm[1], prm[2], True)[0] == g[f"{tag}_put100"]
    # float32 C oracle (the GPU's arithmetic contract) on the same normals, [step][pair] layout
    S32 = orc.heston_paths_from_normals(g[f"{tag}_z1"].T.copy(), g[f"{tag}_z2i"].T.copy(), *prm, scheme=2)
    assert np.abs(S32.T / g[f"{tag}_S"] - 1).max() < 5e-6


# ---- per-step control flow pinned to RUNS OF THE REFERENCE's v1 / v2 pricers (tools/capture_golden_per_step.py)
PER_STEP_TAGS = ["v1_put", "v1_call", "v1_put_odd", "v2_put", "v2_heston_put"]


@pytest.mark.parametrize("tag", PER_STEP_TAGS)
def test_per_step_flow_reproduces_reference_run_bit_for_bit(golden, tag):
    """Options_model.price_american_option / options_model_2.OptionPricer were run for real with every
    ContNet output recorded; fed those continuation values, the oracle's per-step loop must return the
    reference's cash-flows, exercise flags and (mean, std, zero_prob) EXACTLY -- sticky mask, discount
    before the in-the-money test, strict '>', valuation at t = dt, population std."""
    g = golden["per_step"]
    S, cont = g[f"{tag}_S"], g[f"{tag}_cont"]
    S0, Kp, Tp, rp, sig, is_put, seed = g[f"{tag}_params"]
    cf, ex, _, nitm = rf.lsm_per_step(S, Kp, rp, Tp, bool(is_put), textbook=False, cont_values=cont)
    assert np.array_equal(cf, g[f"{tag}_cf"])
    assert np.array_equal(ex, g[f"{tag}_ex"])
    assert (cf.mean(), cf.std(), np.mean(cf == 0)) == tuple(g[f"{tag}_stats"])
    # the continuation matrix is populated exactly on the regression sets the flow visits
    assert int(np.isfinite(cont).sum()) == int(nitm.sum())


@pytest.mark.parametrize("tag", PER_STEP_TAGS)
def test_per_step_flow_float32_paths_flip_at_most_a_few_decisions(golden, tag):
    """What the GPU sees is float32(S): state the effect of that rounding on the reference's own run."""
    g = golden["per_step"]
    S32 = g[f"{tag}_S"].astype(np.float32).astype(np.float64)
    S0, Kp, Tp, rp, sig, is_put, seed = g[f"{tag}_params"]
    # a path whose in-the-money status changes under rounding has no recorded value: treat as "hold"
    cont = np.where(np.isfinite(g[f"{tag}_cont"]), g[f"{tag}_cont"], np.float32(np.inf))
    cf, ex, _, _ = rf.lsm_per_step(S32, Kp, rp, Tp, bool(is_put), textbook=False, cont_values=cont)
    assert int((ex != g[f"{tag}_ex"]).sum()) <= 2
    assert cf.mean() == pytest.approx(g[f"{tag}_stats"][0], rel=2e-4)


def test_heston_put_frozen_mlp_pass2_reproduces_reference_decisions(golden):
    """G4 for a Heston PUT (real exercise decisions under stochastic volatility; the Heston call of
    v3_frozen_nn.npz never exercises early)."""
    g = golden["nn_heston_put"]
    tag = "heston_put"
    S0, Kp, rp, sig, Tp, is_put, hidden = g[f"{tag}_params"]
    S = g[f"{tag}_S"]
    N = S.shape[0] - 1
    state = {k[len(f"{tag}_sd_"):]: g[k] for k in g.files if k.startswith(f"{tag}_sd_")}
    ym, ys = g[f"{tag}_Y_mean_std"]
    reg, pred = rf.two_pass_frozen_mlp_regressor(Kp, Tp, N, state, g[f"{tag}_feat_mean"], g[f"{tag}_feat_std"], ym, ys)
    cf, ex, _ = rf.lsm_two_pass(S, Kp, rp, Tp, bool(is_put), reg, pred)
    assert int((ex != g[f"{tag}_ex_eval"]).sum()) <= 2
    assert cf.mean() == pytest.approx(float(g[f"{tag}_price_eval"]), rel=1e-3)
    assert 0.2 < g[f"{tag}_ex_eval"].mean() < 0.99  # decisions really happen


@pytest.mark.parametrize("tag", PER_STEP_TAGS)
def test_contnet_restatement_prices_the_references_paths_like_the_reference(golden, tag):
    """The restatement of the per-step regressor itself (fresh ContNet, standardised inputs, raw targets,
    10 Adam steps): on the reference's recorded paths its price sits within the spread the reference's own
    unseeded initialisation produces (~0.5 %) of the recorded price, and the continuation values it
    produces are as small as the recorded ones (barely trained nets: SURVEY F1)."""
    torch = pytest.importorskip("torch")
    g = golden["per_step"]
    S0, Kp, Tp, rp, sig, is_put, seed = g[f"{tag}_params"]
    prices, scale = [], []
    for s in range(3):
        torch.manual_seed(s)
        cf, ex, nitm, cont = rf.lsm_per_step_contnet(g[f"{tag}_S"], Kp, rp, Tp, bool(is_put))
        prices.append(cf.mean())
        scale.append(np.abs(cont[cont != 0]).mean())
    ref = g[f"{tag}_stats"][0]
    assert abs(np.mean(prices) - ref) <= 0.01 * ref and max(abs(p - ref) for p in prices) <= 0.02 * ref
    rec = g[f"{tag}_cont"]
    rec_scale = np.abs(rec[np.isfinite(rec)]).mean()
    assert 0.3 * rec_scale < np.mean(scale) < 3.0 * rec_scale
