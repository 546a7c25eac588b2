"""GPU parity tests proper: every call goes through the C ABI (libomc.so) and is compared
with the CPU oracle (oracle/) and with fixtures captured from the reference
(tests/golden/, made by tools/capture_golden.py).

Tolerances (float32 paths, float64 sums; DESIGN.md "Numerics"):
  * Philox words                      bit-exact
  * normals                           |dz| <= 4e-6   (hardware log2/sin/cos vs libm)
  * paths, same normals               rel <= 2e-5 over <= 252 multiplicative steps
  * exercise decisions, same paths    identical vectors (ties have measure zero)
  * prices, same paths                rel <= 1e-9
  * prices, end to end                rel <= 1e-3 (north_star)
"""
import numpy as np
import pytest

from oracle import cpu as orc
from oracle import reference_flow as rf

pytestmark = pytest.mark.gpu

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0
HP = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)


def cashflows(sx, tex, N, is_put, tval, r=R, Tm=T, k=K):
    pay = np.maximum(k - sx.astype(np.float64), 0) if is_put else np.maximum(sx.astype(np.float64) - k, 0)
    return pay * np.exp(-r * Tm / N * (tex - tval))


# ---------------------------------------------------------------------------- RNG
def test_philox_kat_on_device(ctx, golden):
    kats = golden["scalars"]["philox4x32_10_kat"]
    inp = np.array([[int(x, 16) for x in k["ctr"] + k["key"]] for k in kats], np.uint32)
    out = ctx.philox4x32_10(inp)
    for o, k in zip(out, kats):
        assert [f"{v:08x}" for v in o] == k["out"]


def test_philox_random_counters_match_oracle(ctx):
    rng = np.random.default_rng(0)
    inp = rng.integers(0, 2**32, size=(4096, 6), dtype=np.uint64).astype(np.uint32)
    out = ctx.philox4x32_10(inp)
    ref = np.stack([orc.philox4x32_10(row[:4], row[4:]) for row in inp[:512]])
    assert np.array_equal(out[:512], ref)


@pytest.mark.parametrize("n_pairs,n_steps,off", [(1000, 10, 0), (257, 7, 123456789012)])
def test_normals_match_oracle(ctx, n_pairs, n_steps, off):
    z = ctx.gbm_normals(n_pairs, n_steps, seed=42, stream=3, pair_offset=off).to_host()
    zo = orc.gbm_normals(n_pairs, n_steps, 42, 3, off)
    assert np.abs(z - zo).max() <= 4e-6
    assert np.isfinite(z).all()


def test_normals_moments(ctx):
    z = ctx.gbm_normals(1 << 20, 8, seed=7).to_host().astype(np.float64)
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs((z**3).mean()) < 5 * np.sqrt(15 / n)
    assert abs((z**4).mean() - 3) < 5 * np.sqrt(96 / n)


# ---------------------------------------------------------------------------- paths
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_gbm_from_reference_normals(ctx, golden, tag):
    g = golden["paths"]
    S0, r, sig, Tm = g["gbm_params"]
    S = ctx.gbm_paths_from_normals(g[f"gbm_{tag}_zhalf"], S0, r, sig, Tm).to_host()
    ref = g[f"gbm_{tag}_S"]  # float64, the reference's own recurrence on its own normals
    assert S.shape == ref.shape
    assert np.abs(S / ref - 1).max() <= 2e-5


@pytest.mark.parametrize("pname", ["feller", "clamp"])
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_heston_from_reference_normals(ctx, golden, pname, tag):
    g = golden["paths"]
    hp = g[f"heston_{pname}_params"]
    S = ctx.heston_paths_from_normals(g[f"heston_{pname}_{tag}_z1"], g[f"heston_{pname}_{tag}_z2"],
                                      *hp, scheme=0).to_host()
    ref = g[f"heston_{pname}_{tag}_S"]
    assert np.abs(S / ref - 1).max() <= 5e-5
    So = orc.heston_paths_from_normals(g[f"heston_{pname}_{tag}_z1"], g[f"heston_{pname}_{tag}_z2"],
                                       *hp, scheme=0)
    assert np.abs(S / So - 1).max() <= 2e-5


def test_heston_full_truncation_differs_only_when_clamped(ctx, golden):
    g = golden["paths"]
    hp = g["heston_clamp_params"]
    z1, z2 = g["heston_clamp_mid_z1"], g["heston_clamp_mid_z2"]
    S1 = ctx.heston_paths_from_normals(z1, z2, *hp, scheme=1).to_host()
    ref = rf.heston_paths_from_normals(z1, z2, *hp, scheme=1)
    # sqrt(v+) is not Lipschitz at 0: with xi=1 the variance sits on the boundary and the
    # float32-vs-float64 gap is amplified there, so the band is wider than for scheme 0
    assert np.abs(S1 / ref - 1).max() <= 1e-3
    # Against the float32 oracle (same operation order; libm sqrtf / exp2 there, v_sqrt_f32 / v_exp_f32
    # here, 1 ulp apart) the same amplification applies to the few paths whose variance crosses zero:
    # a 1e-9 difference in v becomes a 3e-5 difference in sqrt(v+).  So: the bulk agrees as tightly as
    # scheme 0 does, the worst path stays inside the float64 band.
    d = np.abs(S1 / orc.heston_paths_from_normals(z1, z2, *hp, scheme=1) - 1)
    assert np.quantile(d, 0.99) <= 2e-5 and d.max() <= 1e-3
    S0_ = ctx.heston_paths_from_normals(z1, z2, *hp, scheme=0).to_host()
    assert np.abs(S1 / S0_ - 1).max() > 1e-3  # xi=1 violates Feller: the schemes must differ


@pytest.mark.parametrize("M,N,anti", [(4096, 50, 1), (1000, 13, 1), (10, 3, 1), (777, 9, 0), (2, 1, 1)])
def test_gbm_philox_paths_match_oracle(ctx, M, N, anti):
    S = ctx.gbm_paths(M, N, 100.0, R, SIG, T, seed=42, stream=5, antithetic=bool(anti)).to_host()
    So = orc.gbm_paths(M, N, 100.0, R, SIG, T, 42, 5, 0, anti)
    assert np.abs(S / So - 1).max() <= 2e-5
    assert np.all(S[0] == np.float32(100.0))


@pytest.mark.parametrize("vec", [1, 2, 4])
def test_gbm_vector_widths_agree(ctx, vec):
    ctx.set_option("gbm_vec", vec)
    try:
        S = ctx.gbm_paths(2048, 17, 100.0, R, SIG, T, seed=1).to_host()
    finally:
        ctx.set_option("gbm_vec", 0)
    So = orc.gbm_paths(2048, 17, 100.0, R, SIG, T, 1)
    assert np.abs(S / So - 1).max() <= 2e-5


def test_gbm_shard_invariance(ctx):
    """pair_offset carries the GLOBAL pair index: 2 shards == 1 big run, bit for bit."""
    M, N = 4096, 20
    full = ctx.gbm_paths(M, N, 100.0, R, SIG, T, seed=9).to_host()
    P = M // 2
    a = ctx.gbm_paths(M // 2, N, 100.0, R, SIG, T, seed=9, pair_offset=0).to_host()
    b = ctx.gbm_paths(M // 2, N, 100.0, R, SIG, T, seed=9, pair_offset=P // 2).to_host()
    h = P // 2
    assert np.array_equal(a[:, :h], full[:, :h]) and np.array_equal(a[:, h:], full[:, P:P + h])
    assert np.array_equal(b[:, :h], full[:, h:P]) and np.array_equal(b[:, h:], full[:, P + h:])


@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("M,N", [(4096, 50), (10, 3), (1002, 7)])
def test_heston_philox_paths_match_oracle(ctx, M, N, scheme):
    S = ctx.heston_paths(M, N, 100.0, R, T, seed=42, scheme=scheme, **HP).to_host()
    So = orc.heston_paths(M, N, 100.0, R, T, seed=42, scheme=scheme, **HP)
    assert np.abs(S / So - 1).max() <= 5e-5


def test_gbm_martingale_large(ctx):
    """size-independent property at a BASELINE-sized row count: E[S_T] = S0 e^{rT}."""
    M, N = 1_000_000, 252
    S = ctx.gbm_paths(M, N, 100.0, R, SIG, T, seed=11)
    last = np.empty(M, np.float32)
    ctx.lib.omc_memcpy_d2h(ctx.handle, last.ctypes.data, S.ptr + 4 * M * N, 4 * M)
    S.free()
    x = last.astype(np.float64)
    se = x.std() / np.sqrt(M)
    assert abs(x.mean() - 100 * np.exp(R * T)) < 5 * se
    # antithetic partners: log-returns mirror around the drift
    lr = np.log(x / 100.0)
    assert np.abs((lr[: M // 2] + lr[M // 2:]) / 2 - (R - 0.5 * SIG**2) * T).max() < 2e-4


# ---------------------------------------------------------------------------- LSM
SEM = {"ref": "reference", "textbook": "textbook", "twopass": "two_pass"}


@pytest.mark.parametrize("pc", ["put", "call"])
@pytest.mark.parametrize("name", ["ref", "textbook", "twopass"])
@pytest.mark.parametrize("tag", ["small", "mid"])
def test_lsm_poly_against_golden_flows(ctx, golden, tag, name, pc):
    """Golden = independent numpy lstsq restatement of the reference control flows on the
    reference's own seed-42 paths (float64)."""
    g, pf = golden["paths"], golden["poly"]
    Sref = g[f"gbm_{tag}_S"]
    N = Sref.shape[0] - 1
    is_put = pc == "put"
    S = ctx.to_device(Sref.astype(np.float32))
    out = ctx.lsm_poly(S, K, R, T, is_put, SEM[name], want_state=True)
    S.free()
    cf_g = pf[f"poly_{tag}_{pc}_{name}_cf"]
    ex_g = pf[f"poly_{tag}_{pc}_{name}_ex"]
    tval = 0 if name == "textbook" else 1
    cf = cashflows(out["sx"], out["tex"], N, is_put, tval)
    assert np.array_equal(out["nitm"][1:N], pf[f"poly_{tag}_{pc}_{name}_nitm"][1:N])
    if name != "textbook":  # golden `ex` is the sticky mask; textbook keeps no mask
        assert np.array_equal(out["tex"] < N, ex_g)
    assert np.allclose(cf, cf_g, rtol=2e-6, atol=1e-5)  # atol: float32 rounding of S (ulp(100) = 7.6e-6)
    assert abs(out["price"] - cf_g.mean()) <= 2e-6 * cf_g.mean()
    assert out["n_exercised"] == int((out["tex"] < N).sum())
    assert out["n_zero"] == int((cf == 0).sum())


def test_config1_known_answers_on_the_references_seed42_paths(ctx, golden):
    """BASELINE.md section 2's anchors at configs[0]'s OWN size, on the device: the reference's seed-42 normals
    (RNGManager(42) child generator, 50 x 5,000, checksummed against the capture), the paths built from them by
    omc_gbm_paths_from_normals_f32, and the three polynomial flows through the HIP kernels: 6.480186144667078 (per-step
    sticky flow, 13,046 in-the-money regression rows, 89.91 % exercised), 5.983503863373407 (textbook), 7.444611934789268
    (two-pass; 225,057 pass-1 rows = the reference's own R).  2e-5: float32 paths against the float64 anchors."""
    c1 = golden["scalars"]["c1_seed42_gbm_put"]
    z_half = rf.RNGManager(42).get_child_rng().standard_normal((50, 5000))
    assert z_half.sum() == c1["zhalf_sum"] and list(z_half.ravel()[:4]) == c1["zhalf_first4"]
    S = ctx.gbm_paths_from_normals(z_half, 100.0, R, SIG, T)
    Sh = S.to_host()
    assert Sh.shape == (51, 10000) and Sh[-1].astype(np.float64).sum() == pytest.approx(c1["S_T_sum"], rel=1e-6)
    got = {}
    for sem, key in (("reference", "poly_ref_price"), ("textbook", "poly_textbook_price"), ("two_pass", "poly_twopass_price")):
        out = got[sem] = ctx.lsm_poly(S, K, R, T, True, sem, want_state=True)
        assert out["price"] == pytest.approx(c1[key], rel=2e-5), sem
        o = orc.lsm_poly(Sh, K, R, T, True, sem)  # the C oracle on the very same float32 paths: same decisions
        assert out["price"] == pytest.approx(o["price"], rel=1e-9) and out["sum_nitm"] == o["sum_nitm"]
    S.free()
    assert got["reference"]["sum_nitm"] == c1["poly_ref_sum_nitm"] == 13046
    assert got["reference"]["n_exercised"] / 10000 == pytest.approx(c1["poly_ref_exercised_frac"], abs=2e-4)
    assert got["two_pass"]["sum_nitm"] == c1["poly_twopass_sum_nitm_pass1"] == c1["R"] == 225057


@pytest.mark.parametrize("sem", ["reference", "textbook", "two_pass"])
@pytest.mark.parametrize("is_put", [True, False])
@pytest.mark.parametrize("M,N", [(20000, 50), (1001, 7), (10, 4), (6, 1), (4096, 252)])
def test_lsm_poly_matches_oracle_same_paths(ctx, M, N, is_put, sem):
    So = orc.gbm_paths(M, N, 100.0, R, SIG, T, 123, 0, 0, 1 if M % 2 == 0 else 0)
    S = ctx.to_device(So)
    out = ctx.lsm_poly(S, K, R, T, is_put, sem, want_state=True)
    S.free()
    ref = orc.lsm_poly(So, K, R, T, is_put, sem)
    assert np.array_equal(out["nitm"][1:N], ref["nitm"][1:N])
    assert np.array_equal(out["tex"], ref["tex"])
    assert np.array_equal(out["sx"], ref["sx"])
    assert abs(out["price"] - ref["price"]) <= 1e-9 * max(ref["price"], 1e-12)
    assert abs(out["sumsq"] - ref["sumsq"]) <= 1e-9 * max(ref["sumsq"], 1e-12)
    assert (out["n_exercised"], out["n_zero"], out["sum_nitm"]) == (
        ref["n_exercised"], ref["n_zero"], ref["sum_nitm"])
    big = ref["nitm"][: N + 1] > 50
    assert np.allclose(out["betas"][big], ref["betas"][big], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("is_put", [True, False])
@pytest.mark.parametrize("strike", [100.0, 100.1, 99.99999, 1e-3, 12345.678])
def test_lsm_in_the_money_set_at_the_strike(ctx, strike, is_put):
    """Prices sitting exactly on the strike and one float32 ulp either side of it: the in-the-money
    sets (K - S > 0 / S - K > 0 in float64) must be the oracle's, path for path, in every flow."""
    rng = np.random.default_rng(5)
    M, N = 4096, 6
    kf = np.float32(strike)
    near = np.array([kf, np.nextafter(kf, np.float32(0)), np.nextafter(kf, np.float32(np.inf)),
                     np.nextafter(np.nextafter(kf, np.float32(0)), np.float32(0))], dtype=np.float32)
    So = near[rng.integers(0, 4, size=(N + 1, M))]
    far = rng.random((N + 1, M)) < 0.5
    So[far] = (strike * np.exp(0.2 * rng.standard_normal((N + 1, M)))).astype(np.float32)[far]
    S = ctx.to_device(So)
    for sem in ("reference", "textbook", "two_pass"):
        out = ctx.lsm_poly(S, strike, R, T, is_put, sem, want_state=True)
        ref = orc.lsm_poly(So, strike, R, T, is_put, sem)
        assert np.array_equal(out["nitm"][1:N], ref["nitm"][1:N]), sem
        assert np.array_equal(out["tex"], ref["tex"]), sem
        assert abs(out["price"] - ref["price"]) <= 1e-9 * max(ref["price"], 1e-12), sem
    S.free()


def test_lsm_no_itm_paths_falls_back_to_mean(ctx):
    """deep OTM put: no regression set anywhere -> price = mean terminal payoff = 0
    (options_model_3.py:518-519)."""
    So = orc.gbm_paths(512, 10, 100.0, R, 0.01, 0.1, 3)
    S = ctx.to_device(So)
    out = ctx.lsm_poly(S, 50.0, R, 0.1, True, "reference", want_state=True)
    assert out["price"] == 0.0 and out["sum_nitm"] == 0 and out["n_zero"] == 512
    out2 = ctx.lsm_poly(S, 50.0, R, 0.1, True, "two_pass")
    S.free()
    assert out2["price"] == 0.0


def test_lsm_frozen_betas_replay(ctx, golden):
    g, pf = golden["paths"], golden["poly"]
    Sref = g["gbm_mid_S"]
    N = Sref.shape[0] - 1
    b4 = np.zeros((N + 1, 4))
    b4[:, :3] = pf["poly_mid_put_twopass_betas"]
    b4[:, 3] = pf["poly_mid_put_twopass_nitm"]
    S = ctx.to_device(Sref.astype(np.float32))
    out = ctx.lsm_apply_frozen(S, K, R, T, True, b4)
    S.free()
    assert np.array_equal(out["tex"] < N, pf["poly_mid_put_twopass_ex"])
    cf = cashflows(out["sx"], out["tex"], N, True, 1)
    assert np.allclose(cf, pf["poly_mid_put_twopass_cf"], rtol=2e-6, atol=1e-5)


def test_lsm_unaligned_leading_dimension(ctx):
    """ld not a multiple of 4 -> scalar kernels; same answer as the vector path."""
    M, N = 1002, 9
    So = orc.gbm_paths(M, N, 100.0, R, SIG, T, 5)
    pad = np.zeros((N + 1, M + 3), np.float32)
    pad[:, :M] = So
    Sd = ctx.to_device(pad)
    a = ctx.lsm_poly(Sd, K, R, T, True, "reference", want_state=True, n_paths=M)
    Sd.free()
    ref = orc.lsm_poly(So, K, R, T, True, "reference")
    assert np.array_equal(a["tex"], ref["tex"])
    assert abs(a["price"] - ref["price"]) <= 1e-9 * ref["price"]


# ---------------------------------------------------------------------------- fused pricing
@pytest.mark.parametrize("sem", ["reference", "textbook", "two_pass"])
def test_price_american_gbm_matches_oracle(ctx, sem):
    from options_model_amd import _ffi
    M, N = 100_000, 50
    p = _ffi.make_params(model="gbm", is_put=True, semantics=sem, n_paths=M, n_steps=N, seed=42, stream=1)
    out = ctx.price_american(p)
    So = orc.gbm_paths(M, N, 100.0, R, SIG, T, 42, 1)
    ref = orc.lsm_poly(So, K, R, T, True, sem)
    assert abs(out["price"] - ref["price"]) <= 1e-3 * ref["price"]
    assert out["n_paths"] == M and out["ms_paths"] > 0 and out["ms_lsm"] > 0


def test_price_american_heston_call_matches_oracle(ctx):
    from options_model_amd import _ffi
    M, N = 100_000, 50
    p = _ffi.make_params(model="heston", is_put=False, semantics="reference", n_paths=M, n_steps=N,
                         seed=42, **HP)
    out = ctx.price_american(p)
    So = orc.heston_paths(M, N, 100.0, R, T, seed=42, **HP)
    ref = orc.lsm_poly(So, K, R, T, False, "reference")
    assert abs(out["price"] - ref["price"]) <= 1e-3 * ref["price"]


def test_price_american_keeps_paths_and_is_deterministic(ctx):
    from options_model_amd import _ffi
    M, N = 8192, 20
    p = _ffi.make_params(n_paths=M, n_steps=N, seed=3)
    keep = ctx.empty((N + 1, M), np.float32)
    a = ctx.price_american(p, keep)
    b = ctx.price_american(p)
    assert a["price"] == b["price"] and a["sumsq"] == b["sumsq"]  # bitwise reproducible
    S = keep.to_host()
    ref = orc.lsm_poly(S, K, R, T, True, "reference")
    assert abs(a["price"] - ref["price"]) <= 1e-9 * ref["price"]
    keep.free()


def test_textbook_price_close_to_binomial_anchor(ctx):
    """ATM American put S0=K=100, r=5%, sigma=20%, T=1: binomial/PSOR ~ 6.09."""
    from options_model_amd import _ffi
    p = _ffi.make_params(semantics="textbook", n_paths=1_000_000, n_steps=50, seed=2024)
    out = ctx.price_american(p)
    assert 6.02 < out["price"] < 6.12


def test_price_european_matches_black_scholes(ctx, golden):
    from options_model_amd import _ffi
    bs = golden["scalars"]["black_scholes"]
    for is_put, key in ((True, "put_100_100_1_0.05_0.2"), (False, "call_100_100_1_0.05_0.2")):
        p = _ffi.make_params(is_put=is_put, n_paths=2_000_000, n_steps=16, seed=77)
        out = ctx.price_european(p)
        se = np.sqrt((out["sumsq"] / out["n_paths"] - out["price"]**2) / out["n_paths"])
        assert abs(out["price"] - bs[key]) < 5 * se
    # and equals the terminal row of the stored-path generator on the same stream
    p = _ffi.make_params(is_put=True, n_paths=4096, n_steps=10, seed=5)
    out = ctx.price_european(p)
    So = orc.gbm_paths(4096, 10, 100.0, R, SIG, T, 5)
    s, q = orc.european_from_paths(So, K, R, T, True)
    assert abs(out["sum"] - s) <= 1e-4 * s


# ---------------------------------------------------------------------------- errors
def test_invalid_arguments_raise_value_error(ctx):
    from options_model_amd import _ffi
    for kw, msg in ((dict(S0=-1.0), "S0, K, T must be positive."), (dict(r=-0.01), "r must be non-negative."),
                    (dict(n_paths=0), "num_simulations and num_time_steps must be positive integers."),
                    (dict(sigma=0.0), "S0, K, T, and sigma must be positive.")):
        base = dict(n_paths=1000, n_steps=10)
        base.update(kw)
        with pytest.raises(ValueError, match=msg.replace(".", r"\.")):
            ctx.price_american(_ffi.make_params(**base))


# ---------------------------------------------------------------------------- per-step flow vs RUNS of the reference
PER_STEP_TAGS = ["v1_put", "v1_call", "v1_put_odd", "v2_put", "v2_heston_put"]


@pytest.mark.parametrize("tag", PER_STEP_TAGS)
def test_per_step_kernel_replays_a_recorded_run_of_the_reference(ctx, golden, tag):
    """omc_lsm_apply_values = lsm_step_kernel with the continuation values of a REAL run of
    Options_model.price_american_option / options_model_2.OptionPricer (every ContNet output recorded,
    tools/capture_golden_per_step.py) in place of the polynomial: sticky mask, discount order, strict '>',
    valuation at t = dt and (mean, std, zero_prob) are then the reference's, not a restatement's.
    Exact against the oracle on the same float32 paths; against the reference's float64 run, only paths
    within float32 rounding of their continuation value may differ."""
    g = golden["per_step"]
    S64, cont = g[f"{tag}_S"], g[f"{tag}_cont"]
    S0, Kp, Tp, rp, sig, is_put, seed = g[f"{tag}_params"]
    is_put = bool(is_put)
    N, M = S64.shape[0] - 1, S64.shape[1]
    S32 = S64.astype(np.float32)
    cont_inf = np.where(np.isfinite(cont), cont, np.float32(np.inf)).astype(np.float32)
    cf_o, ex_o, _, _ = rf.lsm_per_step(S32.astype(np.float64), Kp, rp, Tp, is_put, cont_values=cont_inf)
    Sd, Cd = ctx.to_device(S32), ctx.to_device(cont_inf)
    out = ctx.lsm_apply_values(Sd, Kp, rp, Tp, is_put, Cd, "reference")
    Sd.free(), Cd.free()
    assert np.array_equal(out["tex"] < N, ex_o)
    cf = cashflows(out["sx"], out["tex"], N, is_put, 1, r=rp, Tm=Tp, k=Kp)
    assert np.allclose(cf, cf_o, rtol=1e-12, atol=0)
    assert out["price"] == pytest.approx(cf_o.mean(), rel=1e-12)
    assert out["std"] == pytest.approx(cf_o.std(), rel=1e-10)
    assert out["zero_prob"] == np.mean(cf_o == 0) and out["n_exercised"] == int(ex_o.sum())
    # and the reference's own numbers (float64 paths): Options_model.py:153-157
    mean_ref, std_ref, zero_ref = g[f"{tag}_stats"]
    assert int(((out["tex"] < N) != g[f"{tag}_ex"]).sum()) <= 2
    assert out["price"] == pytest.approx(mean_ref, rel=2e-4)
    assert out["std"] == pytest.approx(std_ref, rel=2e-4)
    assert abs(out["zero_prob"] - zero_ref) <= 2.0 / M


def test_values_mode_textbook_overwrites_and_discounts_to_zero(ctx, golden):
    """The same entry under semantics 1 against the oracle's textbook loop fed the same values."""
    g = golden["per_step"]
    tag = "v1_put"
    S32 = g[f"{tag}_S"].astype(np.float32)
    S0, Kp, Tp, rp, sig, is_put, seed = g[f"{tag}_params"]
    N = S32.shape[0] - 1
    rng = np.random.default_rng(3)
    # any continuation surface will do for a control-flow check: a noisy fraction of the payoff
    pay = np.maximum(Kp - S32.astype(np.float64), 0)
    cont = (pay * rng.uniform(0.5, 1.5, size=pay.shape)).astype(np.float32)
    cf_o, ex_o, _, _ = rf.lsm_per_step(S32.astype(np.float64), Kp, rp, Tp, True, textbook=True, cont_values=cont)
    Sd, Cd = ctx.to_device(S32), ctx.to_device(cont)
    out = ctx.lsm_apply_values(Sd, Kp, rp, Tp, True, Cd, "textbook")
    Sd.free(), Cd.free()
    cf = cashflows(out["sx"], out["tex"], N, True, 0, r=rp, Tm=Tp, k=Kp)
    assert np.allclose(cf, cf_o, rtol=1e-12, atol=0)
    assert out["price"] == pytest.approx(cf_o.mean(), rel=1e-12)


def test_values_mode_rejects_bad_arguments(ctx):
    S = ctx.to_device(np.full((4, 8), 100.0, np.float32))
    C_ = ctx.to_device(np.zeros((4, 8), np.float32))
    with pytest.raises(ValueError):
        ctx.lsm_apply_values(S, K, R, T, True, C_, "two_pass")
    S.free(), C_.free()


@pytest.mark.parametrize("sem", ["reference", "textbook"])
def test_per_step_sweep_graph_replay_equals_kernel_by_kernel(ctx, sem):
    """The captured HIP graph of the N-launch sweep and the same launches issued one by one are the same
    kernels on the same data: identical results, also when the geometry changes between calls."""
    from options_model_amd import _ffi
    res = {}
    for graph in (1, 0):
        ctx.set_option("step_graph", graph)
        res[graph] = [ctx.price_american(_ffi.make_params(semantics=sem, n_paths=M_, n_steps=N_, seed=7, stream=s))
                      for (M_, N_, s) in ((40000, 30, 1), (40000, 30, 2), (10002, 17, 3), (40000, 30, 1))]
    ctx.set_option("step_graph", -1)
    for a, b in zip(res[1], res[0]):
        assert (a["price"], a["sumsq"], a["n_exercised"], a["n_zero"], a["sum_nitm"]) == (
            b["price"], b["sumsq"], b["n_exercised"], b["n_zero"], b["sum_nitm"])
    assert res[1][0]["price"] == res[1][3]["price"]
