"""GPU tests of the batched path (the curve workload = thousands of
independent small pricings).  The batched launches reuse the single-problem kernel bodies, so
a batch must reproduce n separate calls bit for bit."""
import math
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)
KEYS = ("price", "sum", "sumsq", "n_paths", "n_exercised", "n_zero", "sum_nitm")


def _problems(_ffi, sem, model, n=23, uneven=False):
    rng = np.random.default_rng(7)
    ps = []
    for i in range(n):
        M = int(rng.choice([2000, 4096, 10000, 20000])) + (2 if uneven and i % 5 == 0 else 0)
        N = int(rng.integers(1, 61)) if i else 1
        kw = dict(model=model, semantics=sem, is_put=bool(i % 3), n_paths=M, n_steps=N,
                  S0=float(rng.uniform(80, 120)), K=100.0, r=0.05, sigma=float(rng.uniform(0.1, 0.4)),
                  T=float(rng.uniform(0.02, 1.5)), seed=int(rng.integers(1, 2**31)), stream=i % 3)
        if model == "heston":
            kw.update(HP)
        ps.append(_ffi.make_params(**kw))
    return ps


@pytest.mark.parametrize("model", ["gbm", "heston"])
@pytest.mark.parametrize("sem", ["two_pass", "reference", "textbook"])
def test_american_batch_equals_sequential_bitwise(ctx, sem, model):
    from options_model_amd import _ffi
    ps = _problems(_ffi, sem, model)
    seq = [ctx.price_american(p) for p in ps]
    bat = ctx.price_american_batch(ps)
    assert len(bat) == len(ps)
    for a, b in zip(seq, bat):
        assert all(a[k] == b[k] for k in KEYS), (a, b)
    assert bat[0]["ms_total"] > 0


def test_american_batch_unaligned_members_use_scalar_kernels(ctx):
    from options_model_amd import _ffi
    ps = _problems(_ffi, "reference", "gbm", n=11, uneven=True)  # some M % 4 == 2
    seq = [ctx.price_american(p) for p in ps]
    bat = ctx.price_american_batch(ps)
    for a, b in zip(seq, bat):
        assert a["n_exercised"] == b["n_exercised"] and a["n_zero"] == b["n_zero"]
        assert a["price"] == pytest.approx(b["price"], rel=1e-12)  # vec4 vs scalar block geometry


@pytest.mark.parametrize("model", ["gbm", "heston"])
def test_european_batch_equals_sequential(ctx, model):
    from options_model_amd import _ffi
    ps = _problems(_ffi, "two_pass", model, n=17)
    seq = [ctx.price_european(p) for p in ps]
    bat = ctx.price_european_batch(ps)
    for a, b in zip(seq, bat):
        assert a["price"] == b["price"] and a["sumsq"] == b["sumsq"] and a["n_zero"] == b["n_zero"]


def test_batch_rejects_mixed_flows_and_bad_members(ctx):
    from options_model_amd import _ffi
    a = _ffi.make_params(semantics="two_pass", n_paths=1000, n_steps=5)
    b = _ffi.make_params(semantics="reference", n_paths=1000, n_steps=5)
    with pytest.raises(ValueError, match="share model"):
        ctx.price_american_batch([a, b])
    with pytest.raises(ValueError, match="S0, K, T must be positive"):
        ctx.price_american_batch([a, _ffi.make_params(semantics="two_pass", n_paths=1000, n_steps=5, T=-1.0)])
    assert ctx.price_american_batch([]) == []


def test_v3_curve_batched_equals_point_by_point(ctx):
    """AdvancedOptionPricer.compute_curve_for_S0 == the reference's loop of price_american_option
    calls, including the master-seed consumption and the default control-variate wrapper."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    for cv in (False, True):
        a = AdvancedOptionPricer(100, 0.05, 0.2, "put", RNGManager(2025), use_control_variate=cv, regressor="poly")
        b = AdvancedOptionPricer(100, 0.05, 0.2, "put", RNGManager(2025), use_control_variate=cv, regressor="poly")
        curve = a.compute_curve_for_S0(95.0, 2, 30, 4000, False)
        ref = []
        for i in range(30, 0, -1):
            d = i / 2
            ref.append(b.price_american_option(95.0, d / 365, 4000, max(10, min(130, int(math.ceil(d))))))
        assert [r["Days to Expiry"] for r in curve] == [i / 2 for i in range(30, 0, -1)]
        assert [r["Option Value"] for r in curve] == ref
        assert a.rng_manager.get_child_seed() == b.rng_manager.get_child_seed()


def test_v1_v2_curves_batched_equal_point_by_point(ctx):
    from options_model_amd.compat import Options_model as v1
    from options_model_amd.compat.options_model_2 import OptionPricer
    recs = v1.compute_curve_for_S0(100.0, 100.0, 0.05, 0.2, 6000, 1, 12, "put", 2, False, 2025)
    for r in recs:
        d = r["Days to Expiry"]
        m, s, z = v1.price_american_option(100.0, 100.0, d / 365, 0.05, 0.2, 6000,
                                           max(10, min(130, int(math.ceil(d)))), "put", 2, False, 2025)
        assert (r["Option Value"], r["Std Dev"], r["Zero Prob"]) == (m, s, z)
    p = OptionPricer(100.0, 0.05, 0.2, "call", 2, 7, True, HP)
    recs = p.compute_curve_for_S0(105.0, 2, 9, 4000, False)
    for r in recs:
        d = r["Days to Expiry"]
        assert r["Option Value"] == p.price_american_option(105.0, d / 365, 4000,
                                                            max(10, min(130, int(math.ceil(d)))))


def test_ui_sized_batch_runs_in_a_blink(ctx):
    """The reference UI's default job (options_model_2_ui.py:39-46,65-71): S0 80..120 step 5,
    90 days x 2 per day = 9 x 180 = 1620 pricings of 10k paths, per-step flow."""
    from options_model_amd import _ffi
    ps = []
    for S0 in range(80, 121, 5):
        for i in range(180, 0, -1):
            d = i / 2
            ps.append(_ffi.make_params(semantics="reference", is_put=True, n_paths=10000,
                                       n_steps=max(10, min(130, int(math.ceil(d)))), S0=float(S0),
                                       T=d / 365, seed=2025))
    ctx.price_american_batch(ps[:50])  # warm-up
    t0 = time.perf_counter()
    out = ctx.price_american_batch(ps)
    dt = time.perf_counter() - t0
    # deep out-of-the-money short-dated puts (S0 = 120, a few days) are legitimately worth 0
    assert len(out) == 1620 and all(o["price"] >= 0 for o in out)
    assert sum(o["price"] > 0 for o in out) > 1200
    work = sum(p.n_paths * p.n_steps for p in ps)
    print(f"\n1620 pricings (10k paths, 10..130 steps): {dt * 1e3:.1f} ms = {work / dt:.3e} path-steps/s")
    assert dt < 1.0
    # spot check against single calls
    for k in (0, 777, 1619):
        assert out[k]["price"] == ctx.price_american(ps[k])["price"]
