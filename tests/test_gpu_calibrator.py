"""GPU tests of SURVEY row f-3: the European Heston Monte-Carlo inside the reference's calibrator
(heston_calibration.py:197-312), against fixtures captured from the real HestonPricer
(tests/golden/calibrator.npz, tools/capture_golden_calibrator.py) and the CPU oracle."""
import os

import numpy as np
import pytest

from oracle import cpu as orc
from oracle import reference_flow as rf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calibrator.npz")


@pytest.fixture(scope="module")
def cal():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["feller", "floor"])
def test_calibrator_scheme_on_reference_normals(ctx, cal, tag):
    """injected-normals mode, scheme 2: the kernel wants [step][pair] with the INDEPENDENT second
    normal; the reference's arrays are [pair][step] -> transpose.  S compared with the
    reference's own float64 paths."""
    prm = cal[f"{tag}_params"]
    z1, z2 = cal[f"{tag}_z1"].T.copy(), cal[f"{tag}_z2i"].T.copy()
    S = ctx.heston_paths_from_normals(z1, z2, *prm, scheme=2).to_host()
    ref = cal[f"{tag}_S"].T  # [step][path]
    assert S.shape == ref.shape
    assert np.abs(S / ref - 1).max() <= 5e-5
    So = orc.heston_paths_from_normals(z1, z2, *prm, scheme=2)
    assert np.abs(S / So - 1).max() <= 2e-5
    # strike prices from those terminal spots == the reference's simulate-then-average
    Sb = ctx.heston_paths_from_normals(cal[f"{tag}_batch_z1"].T.copy(), cal[f"{tag}_batch_z2i"].T.copy(),
                                       *prm, scheme=2).to_host()
    got = rf.strike_prices(Sb[-1], cal[f"{tag}_batch_K"], prm[1], prm[2])
    assert np.allclose(got, cal[f"{tag}_batch_prices"], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_price_strikes_matches_oracle_same_stream(ctx, scheme):
    K = np.array([70.0, 90.0, 100.0, 105.0, 130.0])
    args = (100.0, 0.03, 0.75, 0.04, 2.0, 0.04, 0.6, -0.7)
    for is_put in (False, True):
        prices, errs = ctx.heston_price_strikes(200_000, 40, *args, K, is_put=is_put, seed=42, stream=3,
                                                scheme=scheme)
        ST = orc.heston_terminal(200_000, 40, *args, seed=42, stream=3, scheme=scheme)
        ref = rf.strike_prices(ST, K, 0.03, 0.75, is_put)
        assert np.allclose(prices, ref, rtol=2e-5, atol=1e-6)
        assert np.all(errs > 0) and np.all(errs < 0.2)


def test_heston_pricer_drop_in(ctx, cal):
    from types import SimpleNamespace

    from options_model_amd.heston_pricer import HestonPricer
    # the caller's own objects (the reference's CalibrationConfig / HestonParams instances in production):
    # the pricer reads attributes, it does not re-declare the calibrator's schema
    cfg = SimpleNamespace(n_mc_paths=400_000, n_time_steps=40, seed=42, verbose=False, plot_results=False)
    prm = SimpleNamespace(kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, v0=0.04)
    pr = HestonPricer(cfg)
    K = cal["feller_batch_K"]
    got = pr.price_options_batch(prm, 100.0, np.concatenate([K, K]), np.r_[np.full(5, 0.75), np.full(5, 0.25)], 0.03)
    assert got.shape == (10,) and np.all(np.diff(got[:5]) < 0) and np.all(got[:5] > got[5:])
    # the reference's 512-path estimate of the same prices: agree within its Monte-Carlo error
    ref = cal["feller_batch_prices"]
    assert np.all(np.abs(got[:5] - ref) < 4 * 12.0 / np.sqrt(512) + 0.05)
    # put-call parity on one simulation ties call and put estimates together
    c = pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03, "call")
    p = pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03, "put")
    assert abs((c - p) - (100.0 - 100.0 * np.exp(-0.03 * 0.75))) < 0.15
    assert np.isnan(pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03, "straddle"))
    # a mapping works too; an object without the Heston fields is refused (as nan, like any pricing failure)
    d = pr.price_european_option(dict(kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, v0=0.04), 100.0, 100.0, 0.75, 0.03)
    assert abs(d - c) < 0.3
    assert np.isnan(pr.price_european_option(SimpleNamespace(kappa=2.0), 100.0, 100.0, 0.75, 0.03))
    # consecutive calls draw fresh normals, like the reference's advancing rng
    assert pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03) != pr.price_european_option(prm, 100.0, 100.0, 0.75, 0.03)


# ---- a whole quote surface in one launch set (omc_heston_price_surface) ---------------------------
def _surface_case(rng, n_exp, max_strikes):
    T = np.sort(rng.uniform(0.05, 2.0, n_exp))
    quotes = [(e, k) for e in range(n_exp) for k in rng.uniform(60.0, 150.0, rng.integers(1, max_strikes + 1))]
    order = rng.permutation(len(quotes))  # quotes arrive in the caller's order, not grouped by expiry
    eo = np.array([quotes[i][0] for i in order], np.int32)
    K = np.array([quotes[i][1] for i in order])
    return T, eo, K


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_surface_equals_the_per_expiry_calls_bit_for_bit(ctx, scheme):
    """One objective evaluation of the calibrator (heston_calibration.py:283-312 loops over the distinct expiries): all
    expiries simulated by ONE launch (expiry on grid.y), all quotes reduced by one more, one wait -- and every quote must
    carry the bits of its own per-expiry call (same Philox sub-stream, same stepper, same summation order), for calls and
    puts, ragged strike counts, quotes in any order, a path count that is not a multiple of the block size."""
    rng = np.random.default_rng(11 + scheme)
    args = dict(S0=100.0, r=0.03, v0=0.05, kappa=1.7, theta=0.045, xi=0.55, rho=-0.6)
    for n_paths, n_steps, n_exp in ((100_000, 100, 6), (30_002, 37, 3), (2, 1, 1)):
        T, eo, K = _surface_case(rng, n_exp, 17)
        streams = 40 + np.arange(n_exp)
        for is_put in (False, True):
            got, err = ctx.heston_price_surface(n_paths, n_steps, args["S0"], args["r"], args["v0"], args["kappa"], args["theta"],
                                                args["xi"], args["rho"], T, streams, K, eo, is_put=is_put, seed=9, scheme=scheme)
            for e in range(n_exp):
                m = eo == e
                one, one_err = ctx.heston_price_strikes(n_paths, n_steps, args["S0"], args["r"], float(T[e]), args["v0"],
                                                        args["kappa"], args["theta"], args["xi"], args["rho"], K[m],
                                                        is_put=is_put, seed=9, stream=int(streams[e]), scheme=scheme)
                assert np.array_equal(got[m], one) and np.array_equal(err[m], one_err), (scheme, n_paths, e, is_put)
    # and against the oracle on the same stream, like the single-expiry entry point
    T, eo, K = _surface_case(rng, 3, 5)
    got, _ = ctx.heston_price_surface(200_000, 40, 100.0, 0.03, 0.04, 2.0, 0.04, 0.6, -0.7, T, [3, 4, 5], K, eo, seed=42, scheme=scheme)
    for e in range(3):
        ST = orc.heston_terminal(200_000, 40, 100.0, 0.03, float(T[e]), 0.04, 2.0, 0.04, 0.6, -0.7, seed=42, stream=3 + e, scheme=scheme)
        assert np.allclose(got[eo == e], rf.strike_prices(ST, K[eo == e], 0.03, float(T[e]), False), rtol=2e-5, atol=1e-6)


def test_surface_argument_errors(ctx):
    from options_model_amd import _ffi
    a = (1000, 10, 100.0, 0.03, 0.04, 2.0, 0.04, 0.3, -0.7)
    with pytest.raises(ValueError, match="expiry_of"):
        ctx.heston_price_surface(*a, [0.5, 1.0], [1, 2], [100.0], [2])
    with pytest.raises(ValueError, match="positive"):
        ctx.heston_price_surface(*a, [0.5, 0.0], [1, 2], [100.0], [0])
    with pytest.raises(ValueError, match="matching"):
        ctx.heston_price_surface(*a, [0.5, 1.0], [1], [100.0], [0])
    with pytest.raises((ValueError, _ffi.OmcError)):
        ctx.heston_price_surface(*a, [], [], [100.0], [0])


def test_heston_pricer_batch_is_the_surface_call_with_the_loops_bits(ctx):
    """HestonPricer.price_options_batch now prices the whole surface in one launch set; the reference's loop shape stays
    available (price_options_batch_per_expiry).  Same sub-streams in the same order -> the same bits, and the pricer's
    stream counter -- the stand-in for the reference's advancing generator -- moves by one per distinct expiry either way."""
    from types import SimpleNamespace

    from options_model_amd.heston_pricer import HestonPricer
    cfg = SimpleNamespace(n_mc_paths=100_000, n_time_steps=100, seed=42)
    prm = SimpleNamespace(kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, v0=0.04)
    rng = np.random.default_rng(5)
    T = rng.choice([30 / 365, 60 / 365, 91 / 365, 0.5, 1.0, 2.0], 60)
    K = rng.uniform(80.0, 125.0, 60)
    a, b = HestonPricer(cfg), HestonPricer(cfg)
    for _ in range(2):  # consecutive evaluations keep drawing fresh sub-streams, in step
        pa, pb = a.price_options_batch(prm, 100.0, K, T, 0.03), b.price_options_batch_per_expiry(prm, 100.0, K, T, 0.03)
        assert np.array_equal(pa, pb) and np.all(pa > 0) and a._stream == b._stream
    assert a._stream == 12
    assert not np.array_equal(pa, HestonPricer(cfg).price_options_batch(prm, 100.0, K, T, 0.03))  # (streams 1-6, not 7-12)
    # a failure is the reference's behaviour: a warning on stdout and nan, never an exception (:308-310)
    bad = a.price_options_batch(SimpleNamespace(kappa=2.0), 100.0, K[:3], T[:3], 0.03)
    assert np.isnan(bad).all() and a.price_options_batch(prm, 100.0, [], [], 0.03).shape == (0,)


def test_surface_with_more_quotes_than_a_grid_dimension(ctx):
    """70,000 quotes over two expiries (the quote sits on grid.x of the reduction: any number; the chunk on grid.y): a
    sample of them against their single-expiry calls, bit for bit."""
    rng = np.random.default_rng(2)
    T = np.array([0.3, 1.1])
    eo = rng.integers(0, 2, 70_000).astype(np.int32)
    K = rng.uniform(70.0, 140.0, 70_000)
    a = (100.0, 0.02, 0.04, 1.5, 0.05, 0.4, -0.5)
    got, err = ctx.heston_price_surface(6000, 20, *a, T, [7, 8], K, eo, seed=5, scheme=2)
    assert got.shape == (70_000,) and np.all(np.isfinite(got)) and np.all(err >= 0)
    for e in (0, 1):
        idx = np.flatnonzero(eo == e)[::997]
        one, one_err = ctx.heston_price_strikes(6000, 20, a[0], a[1], float(T[e]), *a[2:], K[idx], seed=5, stream=7 + e, scheme=2)
        assert np.array_equal(got[idx], one) and np.array_equal(err[idx], one_err)
