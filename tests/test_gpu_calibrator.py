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
