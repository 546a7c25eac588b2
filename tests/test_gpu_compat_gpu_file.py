"""options_model_amd.compat.option_model_3_gpu: the entry points of the reference's (non-importable)
GPU file, option_model_3_gpu.py:117-248, 547-956, on the HIP hot path."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
S0, R, SIG, T = 100.0, 0.05, 0.2, 1.0


@pytest.fixture(scope="module")
def gpu():
    from options_model_amd.compat import option_model_3_gpu as g
    assert g.get_device().type == "cuda"
    return g


def test_bs_simulator_shapes_antithetic_layout_and_moments(gpu):
    dev = gpu.get_device()
    torch.manual_seed(3)
    S = gpu.simulate_bs_paths_torch(S0, R, T, SIG, 20001, 40, dev)          # odd: one extra plain path (:140-146)
    assert S.shape == (41, 20001) and S.dtype == torch.float32 and S.device.type == "cuda"
    h = S.cpu().numpy().astype(np.float64)
    assert np.all(h[0] == S0) and np.all(h > 0)
    lr = np.diff(np.log(h), axis=0)
    drift = (R - 0.5 * SIG ** 2) * (T / 40)
    a, b = lr[:, :10000] - drift, lr[:, 10000:20000] - drift                 # partner of j is j + M/2 (:133-134)
    assert np.abs(a + b).max() < 2e-5
    z = a / (SIG * math.sqrt(T / 40))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(h[-1].mean() - S0 * math.exp(R * T)) < 0.25
    torch.manual_seed(3)
    S2 = gpu.simulate_bs_paths_torch(S0, R, T, SIG, 20001, 40, dev)          # torch.manual_seed reproduces it
    assert torch.equal(S, S2)
    S3 = gpu.simulate_bs_paths_torch(S0, R, T, SIG, 20001, 40, dev)          # ...and the generator moves on
    assert not torch.equal(S, S3)


def test_bandwidth_optimized_simulator_is_not_antithetic(gpu):
    torch.manual_seed(5)
    S = gpu.simulate_bs_paths_torch_bandwidth_optimized(S0, R, T, SIG, 30000, 25, gpu.get_device())
    assert S.shape == (26, 30000)
    lr = np.diff(np.log(S.cpu().numpy().astype(np.float64)), axis=0) - (R - 0.5 * SIG ** 2) * (T / 25)
    c = np.corrcoef(lr[:, :15000].ravel(), lr[:, 15000:].ravel())[0, 1]
    assert abs(c) < 0.01                                                      # independent halves (:150-185)
    assert abs(lr.std() / (SIG * math.sqrt(T / 25)) - 1) < 0.01


def test_heston_simulator_matches_the_reference_scheme_in_distribution(gpu):
    hp = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)
    torch.manual_seed(9)
    S = gpu.simulate_heston_paths_torch(S0, R, T, hp["v0"], hp["kappa"], hp["theta"], hp["xi"], hp["rho"], 40001, 50,
                                        gpu.get_device())
    assert S.shape == (51, 40001)
    h = S.cpu().numpy().astype(np.float64)
    assert np.all(h[0] == S0) and np.all(np.isfinite(h)) and np.all(h > 0)
    assert abs(h[-1].mean() - S0 * math.exp(R * T)) < 0.4
    # negative spot / variance correlation shows as a left-skewed log-return
    x = np.log(h[-1] / S0)
    assert ((x - x.mean()) ** 3).mean() / x.std() ** 3 < -0.3
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gpu.simulate_bs_paths_torch(S0, R, T, SIG, 100, 5, torch.device("cpu"))


def test_pricer_surface_of_the_gpu_file(gpu):
    from options_model_amd.pricer import AdvancedOptionPricer as Base, RNGManager
    p = gpu.AdvancedOptionPricer(100.0, R, SIG, "put", gpu.RNGManager(42), False, None, 128, 25, 1e-3, 3, 0.10, False,
                                 regressor="poly")
    b = Base(100.0, R, SIG, "put", RNGManager(42), regressor="poly")
    # same flow, same seed sequence as the v3 surface
    assert p.price_american_enhanced_lsm_gpu(S0, T, 20000, 50) == b.price_american_enhanced_lsm(S0, T, 20000, 50)
    assert p.price_european_gpu(S0, T, 20000, 50) == b.price_european_streaming(S0, T, 20000, 50)
    # the file's own rules: 50,000-path cap (:675) and 2 steps per day under 10 days (:664-667)
    p.price_american_enhanced_lsm_gpu(S0, T, 80000, 50)
    assert p.last_result["n_paths"] == 50000
    p.price_american_enhanced_lsm_gpu(S0, 4 / 365, 4000, 50)
    b2 = Base(100.0, R, SIG, "put", RNGManager(1), regressor="poly")
    p2 = gpu.AdvancedOptionPricer(100.0, R, SIG, "put", gpu.RNGManager(1), regressor="poly")
    assert p2.price_american_enhanced_lsm_gpu(S0, 4 / 365, 4000, 50) == b2.price_american_enhanced_lsm(S0, 4 / 365, 4000, 10)
    cv = p.price_american_option(S0, T, 20000, 50)
    assert 5.5 < cv < 7.5
    eu = gpu.AdvancedOptionPricer(100.0, R, SIG, "put", gpu.RNGManager(42), european_approximation=True,
                                  use_streaming=False, regressor="poly")
    bs = gpu.BlackScholesGreeks.black_scholes_price(S0, 100.0, T, R, SIG, "put")
    assert abs(eu.price_american_option(S0, T, 200000, 20) - bs) < 0.06
    with pytest.raises(ValueError, match="S0, K, T must be positive"):
        p.price_american_enhanced_lsm_gpu(-1.0, T)


def test_batch_entry_points_of_the_gpu_file(gpu, monkeypatch):
    monkeypatch.setenv("OMC_REGRESSOR", "poly")
    recs = gpu.compute_multiple_S0_gpu_batch([95.0, 100.0, 105.0], 100.0, R, SIG, "put", 1, 5, 4000, seed=7)
    assert len(recs) == 15 and [r["S0"] for r in recs[:5]] == [95.0] * 5
    assert [r["Days to Expiry"] for r in recs[:5]] == [5.0, 4.0, 3.0, 2.0, 1.0]
    assert all(np.isfinite(r["Option Value"]) for r in recs)
    again = gpu.compute_multiple_S0_gpu_batch([95.0, 100.0, 105.0], 100.0, R, SIG, "put", 1, 5, 4000, seed=7)
    assert again == recs
    w = gpu.compute_curve_worker_gpu(100.0, 100.0, R, SIG, "put", 7, 1, 3, 2000, False, False, None)
    assert len(w) == 3 and set(w[0]) == {"S0", "Days to Expiry", "Option Value"}
    assert gpu.compute_curve_worker_gpu(100.0, -5.0, R, SIG, "put", 7, 1, 3, 2000, False, False, None) == []
