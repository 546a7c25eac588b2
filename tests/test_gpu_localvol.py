"""GPU tests of SURVEY row f-4: local-vol paths through the IV network, against a fixture captured
from the reference's simulate_local_vol_paths_antithetic (tests/golden/localvol.npz,
tools/capture_golden_localvol.py: its network weights, the normals it drew, the paths it built)."""
import math
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "localvol.npz")


@pytest.fixture(scope="module")
def lv(ctx):
    import torch

    from options_model_amd import local_vol
    g = np.load(GOLD)
    H, L = (int(v) for v in g["arch"])
    S0, r, T, K, m_scale, tau_scale, eps = g["params"]
    net = local_vol.make_iv_network(H, L, eps)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})
    net.scaler = types.SimpleNamespace(m_scale=m_scale, tau_scale=tau_scale)
    return g, local_vol, local_vol.IVModel(net)


def test_iv_network_matches_reference_surface(lv):
    g, local_vol, model = lv
    K = g["params"][3]
    sig = model.get_volatility_batch(K, g["vol_S"], 0.3)
    assert np.allclose(sig, g["vol_out"], rtol=2e-4, atol=1e-6)
    with pytest.raises(ValueError, match="positive"):
        model.get_volatility_batch(K, np.array([100.0, -1.0]), 0.3)
    with pytest.raises(ValueError, match="scaler"):
        local_vol.IVModel(local_vol.make_iv_network())


def test_local_vol_paths_on_reference_normals(lv):
    g, local_vol, model = lv
    S0, r, T, K = g["params"][:4]
    ref = g["S"]
    N, M = ref.shape[0] - 1, ref.shape[1]
    S = local_vol.simulate_local_vol_paths(S0, r, T, M, N, model, K, seed=0, z_half=g["z_half"]).cpu().numpy()
    assert S.shape == ref.shape
    assert np.abs(S / ref - 1).max() <= 3e-4  # float32 path + GPU GEMM order vs float64/CPU-torch
    eu = math.exp(-r * T) * np.maximum(K - S[-1].astype(np.float64), 0).mean()
    assert eu == pytest.approx(float(g["european_put"]), rel=2e-4)


def test_local_vol_philox_paths_and_pricing(lv, ctx):
    g, local_vol, model = lv
    S0, r, T, K = g["params"][:4]
    M, N = 200_000, 24
    S = local_vol.simulate_local_vol_paths(S0, r, T, M, N, model, K, seed=77)
    x = S[-1].double().cpu().numpy()
    assert abs(x.mean() - S0 * math.exp(r * T)) < 5 * x.std() / math.sqrt(M) + 0.02  # martingale (Euler bias)
    a = S[:, : M // 2].double()
    # the library's Philox normals drove it: same seed -> same matrix
    S2 = local_vol.simulate_local_vol_paths(S0, r, T, M, N, model, K, seed=77)
    assert bool((S == S2).all()) and a.shape[1] == M // 2
    from options_model_amd import AdvancedOptionPricer, RNGManager
    p = AdvancedOptionPricer(K, r, None, "put", RNGManager(3), iv_model=model, use_control_variate=False)
    am = p.price_american_option(S0, T, 100_000, N)
    p2 = AdvancedOptionPricer(K, r, None, "put", RNGManager(3), iv_model=model, european_approximation=True)
    eu = p2.price_american_option(S0, T, 100_000, N)
    assert am > eu > 0 and am < K
    assert p.last_result["n_paths"] == 100_000


def test_hip_kernel_matches_torch_backend(lv):
    """The library's one-kernel simulator (network inside the path loop, float32 MFMA) against the
    per-step PyTorch-ROCm evaluation of the same network on the same normals."""
    g, local_vol, model = lv
    S0, r, T, K = g["params"][:4]
    for M, N in ((2, 3), (62, 5), (20_000, 24)):
        a = local_vol.simulate_local_vol_paths(S0, r, T, M, N, model, K, seed=5, backend="hip")
        b = local_vol.simulate_local_vol_paths(S0, r, T, M, N, model, K, seed=5, backend="torch")
        assert a.shape == b.shape == (N + 1, M)
        rel = (a.double() / b.double() - 1).abs().max()
        assert float(rel) <= 2e-5, (M, N, float(rel))  # float32 paths, different summation orders
    with pytest.raises(ValueError, match="backend"):
        local_vol.simulate_local_vol_paths(S0, r, T, 10, 3, model, K, seed=5, backend="cuda")


def test_hip_kernel_rejects_other_widths(lv, ctx):
    g, local_vol, model = lv
    import torch
    assert ctx.lib.omc_localvol_param_count(64, 4) == 64 * 4 + 4 * (64 * 64 + 3 * 64) + 65
    assert ctx.lib.omc_localvol_param_count(128, 4) == -1
    wide = local_vol.make_iv_network(128, 2)
    wide.scaler = types.SimpleNamespace(m_scale=1.0, tau_scale=1.0)
    wm = local_vol.IVModel(wide)
    with pytest.raises(ValueError, match="hidden_dim 64"):
        local_vol.simulate_local_vol_paths(100.0, 0.05, 1.0, 64, 4, wm, 100.0, seed=1, backend="hip")
    S = local_vol.simulate_local_vol_paths(100.0, 0.05, 1.0, 64, 4, wm, 100.0, seed=1)  # auto -> torch
    assert S.shape == (5, 64) and bool(torch.isfinite(S).all())
