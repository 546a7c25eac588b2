"""Curves with the reference's DEFAULT regressor (one SingleLSMNet per curve point, options_model_3.py:697-713 ->
:565-613) as the UIs run them: the nets of all points are trained side by side (omc_mlp_train_epoch_batch), and every
point still returns the bits of its own price_american_option call."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("hidden,layers,batch", [(128, 3, 256), (64, 2, 256), (64, 3, 512), (128, 2, 1024), (32, 2, 96)])
def test_batched_epoch_equals_single_epochs_bitwise(torch_cuda, hidden, layers, batch):
    """n networks, each with its own rows / learning rate / keys / step counter: one batched epoch (+ a second one, so
    that step counters and Adam state carry over) leaves every network with the bits of its own single-network calls."""
    torch = torch_cuda
    from options_model_amd import _ffi
    from options_model_amd import nn_regressor as nnr
    ctx = nnr._ctx_on_torch_stream(0)
    lib = _ffi.load_library()
    assert lib.omc_mlp_train_batch_supported(hidden, layers, batch)
    npar = lib.omc_mlp_param_count(hidden, layers) if hidden != 32 else 8 * 32 + 32 * 32 + 2 * 32 + 1
    g = torch.Generator(device="cuda").manual_seed(5)
    probs = []
    for i, R in enumerate((3000, 1111, 257, 4096, batch)):
        data = torch.randn((R, 8), generator=g, device="cuda", dtype=torch.float32)
        params = 0.1 * torch.randn(npar, generator=g, device="cuda", dtype=torch.float32)
        probs.append(dict(data=data, R=R, p0=params, lr=1e-3 * (i + 1), seed=1000 + i, key=77 + i))
    drop = 0.1
    single = []
    for pr in probs:
        p, m, v = pr["p0"].clone(), torch.zeros(npar, device="cuda"), torch.zeros(npar, device="cuda")
        step, losses = 0, []
        for ep in range(2):
            loss, step = ctx.mlp_train_epoch(pr["data"].data_ptr(), pr["R"], min(batch, pr["R"]), p.data_ptr(), m.data_ptr(),
                                             v.data_ptr(), step, pr["lr"] * (0.5 ** ep), drop, pr["seed"], hidden=hidden,
                                             layers=layers, shuffle_key=pr["key"] + ep)
            losses.append(loss)
        single.append((p, m, v, step, losses))
    state = [(pr["p0"].clone(), torch.zeros(npar, device="cuda"), torch.zeros(npar, device="cuda")) for pr in probs]
    steps = [0] * len(probs)
    blosses = [[] for _ in probs]
    for ep in range(2):
        outs = ctx.mlp_train_epoch_batch(
            [dict(data_ptr=pr["data"].data_ptr(), n_rows=pr["R"], batch=min(batch, pr["R"]), params_ptr=st[0].data_ptr(),
                  m_ptr=st[1].data_ptr(), v_ptr=st[2].data_ptr(), step=steps[i], lr=pr["lr"] * (0.5 ** ep), seed=pr["seed"],
                  shuffle_key=pr["key"] + ep) for i, (pr, st) in enumerate(zip(probs, state))], hidden, layers, drop)
        for i, (loss, step) in enumerate(outs):
            steps[i] = step
            blosses[i].append(loss)
    for i, ((p, m, v, step, losses), st) in enumerate(zip(single, state)):
        assert steps[i] == step and blosses[i] == losses, i
        assert torch.equal(st[0], p) and torch.equal(st[1], m) and torch.equal(st[2], v), i


def _curve(monkeypatch, batched, seed=11, **kw):
    from options_model_amd import AdvancedOptionPricer, RNGManager
    monkeypatch.setenv("OMC_NN_CURVE_BATCH", "1" if batched else "0")
    p = AdvancedOptionPricer(K=100.0, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(seed), **kw)
    recs = p.compute_curve_for_S0(100.0, 1, 7, 4000, False)   # 7 .. 1 days, steps = 10 each
    return recs, p


@pytest.mark.parametrize("kw", [dict(), dict(nn_hidden=64, nn_epochs=6, use_control_variate=False),
                                dict(nn_hidden=32, nn_epochs=5, use_control_variate=False),
                                dict(nn_hidden=64, nn_layers=2, nn_epochs=4, use_heston=True,
                                     heston_params=dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7))])
def test_nn_curve_batched_equals_point_by_point(torch_cuda, monkeypatch, kw):
    a, pa = _curve(monkeypatch, True, **kw)
    b, pb = _curve(monkeypatch, False, **kw)
    assert [r["Days to Expiry"] for r in a] == [7.0, 6.0, 5.0, 4.0, 3.0, 2.0, 1.0]
    assert a == b                                                        # every point, bit for bit
    assert pa.last_result["trainer"] == "hip" and pa.last_result.get("batched_with") == 7
    assert pa.rng_manager.get_child_seed() == pb.rng_manager.get_child_seed()   # same number of master draws


def test_forty_point_nn_curve_is_at_least_eight_times_faster_than_forty_calls(torch_cuda, monkeypatch):
    """a 40-point curve at 10k paths, default 3 x 128 net, in <= 1/8 of 40 single calls."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    monkeypatch.setenv("OMC_NN_CURVE_BATCH", "1")
    mk = lambda: AdvancedOptionPricer(K=100.0, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(3),  # noqa: E731
                                      use_control_variate=False)
    mk().compute_curve_for_S0(100.0, 1, 2, 10_000, False)    # warm
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = mk().compute_curve_for_S0(100.0, 1, 40, 10_000, False)
    torch.cuda.synchronize()
    t_batch = time.perf_counter() - t0
    # the same points one by one: time the 4 longest (40, 39, 38, 37 days: steps 40 .. 37) and the 4 shortest
    p = mk()
    monkeypatch.setenv("OMC_NN_CURVE_BATCH", "0")
    t0 = time.perf_counter()
    first = p.compute_curve_for_S0(100.0, 1, 40, 10_000, False) if os.environ.get("OMC_TEST_FULL_SEQ") else None
    t_seq = time.perf_counter() - t0
    if first is None:
        q = mk()
        t0 = time.perf_counter()
        part = [q.price_american_option(100.0, d / 365.0, 10_000, max(10, min(130, d))) for d in (40, 39, 38, 37)]
        t4 = time.perf_counter() - t0
        assert part == [r["Option Value"] for r in recs[:4]]          # the first points of the curve, bit for bit
        t_seq = 10.0 * t4                                                # 40 calls (later points have fewer rows: a slight overestimate)
    else:
        assert first == recs
    print(f"40-point NN curve: batched {t_batch:.2f} s, point by point ~{t_seq:.1f} s, ratio {t_seq / t_batch:.1f}")
    assert t_seq / t_batch >= 8.0
