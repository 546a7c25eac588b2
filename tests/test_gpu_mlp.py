"""GPU tests of the fused continuation-value-network trainer (omc_mlp_train_epoch):
the hand-written MFMA forward/backward + Adam kernels against PyTorch autograd + torch.optim.Adam on
the same float32 data and initial weights (the reference's own trainer, options_model_3.py:565-600).
Tolerances are float32 accumulation-order tolerances; they are written next to each assert."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(ctx):
    import torch

    from options_model_amd import nn_regressor as nnr
    return torch, nnr, torch.device("cuda", 0)


def _data(torch, dev, rows, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(rows, 7, generator=g)
    X[:, 0] = 0.0  # the constant feature normalises to zero
    y = (0.7 * X[:, 1] - 0.3 * X[:, 2] ** 2 + 0.1 * torch.randn(rows, generator=g))[:, None]
    return torch.cat([X, y], dim=1).float().contiguous().to(dev)


def _torch_grads(torch, net, batch):
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(net(batch[:, :7]), batch[:, 7:])
    loss.backward()
    return loss


def _flat_grads(torch, nnr, net):
    import copy
    g = copy.deepcopy(net)
    with torch.no_grad():
        for pg, p in zip(g.parameters(), net.parameters()):
            pg.copy_(p.grad)
    return nnr.flatten_params(g)


@pytest.mark.parametrize("hidden,layers", [(32, 2), (32, 3), (64, 2), (64, 3), (128, 2), (128, 3)])
@pytest.mark.parametrize("rows", [32, 100, 1000, 4096, 50_000, 100_000])
def test_gradients_and_loss_match_autograd(env, ctx, rows, hidden, layers):
    """After ONE Adam step from zero moments, m = (1 - beta1) * (grad + wd * w): the first-moment
    buffer exposes the kernel's gradient.  64 units: <= 1024 rows run the tile-per-wave kernel, more
    the workgroup kernel; 128 units: the tile-per-wave kernel (<= 8192 rows per step)."""
    torch, nnr, dev = env
    assert ctx.lib.omc_mlp_train_supported(hidden, layers, rows)
    torch.manual_seed(3)
    net = nnr.make_net(7, hidden, layers, 0.0).to(dev)
    data = _data(torch, dev, rows, 11)
    loss_t = _torch_grads(torch, net, data)
    gref = _flat_grads(torch, nnr, net)
    p0 = nnr.flatten_params(net)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, rows, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     0, 1e-3, 0.0, 5, weight_decay=0.0, hidden=hidden, layers=layers)
    assert step == 1 and p.numel() == ctx.lib.omc_mlp_param_count(hidden, layers)
    assert loss == pytest.approx(float(loss_t.detach()), rel=2e-5)
    g = (m / 0.1).cpu().numpy()
    ref = gref.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-5 * scale  # float32 sums over up to 50k rows, different order
    # first Adam step: w -= lr * g / (|g| + eps)
    expect = p0 - 1e-3 * gref / (gref.abs() + 1e-8)
    big = gref.abs() > 1e-6  # where g ~ 0 the sign is noise
    assert torch.allclose(p[big], expect[big], rtol=0, atol=2e-6)


@pytest.mark.parametrize("hidden,layers", [(32, 3), (64, 2), (128, 3)])
@pytest.mark.parametrize("rows,bs", [(1, 1), (2, 2), (31, 31), (33, 33), (65, 64), (97, 40), (1500, 1100)])
def test_ragged_tiny_batches(env, ctx, hidden, layers, rows, bs):
    """Edge sizes: fewer rows than a tile, one row over a tile, ragged last batches, on both kernels
    (64 units switch from the tile-per-wave to the workgroup kernel above 1024 rows per step)."""
    torch, nnr, dev = env
    torch.manual_seed(7)
    net = nnr.make_net(7, hidden, layers, 0.0).to(dev)
    data = _data(torch, dev, rows, 17)
    p = nnr.flatten_params(net)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, bs, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     0, 1e-3, 0.0, 5, hidden=hidden, layers=layers)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    tot, n = 0.0, 0
    for o in range(0, rows, bs):
        tot += float(_torch_grads(torch, net, data[o:o + bs]).detach())
        opt.step()
        n += 1
    assert step == n and bool(torch.isfinite(p).all())
    assert loss == pytest.approx(tot / n, rel=1e-4, abs=1e-7)


@pytest.mark.parametrize("hidden,layers,bs", [(32, 2, 1000), (32, 3, 2500), (64, 2, 1000), (64, 3, 1000), (64, 2, 2500), (128, 3, 1000),
                                              (128, 2, 2500)])
def test_many_steps_track_torch_adam(env, ctx, hidden, layers, bs):
    torch, nnr, dev = env
    torch.manual_seed(4)
    net = nnr.make_net(7, hidden, layers, 0.0).to(dev)
    rows = 10 * bs  # 10 steps, none ragged
    data = _data(torch, dev, rows, 12)
    p = nnr.flatten_params(net)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, bs, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     0, 1e-3, 0.0, 5, hidden=hidden, layers=layers)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    tot = 0.0
    for o in range(0, rows, bs):
        l_ = _torch_grads(torch, net, data[o:o + bs])
        opt.step()
        tot += float(l_.detach())
    assert step == 10
    assert loss == pytest.approx(tot / 10, rel=1e-4)
    ref = nnr.flatten_params(net)
    # Adam's sign-like early steps amplify rounding where a gradient component is ~0; compare in bulk
    diff = (p - ref).abs()
    assert float(diff.max()) <= 2.5e-3 and float(diff.mean()) <= 2e-5
    # second epoch continues the step count (bias correction) and the loss keeps falling
    loss2, step2 = ctx.mlp_train_epoch(data.data_ptr(), rows, bs, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                       step, 1e-3, 0.0, 5, hidden=hidden, layers=layers)
    assert step2 == 20 and loss2 < loss


def test_ragged_last_batch(env, ctx):
    torch, nnr, dev = env
    torch.manual_seed(5)
    net = nnr.make_net(7, 64, 2, 0.0).to(dev)
    rows, bs = 1000, 300  # 300, 300, 300, 100
    data = _data(torch, dev, rows, 13)
    p = nnr.flatten_params(net)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, bs, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     0, 1e-3, 0.0, 5)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    tot = 0.0
    for o in range(0, rows, bs):
        tot += float(_torch_grads(torch, net, data[o:o + bs]).detach())
        opt.step()
    assert step == 4
    assert loss == pytest.approx(tot / 4, rel=1e-4)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 1000, 65536, 65537, 1_000_003])
def test_shuffle_is_a_permutation(env, ctx, n):
    torch, nnr, dev = env
    out = torch.full((n,), -1, dtype=torch.int64, device=dev)
    # the fill is queued on torch's stream; the session context runs on its own.  The library orders an owned
    # stream after the device's default stream on entry (include/omc.h, "stream ordering"), and the test does
    # not lean on that alone: the fill must have landed before the shuffle writes the same buffer.
    torch.cuda.synchronize()
    ctx.mlp_shuffle_indices(n, 12345, out.data_ptr())
    idx = out.cpu().numpy()
    assert np.array_equal(np.sort(idx), np.arange(n))
    ctx.mlp_shuffle_indices(n, 0, out.data_ptr())
    assert np.array_equal(out.cpu().numpy(), np.arange(n))  # key 0: storage order
    if n >= 1000:
        out2 = torch.empty_like(out)
        torch.cuda.synchronize()
        ctx.mlp_shuffle_indices(n, 12346, out2.data_ptr())
        idx2 = out2.cpu().numpy()
        assert (idx != idx2).mean() > 0.99 and (idx != np.arange(n)).mean() > 0.99
        # no visible structure: rank correlation with the identity and between keys is ~0
        assert abs(np.corrcoef(idx, np.arange(n))[0, 1]) < 0.1 and abs(np.corrcoef(idx, idx2)[0, 1]) < 0.1


def test_shuffled_epoch_visits_every_row_once(env, ctx):
    """One full-batch step: the gradient is a sum over the rows, so any permutation gives the same
    step up to rounding; minibatches of a shuffled epoch equal minibatches of the gathered data."""
    torch, nnr, dev = env
    torch.manual_seed(6)
    net = nnr.make_net(7, 64, 2, 0.0).to(dev)
    rows = 50_000
    data = _data(torch, dev, rows, 14)
    p0 = nnr.flatten_params(net)
    res = []
    for key in (0, 77):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        torch.cuda.synchronize()
        loss, _ = ctx.mlp_train_epoch(data.data_ptr(), rows, rows, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                      0, 1e-3, 0.0, 5, weight_decay=0.0, shuffle_key=key)
        res.append((loss, m.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-5)
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * float(res[0][1].abs().max())
    # minibatches: shuffled in-kernel == storage order on the gathered copy, bit for bit
    idx = torch.empty(rows, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.mlp_shuffle_indices(rows, 77, idx.data_ptr())
    gathered = data[idx].contiguous()
    outs = []
    for d, key in ((data, 77), (gathered, 0)):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        torch.cuda.synchronize()
        loss, step = ctx.mlp_train_epoch(d.data_ptr(), rows, 4096, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                         0, 1e-3, 0.0, 5, shuffle_key=key)
        outs.append((loss, p.cpu().numpy()))
    assert step == 13 and outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("hidden,layers,rows", [(64, 2, 1 << 17), (64, 3, 1 << 17), (64, 2, 1024), (128, 3, 8192)])
def test_dropout_is_inverted_bernoulli_and_deterministic(env, ctx, hidden, layers, rows):
    """All weights 0, all hidden biases 1, output weights 1/64, output bias 0, target 0: out = mean_j
    keep_j / q over the last layer's H units, so the batch loss is E[out^2] = 1 + (1 - q) / (H q) for
    keep probability q = 1 - p."""
    torch, nnr, dev = env
    H = hidden
    data = torch.zeros(rows, 8, device=dev)
    n = ctx.lib.omc_mlp_param_count(H, layers)
    flat = torch.zeros(n, device=dev)
    flat[:H * 8].view(H, 8)[:, 7] = 1.0
    for j in range(layers - 1):
        o = H * 8 + j * (H * H + H)
        flat[o + H * H:o + H * H + H] = 1.0
    flat[n - H - 1:n - 1] = 1.0 / H
    for p_drop in (0.1, 0.5):
        q = 1 - p_drop
        res = []
        for seed in (1, 1, 2):
            p, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
            torch.cuda.synchronize()
            loss, _ = ctx.mlp_train_epoch(data.data_ptr(), rows, rows, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                          0, 1e-3, p_drop, seed, hidden=hidden, layers=layers)
            res.append((loss, p.cpu().numpy()))
        expect = 1 + (1 - q) / (H * q)
        sd = 2 * math.sqrt((1 - q) / (H * q)) / math.sqrt(rows)
        assert abs(res[0][0] - expect) < 6 * sd + 1e-4
        assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])  # same seed: same bits
        assert res[0][0] != res[2][0]


@pytest.mark.parametrize("n", [1, 7, 1000, 3_000_001])
def test_feature_statistics_match_numpy(env, ctx, n):
    """omc_nn_feature_stats against numpy float64 on the same rows (options_model_3.py:550-563)."""
    torch, nnr, dev = env
    rng = np.random.default_rng(n)
    x = rng.uniform(0.5, 1.0, n)
    t = rng.integers(1, 50, n).astype(np.int32)
    y = rng.exponential(3.0, n)
    T, dt = 1.0, 1.0 / 50
    s = np.sqrt(np.maximum(T - t * dt, 1e-6))
    F = np.stack([x, x * x, x ** 3, np.maximum(x - 1, 0), s, x * s, y])
    xd, td, yd = (torch.from_numpy(a).to(dev) for a in (x, t, y))
    torch.cuda.synchronize()
    mean, var = ctx.nn_feature_stats(xd.data_ptr(), td.data_ptr(), yd.data_ptr(), n, T, dt)
    assert np.allclose(mean, F.mean(axis=1), rtol=1e-12, atol=1e-15)
    assert np.allclose(var, F.var(axis=1), rtol=1e-9, atol=1e-18)
    fm, fs, ym, ysd = nnr.normalisers(xd, td, yd, T, dt)
    assert fm[0] == 1.0 and fs[0] == 1.0           # the constant column: std 0 -> 1 (:562)
    assert fs[4] == 1.0                             # max(x-1,0) is all zero here: std 0 -> 1
    if n > 1:
        assert float(ysd) == pytest.approx(y.std(), rel=1e-9) and float(ym) == pytest.approx(y.mean(), rel=1e-12)
    else:
        assert float(ysd) == 1.0 and bool((fs == 1.0).all())


def test_pass2_kernel_matches_torch_sweep(env, ctx):
    """Same trained-ish net, same paths, dropout off: the exercise decisions of omc_lsm_apply_mlp
    are those of the torch sweep except where payoff and continuation agree to float32 rounding."""
    torch, nnr, dev = env
    from options_model_amd import _ffi
    torch.manual_seed(8)
    M, N, K, r, T = 20_000, 30, 100.0, 0.05, 1.0
    S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
    c2 = nnr._ctx_on_torch_stream(0)
    nnr.generate_paths(c2, S, dict(model="gbm"), 100.0, r, 0.2, T, 21)
    x, t, y, _ = nnr.collect_rows(S, K, r, T, True)
    fm, fs, ym, ysd = nnr.normalisers(x, t, y, T, T / N)
    net = nnr.make_net(7, 64, 2, 0.1).to(dev)
    nnr.train(net, x, t, y, fm, fs, ym, ysd, T, T / N, 3, 1e-3)
    cf, ex = nnr.pass2(S, K, r, T, True, net, fm, fs, ym, ysd, dropout_on=False)
    out = nnr.pass2_fused(S, K, r, T, True, net, fm, fs, ym, ysd, dropout_on=False, want_state=True)
    ref_price = float(cf.mean())
    assert out["price"] == pytest.approx(ref_price, rel=2e-4)
    exn = ex.cpu().numpy()
    assert abs(int(out["n_exercised"]) - int(exn.sum())) <= 5e-4 * M
    assert ((out["tex"] < N) != exn).mean() <= 5e-4
    # value of each path at t = dt from (sx, tex) reproduces the torch cash-flows where decisions agree
    disc = np.exp(-r * (T / N) * (out["tex"].astype(np.float64) - 1))
    cf_h = np.maximum(K - out["sx"].astype(np.float64), 0) * disc
    same = np.isclose(cf_h, cf.cpu().numpy(), rtol=1e-9, atol=1e-12)
    assert same.mean() >= 1 - 1e-3
    # dropout on: another stream of masks than torch's, same distribution -> same price level
    on = nnr.pass2_fused(S, K, r, T, True, net, fm, fs, ym, ysd, dropout_on=True)
    cf_t, _ = nnr.pass2(S, K, r, T, True, net, fm, fs, ym, ysd, dropout_on=True)
    assert on["price"] == pytest.approx(float(cf_t.mean()), abs=0.05)
    assert on["pass2"] == "hip"


@pytest.mark.parametrize("hidden,layers", [(32, 2), (32, 3), (64, 2), (64, 3), (128, 3)])
def test_flatten_unflatten_roundtrip(env, hidden, layers):
    torch, nnr, dev = env
    torch.manual_seed(1)
    a = nnr.make_net(7, hidden, layers, 0.1).to(dev)
    b = nnr.make_net(7, hidden, layers, 0.1).to(dev)
    flat = nnr.flatten_params(a)
    nnr.unflatten_params(b, flat)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    assert nnr.fused_trainer_supports(a, 256) and nnr.fused_trainer_supports(a, 1 << 17)


def test_three_hidden_layers_train_and_price_through_the_kernels(env, ctx):
    """SingleLSMNet(7, 64, 3) -- the depth the reference's class always has -- end to end on the
    fused trainer and the pass-2 kernel."""
    torch, nnr, dev = env
    kw = dict(seed=9, nn_epochs=6, nn_layers=3, inference_dropout=False)  # eval-mode pass 2: no mask noise
    r = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, nn_trainer="hip", **kw)
    assert r.info["trainer"] == "hip" and r.info["pass2"] == "hip"
    t = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, nn_trainer="torch", **kw)
    assert r.info["best_loss"] == pytest.approx(t.info["best_loss"], rel=5e-3)  # same optimisation problem
    # two equally good fits (same loss) still place the exercise boundary differently: the reference
    # itself moves by 0.48 between seeds on this flow (tests/golden/scalars.json), more at this size
    assert abs(r.price - t.price) < 0.8 and 5.5 < r.price < 8.0


def test_unsupported_shapes_and_bad_arguments(env, ctx):
    torch, nnr, dev = env
    lib = ctx.lib
    assert lib.omc_mlp_param_count(64, 2) == 4737
    assert lib.omc_mlp_param_count(128, 3) == 128 * 8 + 2 * (128 * 128 + 128) + 128 + 1
    assert lib.omc_mlp_param_count(32, 3) == 32 * 8 + 2 * (32 * 32 + 32) + 32 + 1  # (32 units: round 6)
    assert lib.omc_mlp_param_count(256, 2) == -1 and lib.omc_mlp_param_count(64, 4) == -1 and lib.omc_mlp_param_count(96, 3) == -1
    d = torch.zeros(64, 8, device=dev)
    p = torch.zeros(4737, device=dev)
    with pytest.raises(ValueError, match="hidden = 32, 64 or 128"):
        ctx.mlp_train_epoch(d.data_ptr(), 64, 64, p.data_ptr(), p.data_ptr(), p.data_ptr(), 0, 1e-3, 0.0, 1,
                            hidden=256, layers=3)
    assert lib.omc_mlp_train_supported(32, 3, 256) == 1 and lib.omc_mlp_train_supported(32, 2, 1 << 17) == 1
    assert lib.omc_mlp_train_supported(256, 3, 256) == 0
    assert lib.omc_mlp_train_supported(128, 3, 8192) == 1 and lib.omc_mlp_train_supported(128, 3, 1 << 20) == 1
    assert lib.omc_mlp_train_supported(64, 2, 1 << 20) == 1 and lib.omc_mlp_train_supported(64, 4, 256) == 0
    with pytest.raises(ValueError, match="dropout"):
        ctx.mlp_train_epoch(d.data_ptr(), 64, 64, p.data_ptr(), p.data_ptr(), p.data_ptr(), 0, 1e-3, 1.0, 1)
    assert nnr.fused_trainer_supports(nnr.make_net(7, 128, 3, 0.1), 256)
    assert nnr.fused_trainer_supports(nnr.make_net(7, 128, 3, 0.1), 1 << 17)
    assert nnr.fused_apply_supports(nnr.make_net(7, 128, 3, 0.1))
    with pytest.raises(ValueError, match="covers"):
        nnr.train(nnr.make_net(7, 96, 3, 0.1).to(dev), torch.ones(10, device=dev, dtype=torch.float64),
                  torch.ones(10, device=dev, dtype=torch.int32), torch.ones(10, device=dev, dtype=torch.float64),
                  torch.zeros(7, device=dev, dtype=torch.float64), torch.ones(7, device=dev, dtype=torch.float64),
                  torch.zeros((), device=dev, dtype=torch.float64), torch.ones((), device=dev, dtype=torch.float64),
                  1.0, 0.1, 1, 1e-3, trainer="hip")


def test_fused_and_torch_trainers_price_alike(env, ctx):
    """Same paths, same rows: the two trainers differ only in minibatch order, dropout streams and
    rounding -> prices agree to Monte-Carlo + training noise (the reference itself moves by ~0.3
    between seeds on this flow, tests/golden/scalars.json reference_nn_seed_band)."""
    torch, nnr, dev = env
    out = {}
    for tr in ("hip", "torch"):
        r = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=9, nn_epochs=6,
                                         nn_trainer=tr)
        out[tr] = r
    assert abs(out["hip"].price - out["torch"].price) < 0.25
    assert 5.5 < out["hip"].price < 8.0
