"""The DEFAULT branch of the NN regressor -- dropout on, in training (options_model_3.py:85-103, 576-586) and at
inference (:637-640: the reference never calls .eval()) -- against the oracle under IDENTICAL masks.

oracle/dropout.py restates in numpy which units the kernels keep (Philox block -> multiply-with-carry streams -> 16
bits per unit, per kernel the register -> unit order).  Here:
  1. the device's own masks (omc_mlp_dropout_masks: the kernels' relu_dropout* functions on activations of 1.0) equal
     the oracle's bit for bit, for every trainer kernel, pass 2, both generator flavours, ragged row counts, row keys;
  2. every trainer kernel's loss and gradient (exposed by Adam's first-moment buffer after one step from zero moments)
     equal PyTorch autograd through a functional forward pass with THOSE masks injected (relu(z) * mask / keep), at
     p = 0.1 and p = 0.5, 2e-5 of the largest gradient component -- the tolerance of the dropout-free tests in
     test_gpu_mlp.py; likewise a later optimizer step (other masks), the sharded path (masks keyed by the row's
     position in the GLOBAL minibatch) and, through a subprocess, the one-tile-per-wave kernel at 64 units;
  3. pass 2 with dropout ON returns the oracle's decisions (oracle.reference_flow.two_pass_frozen_mlp_regressor with
     the same masks) on the reference's own trained networks and paths (tests/golden/v3_frozen_nn*.npz).
Config 5 at its own size under dropout: tests/test_gpu_nn_full.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dropout as dr
from oracle import reference_flow as rf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env(ctx):
    import torch

    from options_model_amd import nn_regressor as nnr
    return torch, nnr, torch.device("cuda", 0)


# ---------------------------------------------------------------- 1. the masks themselves
@pytest.mark.parametrize("variant,hidden", [(dr.GROUP, 64), (dr.TILE, 64), (dr.TILE, 128), (dr.QUAD, 32), (dr.QUAD, 64),
                                            (dr.QUAD, 128), (dr.Q16, 64), (dr.Q16, 128)])
@pytest.mark.parametrize("p_drop", [0.1, 0.5])
def test_trainer_masks_on_the_device_equal_the_oracle(ctx, variant, hidden, p_drop):
    layers = 3
    for n_rows, step, seed in ((1, 1, 5), (33, 2, 2 ** 62 - 7), (1000, 22_050, 0x1234_5678_9ABC_DEF0)):
        got = ctx.mlp_dropout_masks(variant, hidden, layers, n_rows, step, seed, p_drop)
        want = dr.train_masks(variant, hidden, layers, np.arange(n_rows), step, seed, p_drop)
        assert got.shape == want.shape == (layers, n_rows, hidden)
        assert np.array_equal(got, want), (variant, hidden, n_rows, int((got != want).sum()))
    # row keys (sharded training: positions in the global minibatch), not 0, 1, 2, ...
    keys = np.random.default_rng(3).integers(0, 1 << 17, 257).astype(np.uint32)
    got = ctx.mlp_dropout_masks(variant, hidden, layers, 257, 9, 77, p_drop, keys=keys)
    assert np.array_equal(got, dr.train_masks(variant, hidden, layers, keys, 9, 77, p_drop))
    # the rate: 257 * layers * hidden Bernoulli(keep16 / 65536) draws
    q = dr.keep16_of(p_drop) / 65536.0
    assert abs(got.mean() - q) < 5 * np.sqrt(q * (1 - q) / got.size)


@pytest.mark.parametrize("hidden,layers", [(32, 2), (32, 3), (64, 2), (64, 3), (128, 2), (128, 3)])
def test_pass2_masks_on_the_device_equal_the_oracle(ctx, hidden, layers):
    cols = np.concatenate([np.arange(0, 300), np.arange(500_000, 500_300), [2 ** 32 - 1]]).astype(np.int64)
    for t, seed, p_drop in ((1, 11, 0.1), (251, 2 ** 61 + 3, 0.1), (17, 99, 0.5)):
        got = ctx.mlp_dropout_masks(0, hidden, layers, cols.size, t, seed, p_drop, keys=cols.astype(np.uint32))
        assert np.array_equal(got, dr.apply_masks(hidden, layers, cols, t, seed, p_drop))
    # no dropout: everything is kept (the kernels' uniform branch)
    assert ctx.mlp_dropout_masks(0, hidden, layers, 64, 3, 1, 0.0).all()


def test_probe_rejects_shapes_without_a_kernel(ctx):
    with pytest.raises(ValueError):
        ctx.mlp_dropout_masks(dr.GROUP, 128, 2, 32, 1, 1, 0.1)  # the workgroup kernel exists for 64 units only
    with pytest.raises(ValueError):
        ctx.mlp_dropout_masks(7, 64, 2, 32, 1, 1, 0.1)
    with pytest.raises(ValueError):
        ctx.mlp_dropout_masks(0, 64, 2, 32, 1, 1, 1.0)


# ---------------------------------------------------------------- 2. gradients through the masks
def _data(torch, dev, rows, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(rows, 7, generator=g)
    X[:, 0] = 0.0  # the constant feature normalises to zero
    y = (0.7 * X[:, 1] - 0.3 * X[:, 2] ** 2 + 0.1 * torch.randn(rows, generator=g))[:, None]
    return torch.cat([X, y], dim=1).float().contiguous().to(dev)


def _masked_loss(torch, net, batch, masks, p_drop, denom=None):
    """SingleLSMNet.forward in training mode with the Bernoulli draws of nn.Dropout replaced by `masks`
    ([layers][rows][hidden] bool): h = relu(W h + b) * mask / keep.  denom: minibatch size of the MSE mean."""
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    inv = float(dr.inv_keep_of(p_drop))
    h = batch[:, :7]
    for j, l_ in enumerate(lin[:-1]):
        h = torch.relu(l_(h)) * (torch.from_numpy(masks[j]).to(h.device).float() * inv)
    out = lin[-1](h)
    sq = (out - batch[:, 7:]) ** 2
    return sq.sum() / (denom if denom is not None else batch.shape[0])


def _flat_grads(torch, nnr, net):
    import copy
    g = copy.deepcopy(net)
    with torch.no_grad():
        for pg, p in zip(g.parameters(), net.parameters()):
            pg.copy_(p.grad)
    return nnr.flatten_params(g)


def rows_clear_of_relu_boundaries(torch, ctx, net, data, hidden, layers, p_drop, step, seed, width=1e-5):
    """`data` with every row replaced (by fresh random data) on which a hidden pre-activation lies within `width` of zero
    in a float64 forward pass under the masks the kernel will draw.  A unit that close to its ReLU kink may round to either
    side in float32 -- PyTorch's float32 autograd and the kernels then differ by that row's whole contribution, ~1 / rows
    of the gradient's scale: measured 2e-5 ... 6e-5 at 20,000 rows, 1e-5 at 50,000 (profiles/r05_fuzz_soak.txt: against
    autograd in float64 sometimes the kernel is the odd one out, sometimes PyTorch).  The real rounding differences are
    ~1e-7 wide; the margin taken here costs ~0.5 % of the rows."""
    import copy
    net64 = copy.deepcopy(net).double()
    lin = [m for m in net64.net if isinstance(m, torch.nn.Linear)]
    inv = float(dr.inv_keep_of(p_drop))
    n = data.shape[0]
    variant = ctx.lib.omc_mlp_train_variant(hidden, layers, n)
    masks = dr.train_masks(variant, hidden, layers, np.arange(n), step + 1, seed, p_drop)
    data = data.clone()
    for it in range(8):
        with torch.no_grad():
            h = data[:, :7].double()
            amb = torch.zeros(n, dtype=torch.bool, device=data.device)
            for j, l_ in enumerate(lin[:-1]):
                z = l_(h)
                amb |= (z.abs() < width).any(dim=1)
                h = torch.relu(z) * (torch.from_numpy(masks[j]).to(data.device).double() * inv)
        k = int(amb.sum())
        if k == 0:
            return data, variant
        data[amb] = _data(torch, data.device, k, 1000 + it)
    raise AssertionError("rows keep landing on a ReLU boundary")


def _check_one_step(env, ctx, hidden, layers, rows, p_drop, variant, first_step=0, seed=5, data=None, net=None):
    torch, nnr, dev = env
    assert ctx.lib.omc_mlp_train_variant(hidden, layers, rows) == variant
    if net is None:
        torch.manual_seed(3)
        net = nnr.make_net(7, hidden, layers, p_drop).to(dev)
    if data is None:
        data = _data(torch, dev, rows, 11)
    assert data.shape[0] == rows
    masks = dr.train_masks(variant, hidden, layers, np.arange(rows), first_step + 1, seed, p_drop)
    net.zero_grad(set_to_none=True)
    loss_t = _masked_loss(torch, net, data, masks, p_drop)
    loss_t.backward()
    gref = _flat_grads(torch, nnr, net).cpu().numpy()
    p0 = nnr.flatten_params(net)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, rows, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     first_step, 1e-3, p_drop, seed, weight_decay=0.0, hidden=hidden, layers=layers)
    assert step == first_step + 1
    assert loss == pytest.approx(float(loss_t.detach()), rel=2e-5)
    g = (m / 0.1).cpu().numpy()  # m = (1 - beta1) * grad after one step from zero moments
    scale = np.abs(gref).max()
    err = np.abs(g - gref).max()
    assert err <= 2e-5 * scale, (err / scale, variant, hidden, layers, rows, p_drop)
    # the comparison has teeth: the gradient under a neighbouring step's masks is far outside the tolerance
    other = dr.train_masks(variant, hidden, layers, np.arange(rows), first_step + 2, seed, p_drop)
    net.zero_grad(set_to_none=True)
    _masked_loss(torch, net, data, other, p_drop).backward()
    g_other = _flat_grads(torch, nnr, net).cpu().numpy()
    assert np.abs(g_other - gref).max() > 10 * 2e-5 * scale


@pytest.mark.parametrize("p_drop", [0.1, 0.5])
@pytest.mark.parametrize("hidden,layers,rows,variant", [
    (32, 2, 1, dr.QUAD), (32, 3, 100, dr.QUAD), (32, 2, 256, dr.QUAD), (32, 3, 4097, dr.QUAD), (32, 2, 1 << 17, dr.QUAD),
    (64, 2, 16, dr.Q16), (64, 2, 100, dr.Q16), (64, 3, 1000, dr.Q16), (64, 3, 1024, dr.Q16),
    (128, 3, 1, dr.Q16), (128, 3, 17, dr.Q16), (128, 3, 256, dr.Q16), (128, 2, 1000, dr.Q16), (128, 2, 1024, dr.Q16),
    (128, 3, 1025, dr.Q16), (128, 3, 4096, dr.Q16), (128, 3, 4097, dr.QUAD), (128, 2, 8192, dr.QUAD),
    (64, 2, 1025, dr.GROUP), (64, 2, 4096, dr.GROUP), (64, 3, 4096, dr.GROUP), (64, 2, 50_000, dr.GROUP),
    (64, 3, 100_000, dr.GROUP), (64, 2, 1 << 17, dr.GROUP),
    (128, 3, 8193, dr.TILE), (128, 2, 20_000, dr.TILE), (128, 3, 50_000, dr.TILE),
])
def test_masked_gradients_and_loss_match_autograd(env, ctx, hidden, layers, rows, variant, p_drop):
    _check_one_step(env, ctx, hidden, layers, rows, p_drop, variant)


@pytest.mark.parametrize("hidden,layers,rows,variant", [(128, 3, 256, dr.Q16), (128, 3, 2048, dr.Q16), (128, 3, 6000, dr.QUAD), (64, 2, 4096, dr.GROUP),
                                                        (128, 3, 10_000, dr.TILE)])
def test_a_later_optimizer_step_draws_its_own_masks(env, ctx, hidden, layers, rows, variant):
    """`step` counts over the whole run (first_step + k): step 22,000 of the reference's default call."""
    _check_one_step(env, ctx, hidden, layers, rows, 0.1, variant, first_step=21_999, seed=2 ** 61 + 12345)


@pytest.mark.parametrize("hidden,layers,bs,variant", [(32, 3, 256, dr.QUAD), (128, 3, 256, dr.Q16), (128, 2, 2048, dr.Q16), (128, 2, 5000, dr.QUAD),
                                                      (64, 2, 2048, dr.GROUP)])
def test_several_steps_track_torch_adam_under_the_same_masks(env, ctx, hidden, layers, bs, variant):
    """Ten optimizer steps of an epoch (masks of steps 1..10, rows in storage order) against torch.optim.Adam fed the
    masked losses: the weights stay together as in the dropout-free test (test_gpu_mlp.py), i.e. masks, 1 / keep and
    Adam agree step after step, not only at step 1."""
    torch, nnr, dev = env
    p_drop, seed = 0.1, 77
    torch.manual_seed(4)
    net = nnr.make_net(7, hidden, layers, p_drop).to(dev)
    rows = 10 * bs
    data = _data(torch, dev, rows, 12)
    p = nnr.flatten_params(net)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, bs, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                     0, 1e-3, p_drop, seed, hidden=hidden, layers=layers)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    tot = 0.0
    for k, o in enumerate(range(0, rows, bs)):
        masks = dr.train_masks(variant, hidden, layers, np.arange(bs), k + 1, seed, p_drop)
        opt.zero_grad(set_to_none=True)
        l_ = _masked_loss(torch, net, data[o:o + bs], masks, p_drop)
        l_.backward()
        opt.step()
        tot += float(l_.detach())
    assert step == 10
    assert loss == pytest.approx(tot / 10, rel=1e-4)
    diff = (p - nnr.flatten_params(net)).abs()
    assert float(diff.max()) <= 2.5e-3 and float(diff.mean()) <= 2e-5  # the dropout-free test's bounds


@pytest.mark.parametrize("hidden,layers,batch,variant_global", [(64, 2, 8192, dr.GROUP), (128, 3, 256, dr.Q16),
                                                                (64, 2, 1500, dr.GROUP), (128, 3, 4096, dr.Q16),
                                                                (128, 3, 6000, dr.QUAD), (128, 3, 8000, dr.QUAD)])
def test_sharded_step_draws_the_masks_of_the_global_minibatch(env, ctx, hidden, layers, batch, variant_global):
    """One rank's part of a global minibatch (omc_mlp_train_epoch_sharded without a communicator = the sum of one
    rank): its rows carry their positions in the GLOBAL minibatch as dropout keys, the loss is scaled by the global
    size.  Gradient = autograd of sum_own (o - y)^2 / B_global under the masks of those positions -- drawn by the kernel
    the UNSHARDED run picks for the global minibatch (a rank used to pick by its own share, so a rank holding 937
    of 1,500 rows ran the 16-row kernel while the unsharded run -- and a peer with a larger share -- ran another one, with
    another unit map: masks that are not the unsharded run's).  Cases 3 and 5 are such shares."""
    torch, nnr, dev = env
    p_drop, seed = 0.1, 31
    rng = np.random.default_rng(8)
    pos = np.sort(rng.choice(batch, size=batch * 5 // 8, replace=False)).astype(np.uint32)  # this rank's positions
    n_loc = pos.size
    assert ctx.lib.omc_mlp_train_variant(hidden, layers, batch) == variant_global
    if (hidden, layers, batch) in ((64, 2, 1500), (128, 3, 6000)):
        assert ctx.lib.omc_mlp_train_variant(hidden, layers, n_loc) != variant_global  # the rank's own share would pick another
    variant_local = variant_global
    torch.manual_seed(6)
    net = nnr.make_net(7, hidden, layers, p_drop).to(dev)
    data = _data(torch, dev, n_loc, 15)
    masks = dr.train_masks(variant_local, hidden, layers, pos, 1, seed, p_drop)
    net.zero_grad(set_to_none=True)
    loss_t = _masked_loss(torch, net, data, masks, p_drop, denom=batch)
    loss_t.backward()
    gref = _flat_grads(torch, nnr, net).cpu().numpy()
    p0 = nnr.flatten_params(net)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    dpos = torch.from_numpy(pos.astype(np.int64)).to(dev).to(torch.int32)  # uint32 values < 2^31
    torch.cuda.synchronize()
    loss, step = ctx.mlp_train_epoch_sharded(data.data_ptr(), n_loc, batch, batch, p.data_ptr(), m.data_ptr(), v.data_ptr(),
                                             0, 1e-3, p_drop, seed, np.array([0, n_loc], np.int64), dpos.data_ptr(),
                                             hidden=hidden, layers=layers, weight_decay=0.0)
    assert step == 1
    assert loss == pytest.approx(float(loss_t.detach()), rel=2e-5)
    g = (m / 0.1).cpu().numpy()
    scale = np.abs(gref).max()
    assert np.abs(g - gref).max() <= 2e-5 * scale
    # keyed by the LOCAL row index instead, the gradient would be another one
    wrong = dr.train_masks(variant_local, hidden, layers, np.arange(n_loc), 1, seed, p_drop)
    net.zero_grad(set_to_none=True)
    _masked_loss(torch, net, data, wrong, p_drop, denom=batch).backward()
    assert np.abs(_flat_grads(torch, nnr, net).cpu().numpy() - gref).max() > 50 * 2e-5 * scale


_CHILD_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
from options_model_amd import _ffi, nn_regressor as nnr
from oracle import dropout as dr
sys.path.insert(0, {root!r} + "/tests")
import test_gpu_dropout as T
ctx = _ffi.Context(0)
env = (torch, nnr, torch.device("cuda", 0))
for hidden, layers, rows, p, variant in json.loads(os.environ["CASES"]):
    assert ctx.lib.omc_mlp_train_variant(hidden, layers, rows) == variant, (hidden, layers, rows, variant)
    T._check_one_step(env, ctx, hidden, layers, rows, p, variant)
print("child ok")
"""


@pytest.mark.parametrize("knobs,cases", [
    (dict(OMC_MLP_Q16="0"), [(64, 2, 100, 0.1, dr.QUAD), (64, 3, 1000, 0.5, dr.QUAD), (128, 3, 256, 0.1, dr.QUAD), (128, 2, 31, 0.5, dr.QUAD)]),
    (dict(OMC_MLP_Q16="0", OMC_MLP_QUAD="0"), [(64, 2, 100, 0.1, dr.TILE), (64, 3, 1000, 0.5, dr.TILE), (64, 2, 1024, 0.1, dr.TILE),
                                               (128, 3, 256, 0.1, dr.TILE)]),
    (dict(OMC_MLP_Q16="1024"), [(128, 3, 1024, 0.1, dr.Q16), (128, 2, 2000, 0.5, dr.QUAD)]),
])
def test_kernels_behind_the_environment_switches_under_masks(env, knobs, cases):
    """Small minibatches ran the 32-row one-tile-per-workgroup kernel before the 16-row tiles existed, and the
    one-tile-per-wave kernel before that; both stay selectable (OMC_MLP_Q16 / OMC_MLP_QUAD, read once per process) and
    the batched trainers still run the 32-row kernel -- hence child processes (started, never exec'd into)."""
    import json
    e = dict(os.environ, PYTHONPATH=ROOT, CASES=json.dumps(cases), **knobs)
    r = subprocess.run([sys.executable, "-c", _CHILD_SCRIPT.format(root=ROOT)], env=e, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------- 3. pass 2 with dropout on
def _load_net(torch, nn, tag, hidden):
    from options_model_amd import nn_regressor as nr
    net = nr.make_net(7, int(hidden), 3, 0.1)
    state = {k[len(tag) + 4:]: torch.from_numpy(nn[k]) for k in nn.files if k.startswith(f"{tag}_sd_")}
    net.load_state_dict(state)
    return net.cuda()


def _cashflows(out, K, r, T, N, is_put):
    sx = out["sx"].astype(np.float64)
    pay = np.maximum((K - sx) if is_put else (sx - K), 0)
    return pay * np.exp(-r * (T / N) * (out["tex"].astype(np.float64) - 1))


@pytest.mark.parametrize("fixture,tag", [("nn", "gbm_put"), ("nn", "heston_call"), ("nn_heston_put", "heston_put")])
@pytest.mark.parametrize("p_drop,seed", [(0.1, 2 ** 61 + 17), (0.5, 4)])
def test_dropout_on_pass2_returns_the_oracles_decisions_on_the_references_nets(env, golden, fixture, tag, p_drop, seed):
    """The reference's trained network (3 x 128 for gbm_put -- its default shape), normalisers and paths; the net as the
    reference runs it at inference: dropout ACTIVE (options_model_3.py:637-640).  mlp_apply_kernel vs the oracle's sticky
    sweep (:615-651) whose continuation values come from the float32 numpy forward pass under the same masks.  Same
    allowance as in eval mode (test_gpu_nn.py): <= 3 paths whose payoff sits within float32 rounding of the network's
    output -- but here a flipped path also changes WHEN it is asked again, so moved exercise times are counted too."""
    torch, nnr, dev = env
    nn = golden[fixture]
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    hidden, is_put = int(hidden), bool(is_put)
    S = torch.from_numpy(nn[f"{tag}_S"]).float().cuda().contiguous()
    N, M = S.shape[0] - 1, S.shape[1]
    net = _load_net(torch, nn, tag, hidden)
    for mod in net.net:
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p_drop
    fm, fs = nn[f"{tag}_feat_mean"], nn[f"{tag}_feat_std"]
    ym, ysd = (float(v) for v in nn[f"{tag}_Y_mean_std"])
    f64 = dict(dtype=torch.float64, device=dev)
    hip = nnr.pass2_fused(S, K, r, T, is_put, net, torch.tensor(fm, **f64), torch.tensor(fs, **f64),
                          torch.tensor(ym, **f64), torch.tensor(ysd, **f64), dropout_on=True, want_state=True, seed=seed)
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    regress, predict = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm, fs, ym, ysd,
                                                        dropout=dict(p=p_drop, seed=seed, hidden=hidden, layers=3))
    S64 = S.cpu().numpy().astype(np.float64)
    cf, ex, _ = rf.lsm_two_pass(S64, K, r, T, is_put, regress, predict)
    ex_hip = hip["tex"] < N
    flips = int((ex_hip != ex).sum())
    moved = int((np.abs(_cashflows(hip, K, r, T, N, is_put) - cf) > 2e-5).sum())
    price_o = float(cf.mean())
    print(f"{tag} p={p_drop}: oracle {price_o:.6f} hip {hip['price']:.6f} flips {flips} moved {moved} of {M}; "
          f"eval-mode fixture {float(nn[f'{tag}_price_eval']):.6f}, the reference's own dropout-on run "
          f"{float(nn[f'{tag}_price_ref']):.6f}")
    assert flips <= 3 and moved <= 5, (flips, moved)
    assert abs(hip["price"] - price_o) <= 2e-4 * max(price_o, 1.0)
    # and the masks matter: eval mode decides differently on far more paths than the allowance
    ev = nnr.pass2_fused(S, K, r, T, is_put, net, torch.tensor(fm, **f64), torch.tensor(fs, **f64),
                         torch.tensor(ym, **f64), torch.tensor(ysd, **f64), dropout_on=False, want_state=True)
    if is_put:
        assert int((np.abs(_cashflows(ev, K, r, T, N, is_put) - cf) > 2e-5).sum()) > 50


def test_shard_keys_give_a_shard_the_masks_of_the_unsharded_matrix(env, golden):
    """omc_lsm_apply_mlp_shard: columns [0, m) and [P, P + m) of a matrix priced on their own with col_bases = (0, P) take
    the decisions they take inside the whole matrix (bit for bit), and the oracle keyed by the unsharded column
    agrees."""
    torch, nnr, dev = env
    nn = golden["nn"]
    tag = "gbm_put"
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    hidden = int(hidden)
    S = torch.from_numpy(nn[f"{tag}_S"]).float().cuda().contiguous()
    N, M = S.shape[0] - 1, S.shape[1]
    P, m = M // 2, 200
    assert m <= P  # (the fixture holds 1,024 paths: columns [0, 200) and [512, 712))
    net = _load_net(torch, nn, tag, hidden)
    fm, fs = nn[f"{tag}_feat_mean"], nn[f"{tag}_feat_std"]
    ym, ysd = (float(v) for v in nn[f"{tag}_Y_mean_std"])
    c = nnr._ctx_on_torch_stream(0)
    params = nnr.flatten_params(net)
    torch.cuda.synchronize()
    whole = c.lsm_apply_mlp(S.data_ptr(), S.stride(0), M, N, K, r, T, True, params.data_ptr(), fm, fs, ym, ysd, 0.1, 123,
                            want_state=True, hidden=hidden, layers=3)
    cols = np.concatenate([np.arange(m), np.arange(P, P + m)])
    assert cols.max() < M
    Sp = S[:, torch.from_numpy(cols).to(dev)].contiguous()
    torch.cuda.synchronize()
    part = c.lsm_apply_mlp(Sp.data_ptr(), Sp.stride(0), 2 * m, N, K, r, T, True, params.data_ptr(), fm, fs, ym, ysd, 0.1,
                           123, want_state=True, hidden=hidden, layers=3, col_bases=(0, P))
    assert np.array_equal(part["tex"], whole["tex"][cols]) and np.array_equal(part["sx"], whole["sx"][cols])
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    regress, predict = rf.two_pass_frozen_mlp_regressor(
        K, T, N, state, fm, fs, ym, ysd, dropout=dict(p=0.1, seed=123, hidden=hidden, layers=3, col_of=lambda j: cols[j]))
    cf, ex, _ = rf.lsm_two_pass(Sp.cpu().numpy().astype(np.float64), K, r, T, True, regress, predict)
    assert int(((part["tex"] < N) != ex).sum()) <= 1
    assert int((np.abs(_cashflows(part, K, r, T, N, True) - cf) > 2e-5).sum()) <= 2
