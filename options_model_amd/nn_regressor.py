"""NN continuation-value regressor (BASELINE config 5): the reference's own v3 scheme --
ONE global MLP on 7 features, trained on every in-the-money (t, path) of pass 1, then a sticky
pass 2 -- run on the MI355X.  Paths come from the hand-written HIP kernels, written straight
into the torch tensor; the network BASELINE config 5 names (7 -> 64 -> 64 -> 1) is trained by the
library's own fused MFMA forward/backward + Adam kernels (omc_mlp_train_epoch, csrc/omc_mlp.hip);
PyTorch-ROCm holds the tensors, builds the rows and trains any other SingleLSMNet shape
(north_star: "PyTorch-ROCm only for the optional NN continuation-value regressor").

Reference lines mirrored (options_model_3/options_model_3.py):
    SingleLSMNet                         :85-103   (Linear/ReLU/Dropout stacks, 3 hidden layers)
    create_regression_features           :105-121  [1, x, x^2, x^3, max(x-1,0), s, x*s]
    pass 1: rows = every ITM (t, path), target = discounted TERMINAL payoff      :482-516
    target / feature normalisation (population std, zero std -> 1)               :550-563
    Adam(lr, wd 1e-5), MSE, ReduceLROnPlateau(pat 5, x0.5, min 1e-6), <= nn_epochs,
    early stop after 8 non-improving epochs (delta 1e-6), best-weights restore    :565-613
    pass 2: sticky mask, strict >, dropout still ACTIVE (no .eval())   :615-649
    mean of cash-flows valued at t = dt                                          :651

Deviations, all forced by scale and stated in the result dict:
  * minibatch size: the reference's min(256, R) is kept while R <= 2**18 rows; above that the
    batch grows so an epoch stays ~1024 optimizer steps (the reference's own GPU draft does
    the same thing with 512..8192, option_model_3_gpu.py:751-755).  At config 5, R ~ 1.15e8.
  * pass 2 evaluates the net densely on all M paths per step and masks, instead of gathering
    the active subset: no host sync inside the time loop.
  * torch's GPU generator replaces the CPU generator: same distributions, different streams;
    the fused trainer shuffles each epoch with a keyed Feistel permutation evaluated in the
    kernel (instead of randperm + gather) and draws its dropout bits from Philox-seeded streams.
"""
from __future__ import annotations

import copy
import math
import time

from . import _ffi
from .api import heston_defaults

_ctx_cache = {}
_warned = set()


def _warn_once(key: str, msg: str):
    """A shape the hand-written kernels do not cover runs through PyTorch-ROCm: correct, and several times slower.
    Say so once per process and kind (the result's info["trainer"] / ["pass2"] says it every time)."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _ffi.OmcError("regressor='nn' needs PyTorch-ROCm with a visible GPU")
    return torch


def _ctx_on_torch_stream(device: int):
    torch = _torch()
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    c = _ctx_cache.get(key)
    if c is None:
        c = _ffi.Context(device, stream=key[1] or None)
        _ctx_cache[key] = c
    return c


def make_net(input_dim=7, hidden_dim=128, num_layers=3, dropout=0.1):
    torch = _torch()
    nn = torch.nn
    layers = [nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout)]
    for _ in range(num_layers - 1):
        layers += [nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout)]
    layers.append(nn.Linear(hidden_dim, 1))

    class SingleLSMNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.net = nn.Sequential(*layers)

        def forward(self, x):
            return self.net(x)

    return SingleLSMNet()


def features(x, s_t):
    """x = S/K (float64 tensor [n]); s_t = sqrt(max(T - t*dt, 1e-6)), scalar or tensor [n]."""
    torch = _torch()
    s = s_t if torch.is_tensor(s_t) else torch.full_like(x, float(s_t))
    return torch.stack([torch.ones_like(x), x, x * x, x * x * x, torch.clamp(x - 1, min=0), s, x * s], dim=1)


def generate_paths(ctx, S, model_kw, S0, r, sigma, T, seed, stream=0, pair_offset=0):
    """Fill the torch tensor S [N+1, M] (float32, contiguous) with the HIP path kernels.  pair_offset: global index of
    the matrix's first antithetic pair (a rank's shard of a larger pricing)."""
    torch = _torch()
    N, M = S.shape[0] - 1, S.shape[1]
    assert S.is_contiguous() and S.dtype == torch.float32
    # the library call is host-synchronous on its own stream; make sure torch has no pending
    # work on the (possibly recycled) block behind S before the kernel writes into it
    torch.cuda.synchronize(S.device)
    if model_kw.get("model", "gbm") == "heston":
        hp = {k: model_kw[k] for k in ("v0", "kappa", "theta", "xi", "rho")}
        _ffi._check(ctx.lib, ctx.lib.omc_heston_paths_f32(ctx.handle, S.data_ptr(), M, M, N, S0, r, T,
                                                           hp["v0"], hp["kappa"], hp["theta"], hp["xi"],
                                                           hp["rho"], seed, stream, int(pair_offset), 0))
    else:
        _ffi._check(ctx.lib, ctx.lib.omc_gbm_paths_f32(ctx.handle, S.data_ptr(), M, M, N, S0, r, sigma,
                                                        T, seed, stream, int(pair_offset), 1))


def collect_rows(S, K, r, T, is_put, step_chunk=32):
    """Pass 1 (:482-516): x = S/K, step index and target for every ITM (t, path), t = N-1..1."""
    torch = _torch()
    N, M = S.shape[0] - 1, S.shape[1]
    dt = T / N
    payT = (K - S[N].double()).clamp_(min=0) if is_put else (S[N].double() - K).clamp_(min=0)
    xs, ts, ys = [], [], []
    for hi in range(N - 1, 0, -step_chunk):  # reference order: later steps first
        lo = max(1, hi - step_chunk + 1)
        blk = S[lo:hi + 1].flip(0).double()  # rows hi, hi-1, .., lo
        mask = ((K - blk) if is_put else (blk - K)) > 0
        ti, ji = mask.nonzero(as_tuple=True)
        t_abs = hi - ti
        xs.append(blk[mask] / K)
        ts.append(t_abs.to(torch.int32))
        ys.append(payT[ji] * torch.exp(-r * dt * (N - t_abs).double()))
    if not xs:
        return None
    return torch.cat(xs), torch.cat(ts), torch.cat(ys), payT


def normalisers(x, t, y, T, dt):
    """:550-563 in float64: population std; zero feature std -> 1; zero target std -> 1.
    The sums run in the library (omc_nn_feature_stats: mean pass, then squared deviations)."""
    torch = _torch()
    dev = x.device
    ctx = _ctx_on_torch_stream(dev.index or 0)
    torch.cuda.current_stream(dev).synchronize()
    mean, var = ctx.nn_feature_stats(x.data_ptr(), t.data_ptr(), y.data_ptr(), x.numel(), T, dt)
    fm = torch.tensor([1.0] + list(mean[:6]), dtype=torch.float64, device=dev)
    fs = torch.tensor([0.0] + list(var[:6]), dtype=torch.float64, device=dev).sqrt()
    # a constant column: exactly zero in exact arithmetic, rounding noise of the sums here
    fs = torch.where(fs <= 1e-13 * fm.abs().clamp(min=1e-300), torch.ones_like(fs), fs)
    ym = torch.tensor(float(mean[6]), dtype=torch.float64, device=dev)
    ysd = torch.tensor(math.sqrt(float(var[6])), dtype=torch.float64, device=dev)
    if not float(ysd) > 1e-13 * abs(float(ym)):
        ysd = torch.ones_like(ysd)
    return fm, fs, ym, ysd


def pick_batch(R, nn_batch=None):
    if nn_batch:
        return int(min(R, nn_batch))
    if R <= (1 << 18):
        return int(min(256, R))  # the reference's choice
    return int(min(R, 1 << max(8, math.ceil(math.log2(R / 1024)))))


def build_training_matrix(x, t, y, fm, fs, ym, ysd, T, dt, chunk=1 << 23):
    """[R, 8] float32 = 7 normalised features + normalised target (the reference's
    X_all_norm / Y_all_scaled after their .float() cast, :563-571), built once in chunks."""
    torch = _torch()
    R = x.numel()
    data = torch.empty((R, 8), dtype=torch.float32, device=x.device)
    for o in range(0, R, chunk):
        st = torch.sqrt(torch.clamp(T - t[o:o + chunk].double() * dt, min=1e-6))
        data[o:o + chunk, :7] = ((features(x[o:o + chunk], st) - fm) / fs).float()
        data[o:o + chunk, 7] = ((y[o:o + chunk] - ym) / ysd).float()
    return data


class _GraphedStep:
    """One optimizer step (forward, MSE, backward, Adam) captured in a HIP graph and replayed:
    at batch 256 an eager step is ~40 launches of microsecond kernels, i.e. pure launch latency.
    Falls back to eager execution if capture is not possible."""

    def __init__(self, net, opt, loss_fn, bs, device):
        torch = _torch()
        self.torch, self.net, self.opt, self.loss_fn = torch, net, opt, loss_fn
        self.buf = torch.zeros((bs, 8), dtype=torch.float32, device=device)
        self.loss = torch.zeros((), dtype=torch.float32, device=device)
        self.graph = None
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):  # warm-up off the capture stream, as torch requires
                for _ in range(3):
                    self._eager(self.buf)
            torch.cuda.current_stream(device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                out = self.loss_fn(net(self.buf[:, :7]), self.buf[:, 7:])
                out.backward()
                opt.step()
                self.loss.copy_(out.detach())
            self.graph = g
        except Exception:  # noqa: BLE001
            self.graph = None

    def _eager(self, batch):
        loss = self.loss_fn(self.net(batch[:, :7]), batch[:, 7:])
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, batch):
        if self.graph is not None and batch.shape[0] == self.buf.shape[0]:
            self.buf.copy_(batch)
            self.graph.replay()
            return self.loss
        return self._eager(batch)


def _linear_shape(net):
    """-> (hidden, hidden_layers) of a SingleLSMNet(7, hidden, layers), else None."""
    torch = _torch()
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    dims = [(m.in_features, m.out_features) for m in lin]
    if len(dims) < 2 or dims[0][0] != 7 or dims[-1][1] != 1:
        return None
    H = dims[0][1]
    if any(d != (H, H) for d in dims[1:-1]) or dims[-1][0] != H:
        return None
    return H, len(dims) - 1


def fused_trainer_supports(net, batch=256):
    """The hand-written trainers (omc_mlp_train_epoch) cover 64 hidden units x 2 or 3 hidden layers at
    any minibatch size (workgroup kernel with the weights in LDS, tile-per-wave kernel for small
    minibatches; BASELINE config 5 names 2 x 64), 128 units -- the reference's default width --
    (tile-per-wave kernel) and 32 units (one 32-row tile per workgroup at any minibatch size)."""
    shape = _linear_shape(net)
    if shape is None:
        return False
    lib = _ffi.load_library()
    return bool(lib.omc_mlp_train_supported(int(shape[0]), int(shape[1]), int(batch)))


def fused_apply_supports(net):
    """The pass-2 kernel (omc_lsm_apply_mlp) covers 32, 64 or 128 hidden units x 2 or 3 hidden layers,
    i.e. also the reference's own SingleLSMNet defaults (3 x 128, options_model_3.py:87)."""
    return _linear_shape(net) in ((32, 2), (32, 3), (64, 2), (64, 3), (128, 2), (128, 3))


def flatten_params(net):
    """torch SingleLSMNet -> the library's flat float32 layout (include/omc.h): W1|b1 as [H][8]
    (bias in column 7), then per further hidden layer W [H][H] and b [H], then the output weights
    [H] and bias [1]."""
    torch = _torch()
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        parts = [torch.cat([lin[0].weight, lin[0].bias[:, None]], dim=1).reshape(-1)]
        for m in lin[1:-1]:
            parts += [m.weight.reshape(-1), m.bias]
        parts += [lin[-1].weight.reshape(-1), lin[-1].bias]
        return torch.cat(parts).float().contiguous()


def unflatten_params(net, flat):
    """Inverse of flatten_params: write the flat vector back into the network's Linear layers."""
    torch = _torch()
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    H = lin[0].out_features
    with torch.no_grad():
        w1 = flat[:H * 8].view(H, 8)
        lin[0].weight.copy_(w1[:, :7])
        lin[0].bias.copy_(w1[:, 7])
        o = H * 8
        for m in lin[1:-1]:
            m.weight.copy_(flat[o:o + H * H].view(H, H))
            m.bias.copy_(flat[o + H * H:o + H * H + H])
            o += H * H + H
        lin[-1].weight.copy_(flat[o:o + H].view(1, H))
        lin[-1].bias.copy_(flat[o + H:o + H + 1])


def _dropout_of(net):
    torch = _torch()
    ps = [m.p for m in net.net if isinstance(m, torch.nn.Dropout)]
    return float(ps[0]) if ps else 0.0


class EpochControl:
    """Host-side control of the reference's training loop (options_model_3/options_model_3.py:574-613), as plain
    arithmetic on the epoch-mean losses -- no tensors, no optimizer object:

      * `scheduler.step(avg_loss)` of ReduceLROnPlateau(opt, patience=5, factor=0.5, min_lr=1e-6) with torch's other
        defaults (mode "min", threshold 1e-4 relative, cooldown 0, eps 1e-8): an epoch is "better" when
        loss < best * (1 - 1e-4); after MORE than 5 epochs in a row that are not, lr <- max(lr / 2, 1e-6) (applied only
        if it changes lr by more than 1e-8) and the count restarts;
      * the reference's own rule right after it: `avg_loss < best_loss - 1e-6` keeps the weights of that epoch and
        resets its counter, anything else counts, and the 8th in a row ends training (":607-611");
      * at the end the kept weights are restored (":613-615").

    step(avg_loss) -> (keep_weights, stop).  Pinned in tests/test_nn_epoch_control_cpu.py against torch's scheduler on
    scripted losses and against traces recorded from the reference's real runs (tests/golden/nn_epoch_trace.npz)."""

    def __init__(self, lr, patience=5, factor=0.5, min_lr=1e-6, threshold=1e-4, eps=1e-8, stop_patience=8,
                 min_delta=1e-6):
        self.lr = float(lr)
        self.patience, self.factor, self.min_lr, self.threshold, self.eps = patience, factor, min_lr, threshold, eps
        self.stop_patience, self.min_delta = stop_patience, min_delta
        self.sched_best = float("inf")
        self.num_bad = 0
        self.best_loss = float("inf")
        self.best_epoch = -1   # 0-based epoch whose weights are kept (-1: none yet)
        self.bad = 0
        self.epoch = -1

    def step(self, avg_loss):
        avg_loss = float(avg_loss)
        self.epoch += 1
        # ReduceLROnPlateau.step (mode min, relative threshold, no cooldown)
        if avg_loss < self.sched_best * (1.0 - self.threshold):
            self.sched_best = avg_loss
            self.num_bad = 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad = 0
        # the reference's early-stopping rule
        if avg_loss < self.best_loss - self.min_delta:
            self.best_loss = avg_loss
            self.best_epoch = self.epoch
            self.bad = 0
            return True, False
        self.bad += 1
        return False, self.bad >= self.stop_patience


def _epoch_key(seed, epoch):
    """The keyed permutation of an epoch (stands for the reference's DataLoader shuffle, :575)."""
    return (seed ^ (0x9E3779B97F4A7C15 * (epoch + 1))) % (1 << 64) or 1


def _train_fused(net, data, epochs, lr, bs, verbose):
    """The epoch loop of :565-613 around omc_mlp_train_epoch (hand-written MFMA forward/backward +
    Adam kernels); shuffling, the plateau scheduler and early stopping stay on the host side."""
    torch = _torch()
    dev = data.device
    R = data.shape[0]
    ctx = _ctx_on_torch_stream(dev.index or 0)
    params = flatten_params(net)
    m = torch.zeros_like(params)
    v = torch.zeros_like(params)
    ctl = EpochControl(lr)
    p_drop = _dropout_of(net)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # drawn from torch's seeded generator
    best_params, step = None, 0
    t_kernels = 0.0
    torch.cuda.current_stream(dev).synchronize()  # the training matrix is complete
    for epoch in range(epochs):
        t1 = time.perf_counter()
        # the epoch's shuffle (:575 randperm) is a keyed permutation evaluated inside the kernel
        avg, step = ctx.mlp_train_epoch(data.data_ptr(), R, bs, params.data_ptr(), m.data_ptr(), v.data_ptr(),
                                        step, ctl.lr, p_drop, seed,
                                        hidden=_linear_shape(net)[0], layers=_linear_shape(net)[1], shuffle_key=_epoch_key(seed, epoch))
        t_kernels += time.perf_counter() - t1
        keep, stop = ctl.step(avg)
        if keep:
            best_params = params.clone()
        elif stop:
            if verbose:
                print(f"Early stopping at epoch {epoch + 1}, restoring best weights")
            break
    unflatten_params(net, best_params if best_params is not None else params)
    H_, L_ = _linear_shape(net)
    kernel = {1: "mlp_train_kernel", 2: "mlp_train_tile_kernel", 3: "mlp_train_quad_kernel", 4: "mlp_train_q16_kernel"}.get(
        int(ctx.lib.omc_mlp_train_variant(int(H_), int(L_), int(bs))), "?")
    return dict(batch=bs, optimizer_steps=step, epochs_run=epoch + 1, best_loss=ctl.best_loss, best_epoch=ctl.best_epoch,
                final_lr=ctl.lr, graphed=False, trainer="hip", trainer_kernel=kernel, seconds_train_kernels=t_kernels)


def build_rows_fused(S, K, r, T, is_put):
    """Pass 1 (:482-563) by the library (omc_nn_build_rows): the [R, 8] float32 training matrix in the
    reference's row order plus the normalisers, straight from the path matrix -- no x / t / y
    temporaries.  -> (data, feat_mean, feat_std, y_mean, y_std) or None if nothing is in the money."""
    torch = _torch()
    dev = S.device
    N, M = S.shape[0] - 1, S.shape[1]
    ctx = _ctx_on_torch_stream(dev.index or 0)
    torch.cuda.current_stream(dev).synchronize()
    R = ctx.nn_build_rows(S.data_ptr(), S.stride(0), M, N, K, r, T, is_put)
    if R == 0:
        return None
    data = torch.empty((R, 8), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    _, fm, fs, ym, ysd = ctx.nn_build_rows(S.data_ptr(), S.stride(0), M, N, K, r, T, is_put, data.data_ptr(), R)
    f64 = dict(dtype=torch.float64, device=dev)
    return (data, torch.tensor(fm, **f64), torch.tensor(fs, **f64), torch.tensor(ym, **f64), torch.tensor(ysd, **f64))


def train(net, x, t, y, fm, fs, ym, ysd, T, dt, epochs, lr, nn_batch=None, verbose=False,
          use_graph=True, trainer="auto"):
    """:565-613 on the rows (x, t, y) of collect_rows: builds the training matrix, then train_on_matrix."""
    torch = _torch()
    t_m = time.perf_counter()
    data = build_training_matrix(x, t, y, fm, fs, ym, ysd, T, dt)
    torch.cuda.synchronize(x.device)
    t_m = time.perf_counter() - t_m
    return dict(train_on_matrix(net, data, epochs, lr, nn_batch, verbose, use_graph, trainer), seconds_matrix=t_m)


def train_on_matrix(net, data, epochs, lr, nn_batch=None, verbose=False, use_graph=True, trainer="auto"):
    """:565-613: Adam(lr, wd 1e-5), MSE, shuffled minibatches, ReduceLROnPlateau on the epoch-mean
    loss, early stop after 8 non-improving epochs, best-weights restore.
    trainer: "hip" = the library's fused MFMA kernels (32, 64 or 128 units x 2 or 3 hidden layers), "torch" = PyTorch-ROCm
    autograd (any SingleLSMNet shape), "auto" = hip where it applies."""
    torch = _torch()
    R = data.shape[0]
    bs = pick_batch(R, nn_batch)
    dev = data.device
    if trainer not in ("auto", "hip", "torch"):
        raise ValueError("trainer must be 'auto', 'hip' or 'torch'")
    if trainer == "hip" and not fused_trainer_supports(net, bs):
        raise ValueError("trainer='hip' covers SingleLSMNet(7, 32 | 64 | 128, 2 | 3)")
    if trainer != "torch" and fused_trainer_supports(net, bs):
        return _train_fused(net, data, epochs, lr, bs, verbose)
    if trainer == "auto":
        _warn_once("trainer", f"options_model_amd: SingleLSMNet shape {_linear_shape(net)} (hidden units, hidden layers) is not "
                              "covered by the HIP trainer kernels (32 | 64 | 128 units x 2 | 3 layers): training through PyTorch-ROCm "
                              "autograd instead (same arithmetic, several times slower); info['trainer'] == 'torch'")
    # state snapshots for the warm-up steps of the graph capture must not leak into training
    init_state = copy.deepcopy(net.state_dict())
    lr_t = torch.tensor(float(lr), dtype=torch.float32, device=dev)
    opt = torch.optim.Adam(net.parameters(), lr=lr_t, weight_decay=1e-5, capturable=True)
    loss_fn = torch.nn.MSELoss()
    net.train()
    step = _GraphedStep(net, opt, loss_fn, bs, dev) if use_graph else None
    graphed = step is not None and step.graph is not None
    if step is not None:
        # The warm-up steps of the capture moved the weights and the Adam moments.  The graph
        # holds those very tensors, so reset them IN PLACE: training starts from the initial
        # weights with a pristine optimizer, exactly as without the graph.
        net.load_state_dict(init_state)
        for st_ in opt.state.values():
            for v_ in st_.values():
                if torch.is_tensor(v_):
                    v_.zero_()
    ctl = EpochControl(lr)
    best_state, steps = None, 0
    for epoch in range(epochs):
        perm = torch.randperm(R, device=dev)
        shuf = data[perm]
        tot = torch.zeros((), dtype=torch.float64, device=dev)
        nb = 0
        for o in range(0, R, bs):
            batch = shuf[o:o + bs]
            loss = step(batch) if step is not None else None
            if loss is None:
                l_ = loss_fn(net(batch[:, :7]), batch[:, 7:])
                opt.zero_grad(set_to_none=True)
                l_.backward()
                opt.step()
                loss = l_.detach()
            tot += loss.double()
            nb += 1
        del shuf
        steps += nb
        avg = float(tot) / max(nb, 1)  # one host sync per epoch
        keep, stop = ctl.step(avg)
        lr_t.fill_(ctl.lr)  # the optimizer reads its learning rate from this tensor (capturable Adam)
        if keep:
            best_state = copy.deepcopy(net.state_dict())
        elif stop:
            if verbose:
                print(f"Early stopping at epoch {epoch + 1}, restoring best weights")
            break
    if best_state is not None:
        net.load_state_dict(best_state)
    return dict(batch=bs, optimizer_steps=steps, epochs_run=epoch + 1, best_loss=ctl.best_loss, best_epoch=ctl.best_epoch,
                final_lr=ctl.lr, graphed=graphed, trainer="torch")


def pass2(S, K, r, T, is_put, net, fm, fs, ym, ysd, dropout_on=True, path_chunk=1 << 20):
    """:615-651, dense over paths: sticky mask, strict >, mean valued at t = dt."""
    torch = _torch()
    N, M = S.shape[0] - 1, S.shape[1]
    dt = T / N
    disc = math.exp(-r * dt)
    net.train(dropout_on)  # the reference never calls .eval() on this net
    payf = (lambda s: (K - s).clamp(min=0)) if is_put else (lambda s: (s - K).clamp(min=0))
    cf = payf(S[N].double())
    exercised = torch.zeros(M, dtype=torch.bool, device=S.device)
    ym_f, ys_f = float(ym), float(ysd)
    with torch.no_grad():
        for t in range(N - 1, 0, -1):
            cf *= disc
            st = S[t].double()
            imm = payf(st)
            itm = (imm > 0) & ~exercised
            s_t = math.sqrt(max(T - t * dt, 1e-6))
            cont = torch.empty(M, dtype=torch.float64, device=S.device)
            for o in range(0, M, path_chunk):
                fb = ((features(st[o:o + path_chunk] / K, s_t) - fm) / fs).float()
                cont[o:o + path_chunk] = net(fb).squeeze(1).double() * ys_f + ym_f
            ex = itm & (imm > cont)
            cf = torch.where(ex, imm, cf)
            exercised |= ex
    return cf, exercised


def pass2_fused(S, K, r, T, is_put, net, fm, fs, ym, ysd, dropout_on=True, want_state=False, seed=None):
    """:615-651 by the library's kernel (omc_lsm_apply_mlp): the same sticky sweep, the network
    evaluated by float32 MFMA per 32-path tile, dropout bits from Philox-seeded streams."""
    torch = _torch()
    dev = S.device
    N, M = S.shape[0] - 1, S.shape[1]
    ctx = _ctx_on_torch_stream(dev.index or 0)
    H, L = _linear_shape(net)
    params = flatten_params(net)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.cuda.current_stream(dev).synchronize()
    out = ctx.lsm_apply_mlp(S.data_ptr(), S.stride(0), M, N, K, r, T, is_put, params.data_ptr(),
                            fm.cpu().numpy(), fs.cpu().numpy(), float(ym), float(ysd),
                            _dropout_of(net) if dropout_on else 0.0, seed, want_state=want_state,
                            hidden=H, layers=L)
    res = dict(price=out["price"], std=out["std"], n_exercised=out["n_exercised"], zero_prob=out["zero_prob"],
               pass2="hip")
    if want_state:
        res.update(sx=out["sx"], tex=out["tex"])
    return res


def price_with_paths(S, K, r, T, is_put, torch_seed, nn_hidden=128, nn_layers=3, nn_dropout=0.1,
                     nn_epochs=25, nn_lr=1e-3, nn_batch=None, inference_dropout=True, verbose=False,
                     trainer="auto"):
    """The whole v3 NN flow on a device path matrix S (torch float32 [N+1, M])."""
    torch = _torch()
    N, M = S.shape[0] - 1, S.shape[1]
    dt = T / N
    torch.manual_seed(int(torch_seed))  # :455
    t0 = time.perf_counter()
    net = make_net(7, nn_hidden, nn_layers, nn_dropout).to(S.device)
    if trainer != "torch":
        # pass 1 by the library: rows + normalisers straight from S
        built = build_rows_fused(S, K, r, T, is_put)
        if built is None:  # :518-519 nothing ever in the money
            payT = (K - S[N].double()).clamp_(min=0) if is_put else (S[N].double() - K).clamp_(min=0)
            cf = payT * math.exp(-r * dt * (N - 1))
            return dict(price=float(cf.mean()), R=0, n_paths=M, n_exercised=0)
        data, fm, fs, ym, ysd = built
        R = data.shape[0]
        torch.cuda.synchronize(S.device)
        t_c = t1 = time.perf_counter()
        info = dict(train_on_matrix(net, data, nn_epochs, nn_lr, nn_batch, verbose, trainer=trainer), rows="hip")
        del data
    else:
        rows = collect_rows(S, K, r, T, is_put)
        if rows is None:  # :518-519 nothing ever in the money
            payT = (K - S[N].double()).clamp_(min=0) if is_put else (S[N].double() - K).clamp_(min=0)
            cf = payT * math.exp(-r * dt * (N - 1))
            return dict(price=float(cf.mean()), R=0, n_paths=M, n_exercised=0)
        x, t, y, _ = rows
        R = int(x.numel())
        torch.cuda.synchronize(S.device)
        t_c = time.perf_counter()
        fm, fs, ym, ysd = normalisers(x, t, y, T, dt)
        torch.cuda.synchronize(S.device)
        t1 = time.perf_counter()
        info = dict(train(net, x, t, y, fm, fs, ym, ysd, T, dt, nn_epochs, nn_lr, nn_batch, verbose, trainer=trainer),
                    rows="torch")
        del x, t, y
    torch.cuda.synchronize(S.device)
    t2 = time.perf_counter()
    if trainer != "torch" and fused_apply_supports(net):
        res = pass2_fused(S, K, r, T, is_put, net, fm, fs, ym, ysd, dropout_on=inference_dropout)
    else:
        if trainer != "torch":
            _warn_once("pass2", f"options_model_amd: SingleLSMNet shape {_linear_shape(net)} is not covered by the HIP pass-2 "
                                "kernel (32 | 64 | 128 units x 2 | 3 layers): the backward sweep evaluates the network through "
                                "PyTorch-ROCm; info['pass2'] == 'torch'")
        cf, ex = pass2(S, K, r, T, is_put, net, fm, fs, ym, ysd, dropout_on=inference_dropout)
        price = float(cf.mean())
        var = float(((cf - price) ** 2).mean())
        res = dict(price=price, std=math.sqrt(var), n_exercised=int(ex.sum()),
                   zero_prob=float((cf == 0).double().mean()), pass2="torch")
    t3 = time.perf_counter()
    info.update(res, stderr=res["std"] / math.sqrt(M), R=R, n_paths=M,
                Y_mean=float(ym), Y_std=float(ysd), seconds_collect=t_c - t0, seconds_normalise=t1 - t_c,
                seconds_train=t2 - t1, seconds_pass2=t3 - t2, net=net, feat_mean=fm, feat_std=fs)
    return info


def price_curve_nn(pricer, jobs, chunk=256):
    """Many pricings of AdvancedOptionPricer(regressor='nn') -- the points of a value-vs-expiry curve
    (compute_curve_for_S0, options_model_3.py:697-713) -- with their networks TRAINED SIDE BY SIDE: training is
    97 % of such a pricing and one network at the reference's minibatch of 256 rows occupies 8 of the chip's 256 CUs,
    so every optimizer step of up to `chunk` points goes in one launch pair (omc_mlp_train_epoch_batch).  Paths, rows,
    network initialisation and pass 2 run per point as in price_two_pass_nn; each point draws from torch's generator
    exactly what its own call draws (manual_seed -> net init -> two seeds), keeps its own learning-rate schedule and
    early stopping (EpochControl) -- the result of every point equals its single call, bit for bit.
    jobs: [(S0, T, M, N, path_seed, torch_seed)] -> [result dict]."""
    torch = _torch()
    dev = torch.device("cuda", pricer.device)
    is_put = pricer.option_type == "put"
    K, r = pricer.K, pricer.r
    results = [None] * len(jobs)
    with torch.cuda.device(dev):
        ctx = _ctx_on_torch_stream(pricer.device)
        for lo in range(0, len(jobs), chunk):
            probs = []
            for idx in range(lo, min(lo + chunk, len(jobs))):
                S0, T, M, N, path_seed, torch_seed = jobs[idx]
                S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
                generate_paths(ctx, S, pricer._model_kw(), S0, r, pricer.sigma or 0.0, T, path_seed)
                torch.manual_seed(int(torch_seed))  # :455
                net = make_net(7, pricer.nn_hidden, pricer.nn_layers, pricer.nn_dropout).to(dev)
                built = build_rows_fused(S, K, r, T, is_put)
                bs = pick_batch(built[0].shape[0], None) if built is not None else 0
                H, L = _linear_shape(net)
                if built is None or not fused_trainer_supports(net, bs) or not _batch_trainer_supports(H, L, bs):
                    # nothing in the money, or a shape the side-by-side trainer does not cover: the single path
                    del S, net, built
                    out = price_two_pass_nn(pricer, S0, T, M, N, path_seed, torch_seed)
                    results[idx] = out
                    continue
                data, fm, fs, ym, ysd = built
                params = flatten_params(net)
                p = dict(idx=idx, S=S, T=T, N=N, M=M, net=net, data=data, fm=fm, fs=fs, ym=ym, ysd=ysd, bs=bs,
                         params=params, m=torch.zeros_like(params), v=torch.zeros_like(params), step=0,
                         seed=int(torch.randint(0, 2 ** 62, (1,)).item()),        # _train_fused's draw
                         seed2=int(torch.randint(0, 2 ** 62, (1,)).item()),       # pass2_fused's draw
                         ctl=EpochControl(pricer.nn_lr), best=None, done=False, epochs_run=0)
                probs.append(p)
            if not probs:
                continue
            H, L = _linear_shape(probs[0]["net"])
            p_drop = _dropout_of(probs[0]["net"])
            torch.cuda.current_stream(dev).synchronize()  # every training matrix is complete
            for epoch in range(pricer.nn_epochs):
                act = [p for p in probs if not p["done"]]
                if not act:
                    break
                outs = ctx.mlp_train_epoch_batch(
                    [dict(data_ptr=p["data"].data_ptr(), n_rows=p["data"].shape[0], batch=p["bs"],
                          params_ptr=p["params"].data_ptr(), m_ptr=p["m"].data_ptr(), v_ptr=p["v"].data_ptr(),
                          step=p["step"], lr=p["ctl"].lr, seed=p["seed"], shuffle_key=_epoch_key(p["seed"], epoch))
                     for p in act], H, L, p_drop)
                for p, (avg, step) in zip(act, outs):
                    p["step"], p["epochs_run"] = step, epoch + 1
                    keep, stop = p["ctl"].step(avg)
                    if keep:
                        p["best"] = p["params"].clone()
                    elif stop:
                        p["done"] = True
            for p in probs:
                unflatten_params(p["net"], p["best"] if p["best"] is not None else p["params"])
                res = pass2_fused(p["S"], K, r, p["T"], is_put, p["net"], p["fm"], p["fs"], p["ym"], p["ysd"],
                                  dropout_on=True, seed=p["seed2"])
                R = p["data"].shape[0]
                res.update(stderr=res["std"] / math.sqrt(p["M"]), R=R, n_paths=p["M"], Y_mean=float(p["ym"]),
                           Y_std=float(p["ysd"]), batch=p["bs"], optimizer_steps=p["step"], epochs_run=p["epochs_run"],
                           best_loss=p["ctl"].best_loss, best_epoch=p["ctl"].best_epoch, final_lr=p["ctl"].lr,
                           graphed=False, trainer="hip", rows="hip", batched_with=len(probs))
                results[p["idx"]] = res
            del probs
    return results


def _batch_trainer_supports(H, L, bs):
    """The side-by-side trainer is the one-tile-per-workgroup kernel: which (shape, minibatch) it covers."""
    return bool(_ffi.load_library().omc_mlp_train_batch_supported(int(H), int(L), int(bs)))


def price_two_pass_nn(pricer, S0, T, M, N, path_seed, torch_seed):
    """Backend of AdvancedOptionPricer(regressor='nn').price_american_enhanced_lsm."""
    torch = _torch()
    dev = torch.device("cuda", pricer.device)
    with torch.cuda.device(dev):
        ctx = _ctx_on_torch_stream(pricer.device)
        S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
        generate_paths(ctx, S, pricer._model_kw(), S0, pricer.r, pricer.sigma or 0.0, T, path_seed)
        out = price_with_paths(S, pricer.K, pricer.r, T, pricer.option_type == "put", torch_seed,
                               nn_hidden=pricer.nn_hidden, nn_layers=pricer.nn_layers,
                               nn_dropout=pricer.nn_dropout, nn_epochs=pricer.nn_epochs,
                               nn_lr=pricer.nn_lr, verbose=pricer.verbose)
    out.pop("net", None)
    return out


def price_american_option_nn(S0, K, r, sigma, T, n_paths, n_steps, model="GBM", option_type="put",
                             heston_params=None, seed=42, stream=0, device=0, nn_hidden=64,
                             nn_layers=2, nn_dropout=0.1, nn_epochs=25, nn_lr=1e-3, nn_batch=None,
                             inference_dropout=True, nn_trainer="auto", torch_seed=None):
    """Facade backend for regressor='nn' (BASELINE config 5 names a 2x64 MLP).  torch_seed: what the reference
    hands to torch.manual_seed (options_model_3.py:455); default seed + 1."""
    from .api import PriceResult, _validate
    torch = _torch()
    model_l = str(model).lower()
    _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
    M = int(n_paths) // 2 * 2
    dev = torch.device("cuda", device)
    kw = dict(model=model_l, **heston_defaults(sigma, heston_params))
    with torch.cuda.device(dev):
        ctx = _ctx_on_torch_stream(device)
        S = torch.empty((int(n_steps) + 1, M), dtype=torch.float32, device=dev)
        generate_paths(ctx, S, kw, S0, r, sigma or 0.0, T, seed, stream)
        out = price_with_paths(S, K, r, T, option_type == "put", seed + 1 if torch_seed is None else torch_seed,
                               nn_hidden, nn_layers,
                               nn_dropout, nn_epochs, nn_lr, nn_batch, inference_dropout, trainer=nn_trainer)
    return PriceResult(price=out["price"], stderr=out.get("stderr", 0.0), std=out.get("std", 0.0),
                       zero_prob=out.get("zero_prob", 0.0), n_paths=M,
                       n_exercised=out.get("n_exercised", 0), sum_nitm=out.get("R", 0), model=model_l,
                       semantics="two_pass", option_type=option_type,
                       timings_ms={k[len("seconds_"):]: 1e3 * v for k, v in out.items() if k.startswith("seconds_")},
                       info={k: out[k] for k in ("trainer", "trainer_kernel", "pass2", "rows", "batch", "epochs_run", "optimizer_steps",
                                                 "best_loss", "graphed") if k in out})
