"""The NN continuation-value regressor (BASELINE config 5; options_model_3.py:542-651) sharded over the ranks of a job.

The reference trains ONE SingleLSMNet on the pass-1 rows of ALL paths (:565-613) and has no distributed code.  Here the
paths shard by antithetic pair (dist.shard: the Philox counter carries the global pair index), every rank builds the
rows of its own paths, and the job trains the network the single GPU would train:

  statistics  row count, sums and squared deviations from the global means all-reduced (3 x 8 doubles) inside
              omc_nn_build_rows -> the normalisers of :550-563 over the union of the shards;
  minibatches epoch position i trains global row perm(i) -- the single-GPU trainer's keyed permutation over the job's
              rows in the reference's order (step N-1 .. 1, global column ascending).  That order is a concatenation of
              (step, half, rank) segments (`segment_tables`); each rank keeps the positions whose row it owns
              (omc_mlp_shard_epoch), so minibatch k of the job is exactly minibatch k of the single-GPU run, split;
  gradients   forward / backward over the rank's part of a minibatch scaled by the GLOBAL minibatch size; the gradient
              sums + loss sum (parameter count + 1 doubles, 38 KB for 2 x 64) all-reduced per optimizer step on the
              context's stream (omc_mlp_train_epoch_sharded); every rank applies the same Adam step.  Dropout masks are
              keyed by the row's position in the global minibatch, so they are those of the single-GPU run;
  pass 2      local (omc_lsm_apply_mlp_shard, dropout keyed by the global column); 6 sums all-reduced (dist.merge).

Identical seeds -> identical initial weights, identical all-reduced updates -> identical weights on every rank, equal to
the single-GPU run's up to float32 summation order (the per-rank partial sums are added in another association).
"""
from __future__ import annotations

import math
import time

import numpy as np

from . import dist


def segment_tables(counts_all, rank: int):
    """counts_all: int64 [world][n_steps - 1][2] -- per rank, per step (index n_steps - 1 - t, i.e. in the reference's
    row order: later steps first) the in-the-money paths among the rank's first-partner columns [0, P_local) and its
    second-partner columns [P_local, 2 P_local) (Context.nn_half_counts).

    The unsharded matrix holds all ranks' first partners (rank order), then all second partners (options_model_3.py:476
    with the pairs of rank r at [r P_local, (r + 1) P_local)), so its rows in the reference's order -- step by step,
    column ascending -- are the segments (step, half, rank) in exactly that nesting.  A rank's OWN rows (same order on
    its own matrix) are (step, half).
    -> gstart int64 [nseg + 1] (global index of each segment's first row; gstart[-1] = rows of the job),
       lstart int64 [nseg]     (first own row of the segment, -1 = another rank's)."""
    c = np.ascontiguousarray(counts_all, np.int64)
    W, S, two = c.shape
    assert two == 2 and 0 <= rank < W
    seg = c.transpose(1, 2, 0).reshape(-1)                      # [step][half][rank]
    gstart = np.concatenate([[0], np.cumsum(seg)]).astype(np.int64)
    own = c[rank].reshape(-1)                                    # [step][half]
    own_start = (np.cumsum(own) - own).reshape(S, 2)
    lstart = np.full((S, 2, W), -1, np.int64)
    lstart[:, :, rank] = own_start
    return gstart, lstart.reshape(-1)


def locate(gstart, lstart, g):
    """Host mirror of the kernel's lookup (csrc/omc_mlp.hip shard_locate): own row of global row g, or -1."""
    s = int(np.searchsorted(gstart, g, side="right")) - 1  # the LAST segment starting at or before g: the non-empty one
    return -1 if lstart[s] < 0 else int(lstart[s] + (g - gstart[s]))


def gather_counts(ctx, counts_local, rank: int, world: int):
    """All ranks' nn_half_counts tables through the communicator (a zero-padded table, summed): [world][S][2] int64."""
    S = counts_local.shape[0]
    table = np.zeros((world, S, 2), np.float64)  # row counts are far below 2**53: exact in float64
    table[rank] = counts_local
    flat = table.reshape(-1)
    for o in range(0, flat.size, 4096):  # omc_comm_allreduce_f64 takes at most 4096 doubles
        flat[o:o + 4096] = ctx.comm_allreduce(flat[o:o + 4096])
    return np.rint(flat).astype(np.int64).reshape(world, S, 2)


def price_american_option_nn_sharded(sp, S0, K, r, sigma, T, n_paths, n_steps, model="GBM", option_type="put",
                                     heston_params=None, seed=42, stream=0, nn_hidden=64, nn_layers=2, nn_dropout=0.1,
                                     nn_epochs=25, nn_lr=1e-3, nn_batch=None, inference_dropout=True, torch_seed=None,
                                     verbose=False):
    """Backend of price_american_option(regressor="nn", n_gpus=N) on ONE rank of the job (sp: dist.RcclPricer or any
    pricer whose context sums over the ranks).  Every rank returns the same global PriceResult."""
    from . import nn_regressor as nr
    from .api import PriceResult, _validate, heston_defaults
    torch = nr._torch()
    ctx, rank, world = sp.ctx, sp.rank, sp.world
    if ctx.comm_info()[1] != world:
        raise RuntimeError("the sharded NN regressor needs the library's own communicator on the rank's context "
                           "(dist.RcclPricer); the torch.distributed hook transport does not carry its host-side sums")
    model_l = str(model).lower()
    _validate(S0, K, T, r, sigma, n_paths, n_steps, option_type, need_sigma=(model_l == "gbm"))
    M = int(n_paths) // 2 * 2
    N = int(n_steps)
    is_put = option_type == "put"
    n_local, pair_off = dist.shard(M, world, rank)
    dev = torch.device("cuda", ctx.device)
    kw = dict(model=model_l, **heston_defaults(sigma, heston_params))
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
        S = torch.empty((N + 1, n_local), dtype=torch.float32, device=dev)
        nr.generate_paths(ctx, S, kw, S0, r, sigma or 0.0, T, seed, stream, pair_offset=pair_off)
        torch.manual_seed(int(seed + 1 if torch_seed is None else torch_seed))  # :455 -- the same on every rank
        net = nr.make_net(7, nn_hidden, nn_layers, nn_dropout).to(dev)
        if not nr.fused_apply_supports(net):
            raise ValueError("the sharded NN regressor covers SingleLSMNet(7, 32 | 64 | 128, 2 | 3)")
        H, L = nr._linear_shape(net)
        # pass 1: own rows, the job's normalisers
        R_local = ctx.nn_build_rows(S.data_ptr(), S.stride(0), n_local, N, K, r, T, is_put)
        data = torch.empty((max(R_local, 1), 8), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        _, fm, fs, ym, ysd = ctx.nn_build_rows(S.data_ptr(), S.stride(0), n_local, N, K, r, T, is_put, data.data_ptr(),
                                               max(R_local, 1))
        counts = gather_counts(ctx, ctx.nn_half_counts(S.data_ptr(), S.stride(0), n_local, N, K, is_put), rank, world)
        if int(counts[rank].sum()) != R_local:
            raise RuntimeError("row counts of the two pass-1 kernels disagree")
        gstart, lstart = segment_tables(counts, rank)
        R = int(gstart[-1])
        t1 = time.perf_counter()
        info = dict(trainer="hip", pass2="hip", rows="hip", n_gpus=world, rank=rank, transport=sp.transport)
        if R == 0:  # :518-519 nothing ever in the money, on any rank
            payT = (K - S[N].double()).clamp_(min=0) if is_put else (S[N].double() - K).clamp_(min=0)
            cf = payT * math.exp(-r * (T / N) * (N - 1))
            loc = dict(sum=float(cf.sum()), sumsq=float((cf * cf).sum()), n_paths=n_local, n_exercised=0,
                       n_zero=int((cf == 0).sum()), sum_nitm=0)
            out = dist.merge(loc, ctx.comm_allreduce)
        else:
            bs = nr.pick_batch(R, nn_batch)
            params = nr.flatten_params(net)
            m, v = torch.zeros_like(params), torch.zeros_like(params)
            data_epoch = torch.empty_like(data)
            drop_pos = torch.empty(max(R_local, 1), dtype=torch.int32, device=dev)
            ctl = nr.EpochControl(nn_lr)
            p_drop = nr._dropout_of(net)
            tseed = int(torch.randint(0, 2 ** 62, (1,)).item())  # the same draw on every rank
            best, step = None, 0
            torch.cuda.synchronize(dev)
            t_sel = t_trn = 0.0
            for epoch in range(int(nn_epochs)):
                ta = time.perf_counter()
                so = ctx.mlp_shard_epoch(data.data_ptr(), R_local, R, bs, nr._epoch_key(tseed, epoch), gstart, lstart,
                                         data_epoch.data_ptr(), drop_pos.data_ptr(), segs_per_step=2 * world)
                tb = time.perf_counter()
                avg, step = ctx.mlp_train_epoch_sharded(data_epoch.data_ptr(), R_local, R, bs, params.data_ptr(),
                                                        m.data_ptr(), v.data_ptr(), step, ctl.lr, p_drop, tseed, so,
                                                        drop_pos.data_ptr(), hidden=H, layers=L)
                t_sel += tb - ta
                t_trn += time.perf_counter() - tb
                keep, stop = ctl.step(avg)  # the job's mean loss: the same decision on every rank
                if keep:
                    best = params.clone()
                elif stop:
                    if verbose:
                        print(f"Early stopping at epoch {epoch + 1}, restoring best weights")
                    break
            if best is not None:
                params = best
            info.update(batch=bs, optimizer_steps=step, epochs_run=epoch + 1, best_loss=ctl.best_loss,
                        best_epoch=ctl.best_epoch, graphed=False, seconds_select=t_sel, seconds_train_epochs=t_trn,
                        rows_local=R_local)
            del data, data_epoch
            t2 = time.perf_counter()
            p2seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            torch.cuda.synchronize(dev)
            loc = ctx.lsm_apply_mlp(S.data_ptr(), S.stride(0), n_local, N, K, r, T, is_put, params.data_ptr(), fm, fs,
                                    ym, ysd, p_drop if inference_dropout else 0.0, p2seed, hidden=H, layers=L,
                                    col_bases=(pair_off, M // 2 + pair_off))
            loc = dict(loc, n_paths=n_local, sum_nitm=R_local)
            out = dist.merge(loc, ctx.comm_allreduce)
            info.update(params_checksum=float(params.double().sum()), Y_mean=ym, Y_std=ysd)
            info["seconds_train"] = t2 - t1
    return PriceResult(price=out["price"], stderr=out["stderr"], std=out["std"], zero_prob=out["zero_prob"], n_paths=M,
                       n_exercised=out["n_exercised"], sum_nitm=out["sum_nitm"], model=model_l, semantics="two_pass",
                       option_type=option_type, timings_ms=dict(pass1=1e3 * (t1 - t0), total=1e3 * (time.perf_counter() - t0)),
                       info=info)
