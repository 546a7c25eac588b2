"""CPU suite: the algebra of the NN regressor sharded over ranks (options_model_amd/nn_dist.py),
world_size 2 over gloo with the oracle as the per-shard engine -- as tests/test_dist_cpu.py does for the polynomial flow.

  * segment tables: the (step, half, rank) segments rebuild the UNSHARDED run's row order from the shards' own rows;
  * statistics: sums, then squared deviations from the all-reduced means, all-reduced == oracle.normalisers of the whole
    problem (options_model_3.py:550-563);
  * training: each rank takes ITS rows of every global minibatch (the same permutation on all ranks), scales by the
    global minibatch size, all-reduces the gradient sums, applies the same Adam step (float32 torch on the CPU)
    == the unsharded training to float32 summation order, and bit-identical weights on both ranks.
The GPU twin (kernels + native communicator through the stand-in) is tests/test_gpu_nn_dist.py."""
import os
import socket

import numpy as np
import pytest

from options_model_amd import dist as omc_dist
from options_model_amd import nn_dist
from oracle import reference_flow as rf

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0
M_GLOBAL, N = 2048, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows_of(S):
    """Pass 1 of options_model_3.py:482-516 on a [N+1][M] matrix in the reference's order: list of (t, path j, x, y)."""
    got = {}
    rf.lsm_two_pass(S, K, R, T, True, lambda rows: got.setdefault("rows", rows), lambda m, t, s: None)
    out = []
    disc = np.exp(-R * T / N)
    payT = rf.payoff(S[-1], K, True)
    for t in range(N - 1, 0, -1):
        itm = np.where(rf.payoff(S[t], K, True) > 0)[0]
        for j in itm:
            out.append((t, int(j), S[t, j], payT[j] * disc ** (N - t)))
    assert len(out) == sum(len(s) for _, s, _ in got["rows"])
    return got["rows"], out


def _half_counts(S):
    P = S.shape[1] // 2
    c = np.zeros((N - 1, 2), np.int64)
    for t in range(1, N):
        itm = rf.payoff(S[t], K, True) > 0
        c[N - 1 - t] = (itm[:P].sum(), itm[P:].sum())
    return c


def _global_and_shards(world):
    from oracle import cpu as orc
    full = orc.gbm_paths(M_GLOBAL, N, 100.0, R, SIG, T, 42).astype(np.float64)
    shards = []
    for r in range(world):
        n_local, off = omc_dist.shard(M_GLOBAL, world, r)
        shards.append(orc.gbm_paths(n_local, N, 100.0, R, SIG, T, 42, 0, off).astype(np.float64))
    return full, shards


@pytest.mark.parametrize("world", [2, 4])
def test_segment_tables_rebuild_the_unsharded_row_order(world):
    full, shards = _global_and_shards(world)
    P, Pl = M_GLOBAL // 2, M_GLOBAL // 2 // world
    # the unsharded matrix IS the shards' halves side by side (options_model_3.py:476 with global pair indices)
    for r, Sr in enumerate(shards):
        assert np.array_equal(Sr[:, :Pl], full[:, r * Pl:(r + 1) * Pl]) and np.array_equal(Sr[:, Pl:], full[:, P + r * Pl:P + (r + 1) * Pl])
    _, rows_g = _rows_of(full)
    rows_r = [_rows_of(Sr)[1] for Sr in shards]
    counts = np.stack([_half_counts(Sr) for Sr in shards])
    total = 0
    for r in range(world):
        gstart, lstart = nn_dist.segment_tables(counts, r)
        assert gstart[-1] == len(rows_g) and gstart[0] == 0
        own = 0
        for g in range(len(rows_g)):
            l_ = nn_dist.locate(gstart, lstart, g)
            if l_ < 0:
                continue
            own += 1
            t, j, x, y = rows_g[g]
            tl, jl, xl, yl = rows_r[r][l_]
            assert (t, x, y) == (tl, xl, yl)  # the same (step, spot, target): the same training row
            jg = r * Pl + jl if jl < Pl else P + r * Pl + (jl - Pl)
            assert jg == j
        assert own == len(rows_r[r])
        total += own
    assert total == len(rows_g)  # every global row has exactly one owner


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as td

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        _, shards = _global_and_shards(world)
        S = shards[rank]
        rows, flat = _rows_of(S)
        dt = T / N

        def allsum(a):
            t_ = torch.from_numpy(np.ascontiguousarray(a, np.float64))
            td.all_reduce(t_)
            return t_.numpy()

        # ---- statistics (omc_nn_build_rows on a distributed context): count + sums, then squared deviations
        X = np.vstack([rf.regression_features(s, K, T, t * dt) for t, s, _ in rows])
        Y = np.concatenate([y for _, _, y in rows])
        tot = allsum(np.concatenate([X.sum(0), [Y.sum(), float(len(Y))]]))
        Rg = tot[-1]
        fm, ym = tot[:7] / Rg, tot[7] / Rg
        dev = allsum(np.concatenate([((X - fm) ** 2).sum(0), [((Y - ym) ** 2).sum()]]))
        fs, ysd = np.sqrt(dev[:7] / Rg), np.sqrt(dev[7] / Rg)
        fs[fs <= 1e-13 * np.abs(fm)] = 1.0
        data = np.concatenate([(X - fm) / fs, ((Y - ym) / ysd)[:, None]], axis=1).astype(np.float32)

        # ---- segment tables from the all-gathered half counts
        table = np.zeros((world, N - 1, 2))
        table[rank] = _half_counts(S)
        counts = np.rint(allsum(table)).astype(np.int64)
        gstart, lstart = nn_dist.segment_tables(counts, rank)
        Rglob = int(gstart[-1])
        assert Rglob == int(Rg)

        # ---- training: 3 epochs of minibatches over a permutation every rank draws alike
        torch.manual_seed(7)
        net = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                  torch.nn.Linear(16, 1))
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
        B = 256
        dtl = torch.from_numpy(data)
        losses = []
        for epoch in range(3):
            perm = np.random.default_rng(100 + epoch).permutation(Rglob)
            for o in range(0, Rglob, B):
                pos = perm[o:o + B]
                mine = [nn_dist.locate(gstart, lstart, int(g)) for g in pos]
                mine = torch.tensor([l_ for l_ in mine if l_ >= 0], dtype=torch.long)
                opt.zero_grad(set_to_none=True)
                b = dtl[mine]
                sq = ((net(b[:, :7]) - b[:, 7:]) ** 2).sum() if len(mine) else sum(p.sum() * 0 for p in net.parameters())
                (sq / len(pos)).backward()  # scaled by the GLOBAL minibatch size
                flatg = torch.cat([p.grad.reshape(-1) for p in net.parameters()] + [sq.detach().reshape(1)]).double()
                td.all_reduce(flatg)  # the gradient sums + loss sum of a step: one all-reduce
                o2 = 0
                for p in net.parameters():
                    p.grad.copy_(flatg[o2:o2 + p.numel()].reshape(p.shape).float())
                    o2 += p.numel()
                opt.step()
                losses.append(float(flatg[-1]) / len(pos))
        w = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), fm=fm, fs=fs, ym=ym, ysd=ysd, w=w, losses=np.array(losses), R=Rglob)
    finally:
        td.destroy_process_group()


def test_two_rank_sharded_nn_training_equals_unsharded(tmp_path):
    import torch
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["w"], r1["w"])  # identical weights on every rank, bit for bit
    full, _ = _global_and_shards(2)
    rows, flat = _rows_of(full)
    dt = T / N
    X_all, Y_all, ofm, ofs, oym, oys = rf.normalisers(rows, K, T, dt)
    assert int(r0["R"]) == X_all.shape[0]
    assert np.allclose(r0["fm"], ofm, rtol=1e-12) and np.allclose(r0["fs"], ofs, rtol=1e-10)
    assert r0["ym"] == pytest.approx(oym, rel=1e-12) and r0["ysd"] == pytest.approx(oys, rel=1e-12)
    # the unsharded training: the same permutation, the same minibatches, ordinary mean-loss backward
    data = np.concatenate([(X_all - ofm) / ofs, (Y_all - oym) / oys], axis=1).astype(np.float32)
    torch.manual_seed(7)
    torch.set_num_threads(1)
    net = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                              torch.nn.Linear(16, 1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)
    dtl = torch.from_numpy(data)
    losses = []
    for epoch in range(3):
        perm = np.random.default_rng(100 + epoch).permutation(len(dtl))
        for o in range(0, len(dtl), 256):
            b = dtl[torch.from_numpy(perm[o:o + 256])]
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(net(b[:, :7]), b[:, 7:])
            loss.backward()
            opt.step()
            losses.append(float(loss))
    w = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    # float32 sums in another association (per-rank partial sums added in float64): a few ulps per step, over
    # ~200 Adam steps whose update is g / (|g| + eps) -- a sign-sensitive map at g ~ 0
    assert np.allclose(r0["losses"], np.array(losses), rtol=2e-4, atol=1e-6)
    assert np.abs(r0["w"] - w).max() <= 2e-3 * np.abs(w).max()
