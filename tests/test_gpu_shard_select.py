"""Unit test of the sharded trainer's epoch selection (csrc/omc_mlp.hip shard_select / gather / step_off kernels behind
omc_mlp_shard_epoch) against a numpy mirror: the device's own keyed permutation (omc_mlp_shuffle_indices, proven a
permutation in test_gpu_mlp.py) + nn_dist.locate on the same segment tables.  One process plays every rank in turn:
the ranks' selections must partition the epoch's positions, each in ascending position order, rows gathered from the
right place, dropout keys = position inside the global minibatch, step offsets = first own row of every minibatch."""
import numpy as np
import pytest

from options_model_amd import nn_dist

pytestmark = pytest.mark.gpu


def _tables(rng, world, steps, lo, hi, zero_frac):
    counts = rng.integers(lo, hi, size=(world, steps, 2)).astype(np.int64)
    counts[rng.random(counts.shape) < zero_frac] = 0  # empty segments: ranks without rows in a step / half
    return counts


@pytest.mark.parametrize("world,steps,lo,hi,zero_frac,batch", [
    (1, 5, 0, 40, 0.2, 64),          # one rank owns everything
    (2, 7, 0, 300, 0.3, 256),
    (3, 11, 0, 50, 0.5, 100),        # batch not a power of two, many empty segments
    (8, 251, 0, 30, 0.1, 4096),      # config 5's segment count (4,016 segments: two-level search)
    (4, 3, 0, 3, 0.6, 7),            # a handful of rows
    (16, 40, 100, 400, 0.0, 8192),
])
def test_epoch_selection_matches_numpy_mirror(ctx, world, steps, lo, hi, zero_frac, batch):
    import torch
    rng = np.random.default_rng(world * 1000 + steps)
    counts = _tables(rng, world, steps, lo, hi + 1, zero_frac)
    R = int(counts.sum())
    if R == 0:
        counts[0, 0, 0] = 5
        R = 5
    key = 0x1234ABCD5678 + world
    perm_d = torch.empty(R, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.mlp_shuffle_indices(R, key, perm_d.data_ptr())
    perm = perm_d.cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(R))
    nsteps = (R + batch - 1) // batch
    seen = np.zeros(R, np.int32)
    for rank in range(world):
        gstart, lstart = nn_dist.segment_tables(counts, rank)
        assert gstart[-1] == R
        n_local = int(counts[rank].sum())
        # expected: positions whose row this rank owns, ascending
        own = np.array([nn_dist.locate(gstart, lstart, int(g)) for g in perm], np.int64)
        pos = np.nonzero(own >= 0)[0]
        assert len(pos) == n_local
        # rows labelled by their local index (column 0) so that the gather can be checked
        data = torch.zeros((max(n_local, 1), 8), dtype=torch.float32, device="cuda")
        data[:, 0] = torch.arange(max(n_local, 1), dtype=torch.float32, device="cuda")
        data[:, 7] = 0.5
        out = torch.full((max(n_local, 1), 8), -1.0, dtype=torch.float32, device="cuda")
        drop = torch.full((max(n_local, 1),), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for group in (2 * world, 0):  # two-level search in LDS and the flat search: the same answer
            so = ctx.mlp_shard_epoch(data.data_ptr(), n_local, R, batch, key, gstart, lstart, out.data_ptr(),
                                     drop.data_ptr(), segs_per_step=group)
            assert so[0] == 0 and so[-1] == n_local and len(so) == nsteps + 1
            exp_so = np.searchsorted(pos, np.arange(nsteps + 1) * batch, side="left")
            assert np.array_equal(so, exp_so)
            if n_local:
                got_rows = out[:n_local, 0].cpu().numpy().astype(np.int64)
                assert np.array_equal(got_rows, own[pos])                       # the right rows, in position order
                assert np.all(out[:n_local, 7].cpu().numpy() == 0.5)            # whole 32-byte rows moved
                assert np.array_equal(drop[:n_local].cpu().numpy().astype(np.int64), pos % batch)
        seen[pos] += 1
    assert np.all(seen == 1)  # the ranks' selections partition the epoch


def test_bad_tables_are_refused_before_any_kernel_runs(ctx):
    import torch
    data = torch.zeros((10, 8), dtype=torch.float32, device="cuda")
    out = torch.zeros_like(data)
    drop = torch.zeros(10, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = np.array([0, 4, 10], np.int64)
    with pytest.raises(ValueError, match="tile"):          # own segments must tile [0, n_rows_local)
        ctx.mlp_shard_epoch(data.data_ptr(), 10, 10, 4, 7, g, np.array([1, 5], np.int64), out.data_ptr(), drop.data_ptr())
    with pytest.raises(ValueError, match="add up"):
        ctx.mlp_shard_epoch(data.data_ptr(), 10, 10, 4, 7, g, np.array([0, -1], np.int64), out.data_ptr(), drop.data_ptr())
    with pytest.raises(ValueError, match="cover"):
        ctx.mlp_shard_epoch(data.data_ptr(), 10, 12, 4, 7, g, np.array([0, 4], np.int64), out.data_ptr(), drop.data_ptr())
    with pytest.raises(ValueError, match="ascending"):
        ctx.mlp_shard_epoch(data.data_ptr(), 10, 10, 4, 7, np.array([0, 12, 10], np.int64), np.array([0, -1], np.int64),
                            out.data_ptr(), drop.data_ptr())
