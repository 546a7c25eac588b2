"""CPU suite, part 3: the N>1 path with world_size 2 over gloo.

The sharding/merge host logic is the same code the GPU ranks run (options_model_amd/dist.py);
the per-shard engine here is the CPU oracle, so the test proves the algebra: shards built from
GLOBAL pair indices + one all-reduce of the moment table + one of the sums == the unsharded
pricing, to float64 round-off.
"""
import os
import socket

import numpy as np
import pytest

from options_model_amd import dist as omc_dist

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0
M_GLOBAL, N = 4096, 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, is_put, out_dir):
    import torch
    import torch.distributed as td

    from oracle import cpu as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_local, off = omc_dist.shard(M_GLOBAL, world, rank)
        S = orc.gbm_paths(n_local, N, 100.0, R, SIG, T, 42, 0, off)
        # regression moments: ONE all-reduce of the whole [N+1][8] table
        m = torch.from_numpy(orc.lsm_pass1_moments(S, K, R, T, is_put))
        td.all_reduce(m)
        betas4 = orc.solve_poly2(m.numpy())
        loc = orc.lsm_apply_frozen(S, K, R, T, is_put, betas4[:, :3], betas4[:, 3].astype(np.int64))
        loc["n_paths"] = n_local
        loc["sum_nitm"] = 0

        def all_reduce_sum(vals):
            t = torch.tensor(vals, dtype=torch.float64)
            td.all_reduce(t)
            return t.tolist()

        out = omc_dist.merge(loc, all_reduce_sum)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"),
                np.array([out["price"], out["sumsq"], out["n_paths"], out["n_exercised"], out["n_zero"]]))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("is_put", [True, False])
def test_two_rank_sharded_two_pass_equals_single(tmp_path, is_put):
    import torch.multiprocessing as mp

    from oracle import cpu as orc

    port = _free_port()
    mp.spawn(_worker, args=(2, port, is_put, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)  # every rank ends with the same global answer
    # single process, all paths: pair p of shard k is global pair k*P_local + p, and the
    # column order inside a shard does not matter for any sum
    full = orc.lsm_poly(orc.gbm_paths(M_GLOBAL, N, 100.0, R, SIG, T, 42), K, R, T, is_put, "two_pass")
    assert r0[0] == pytest.approx(full["price"], rel=1e-12)
    assert r0[1] == pytest.approx(full["sumsq"], rel=1e-12)
    assert (int(r0[2]), int(r0[3]), int(r0[4])) == (M_GLOBAL, full["n_exercised"], full["n_zero"])


def test_shard_layout():
    assert omc_dist.shard(64_000_000, 8, 3) == (8_000_000, 12_000_000)
    assert omc_dist.shard(1000, 2, 1, antithetic=False) == (500, 500)
    parts = [omc_dist.shard(4096, 4, r) for r in range(4)]
    assert [p[1] for p in parts] == [0, 512, 1024, 1536] and all(p[0] == 1024 for p in parts)
    with pytest.raises(ValueError):
        omc_dist.shard(1001, 2, 0)
    with pytest.raises(ValueError):
        omc_dist.shard(4098, 4, 0)


def test_merge_statistics():
    loc = dict(sum=10.0, sumsq=30.0, n_paths=4, n_exercised=1, n_zero=2, sum_nitm=7)
    out = omc_dist.merge(loc, lambda v: [2 * x for x in v])  # two identical shards
    assert out["price"] == 2.5 and out["n_paths"] == 8 and out["n_exercised"] == 2
    assert out["std"] == pytest.approx(np.sqrt(60 / 8 - 2.5**2))
    assert out["zero_prob"] == 0.5 and out["sum_nitm"] == 14


# ------------------------------------------------------------------ RCCL unique-id rendezvous (no torch)
def _rdzv_worker(rank, tag, q):
    from options_model_amd import rendezvous
    payload, path = rendezvous.exchange(rank, lambda: bytes(range(128)), 128, tag, timeout_s=30.0)
    q.put((rank, payload, path))


def test_unique_id_rendezvous_two_processes(tmp_path, monkeypatch):
    """Rank 0 publishes the 128-byte id, rank 1 (started FIRST, so it has to wait) receives exactly it;
    the file goes away when rank 0 retires it.  This is the gloo-free exchange bench.py's native RCCL
    transport uses (options_model_amd/rendezvous.py)."""
    import multiprocessing as mp

    from options_model_amd import rendezvous
    monkeypatch.setenv("OMC_RDZV_DIR", str(tmp_path))
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    tag = f"t{os.getpid()}"
    p1 = ctxm.Process(target=_rdzv_worker, args=(1, tag, q))
    p1.start()
    p0 = ctxm.Process(target=_rdzv_worker, args=(0, tag, q))
    p0.start()
    got = dict()
    for _ in range(2):
        r, payload, path = q.get(timeout=60)
        got[r] = (payload, path)
    p0.join(30)
    p1.join(30)
    assert got[0][0] == got[1][0] == bytes(range(128))
    assert got[1][1] is None and os.path.exists(got[0][1])
    rendezvous.retire(got[0][1])
    assert not os.path.exists(got[0][1])
    rendezvous.retire(got[0][1])  # idempotent


def test_rendezvous_times_out_instead_of_hanging(tmp_path, monkeypatch):
    from options_model_amd import rendezvous
    monkeypatch.setenv("OMC_RDZV_DIR", str(tmp_path))
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag="nobody", timeout_s=0.2)
    # a half-written file (wrong size) is not accepted either
    open(os.path.join(str(tmp_path), "omc_rccl_uid_short"), "wb").write(b"x" * 10)
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag="short", timeout_s=0.2)


# ------------------------------------------------------------------ bench.py --gpus N launcher
def test_bench_launcher_starts_n_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher around it must start 2 rank processes itself (round 1
    ignored --gpus).  In this container there is no GPU, so both ranks must refuse to run -- and the
    parent must report that with a non-zero exit code, not print a 1-rank line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: no device for this check
    env["ROCR_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0
    assert out.stdout.strip() == ""  # no JSON line from a failed job
    assert "rank" in out.stderr and "stopping the other ranks" in out.stderr
    assert out.stderr.count("needs a GPU") >= 1


def test_bench_world_size_must_match_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=120, cwd=root)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
