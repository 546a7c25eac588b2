"""CPU suite, part 3: the N>1 path with world_size 2 over gloo.

The sharding/merge host logic is the same code the GPU ranks run (options_model_amd/dist.py);
the per-shard engine here is the CPU oracle, so the test proves the algebra: shards built from
GLOBAL pair indices + one all-reduce of the moment table + one of the sums == the unsharded
pricing, to float64 round-off.
"""
import os
import socket
import sys

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


# ------------------------------------------------------------------ round 3: hardened rendezvous + launcher deadline
def _agree_worker(rank, world, ok, tag, q, d):
    os.environ["OMC_RDZV_DIR"] = d
    from options_model_amd import rendezvous
    try:
        q.put((rank, rendezvous.agree(rank, world, ok, "phase", tag, timeout_s=20.0)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("votes,expect", [((True, True, True), True), ((True, False, True), False)])
def test_ranks_agree_collectively(tmp_path, votes, expect):
    """Every rank gets the same verdict: True only if ALL voted True (the transport decision of dist.RcclPricer)."""
    import multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    tag = f"agree{os.getpid()}"
    ps = [ctxm.Process(target=_agree_worker, args=(r, len(votes), v, tag, q, str(tmp_path))) for r, v in enumerate(votes)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    assert got == {r: expect for r in range(len(votes))}
    assert len([f for f in os.listdir(tmp_path) if tag in f]) <= len(votes)  # at most late votes stay behind


def test_agree_names_the_rank_that_never_voted(tmp_path, monkeypatch):
    from options_model_amd import rendezvous
    monkeypatch.setenv("OMC_RDZV_DIR", str(tmp_path))
    with pytest.raises(TimeoutError, match=r"rank\(s\) \[1\]"):
        rendezvous.agree(0, 2, True, "lonely", tag=f"l{os.getpid()}", timeout_s=0.3)


def test_rendezvous_files_are_private_fresh_and_never_followed(tmp_path, monkeypatch):
    """0600, O_EXCL|O_NOFOLLOW, stale files of an earlier run under the same name are not accepted,
    and rank 0 replaces whatever sits under its name."""
    import stat
    import struct

    from options_model_amd import rendezvous
    monkeypatch.setenv("OMC_RDZV_DIR", str(tmp_path))
    tag = "sec"
    path = rendezvous.publish(b"x" * 128, tag)
    assert stat.S_IMODE(os.stat(path).st_mode) == 0o600
    assert rendezvous.fetch(128, tag, timeout_s=1.0) == b"x" * 128
    # a frame written long before this process started (a killed earlier run) is ignored
    raw = open(path, "rb").read()
    old = raw[:8] + struct.pack("<d", 1.0e9) + raw[16:]
    os.unlink(path)
    open(path, "wb").write(old)
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag, timeout_s=0.2)
    # a bare 128-byte file (round 2's format) or another tag's frame is not accepted either
    open(path, "wb").write(b"y" * 128)
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag, timeout_s=0.2)
    other = rendezvous.publish(b"z" * 128, "other")
    os.replace(other, path)
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag, timeout_s=0.2)
    # a symlink planted under the name: readers do not follow it, rank 0 replaces it without writing through it
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.unlink(path)
    os.symlink(victim, path)
    with pytest.raises(TimeoutError):
        rendezvous.fetch(128, tag, timeout_s=0.2)
    path2 = rendezvous.publish(b"w" * 128, tag)
    assert path2 == path and not os.path.islink(path) and victim.read_bytes() == b"precious"
    assert rendezvous.fetch(128, tag, timeout_s=1.0) == b"w" * 128


def test_an_explicit_tag_carries_the_nonce_too(tmp_path, monkeypatch):
    """under a nonce the staleness test is skipped -- sound only if the FILE NAME holds the nonce.  Explicit
    tags (RcclPricer(tag=...), enable_p2p(tag=...)) now get it appended: a leftover frame of an earlier launch under the
    same explicit tag has another name and is never read."""
    from options_model_amd import rendezvous as rz
    monkeypatch.setenv("OMC_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("OMC_RDZV_NONCE", "launchA")
    rz.publish(b"A" * 16, tag="mytag")
    assert rz._name("mytag") == "omc_rccl_uid_mytag_launchA"
    assert rz.fetch(16, tag="mytag", timeout_s=2) == b"A" * 16
    monkeypatch.setenv("OMC_RDZV_NONCE", "launchB")  # the next launch, same explicit tag: the old frame is invisible
    assert rz._name("mytag") == "omc_rccl_uid_mytag_launchB"
    with pytest.raises(TimeoutError):
        rz.fetch(16, tag="mytag", timeout_s=0.5)
    assert rz._name(rz.default_tag() + "_p2p0").count("launchB") == 1  # (a tag built from the default is not doubled)
    monkeypatch.delenv("OMC_RDZV_NONCE")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    assert rz._name("mytag") == "omc_rccl_uid_mytag"


def test_tag_carries_the_launch_nonce(monkeypatch):
    from options_model_amd import rendezvous
    monkeypatch.setenv("MASTER_PORT", "12345")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    monkeypatch.setenv("OMC_RDZV_NONCE", "abc123")
    assert rendezvous.default_tag() == f"12345_{os.getppid()}_abc123"
    monkeypatch.delenv("OMC_RDZV_NONCE")
    assert rendezvous.default_tag() == f"12345_{os.getppid()}"


def test_launcher_deadline_ends_ranks_that_never_finish():
    """bench.launch_ranks: ranks that hang (here: sleep) are terminated at --rank-timeout and the job exits 124."""
    import importlib.util
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, 1.0, argv=[sys.executable, "-c", "import time; time.sleep(120)"])
    assert rc == 124 and time.monotonic() - t0 < 30
    # a rank that fails takes the others down at once, with its own exit code
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, 60.0, argv=[sys.executable, "-c",
                                           "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(120)"])
    assert rc == 7 and time.monotonic() - t0 < 30


def test_watchdog_ends_a_stuck_process():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from options_model_amd.dist import Watchdog\n"
            "w = Watchdog(0.5, 'test section'); time.sleep(30)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3 and "test section did not finish within" in out.stderr
    code = code.replace("time.sleep(30)", "w.cancel(); time.sleep(1.0)")
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60).returncode == 0


# ---- the N > 1 bench line checks itself: N ranks on N distinct cards ---------------------------
def test_pci_bus_id_round_trip():
    for s_ in ("0000:c1:00.0", "0002:05:1f.7", "ffff:ff:ff.f"):
        assert omc_dist.number_to_pci(omc_dist.pci_to_number(s_)) == s_
    assert omc_dist.pci_to_number("garbage") == -1.0 and omc_dist.number_to_pci(-1.0) == "unknown"


def _table(entries, world):
    """What a SUM all-reduce of the ranks' vectors returns when rank r contributes entries[r] (None = never shows up)."""
    vec = [0.0] * (omc_dist.RANK_ROW * world)
    for e in entries:
        if e is None:
            continue
        r, dev, pci, ms = e
        got = omc_dist.gather_rank_table(lambda v: v, r, world, dev, pci, ms)
        flat = [x for row in got for x in row]
        vec = [a + b for a, b in zip(vec, flat)]
    return [vec[omc_dist.RANK_ROW * r:omc_dist.RANK_ROW * (r + 1)] for r in range(world)]


def test_rank_table_accepts_n_ranks_on_n_cards_and_names_the_slowest():
    rows = _table([(r, r, f"0000:{0x10 + r:02x}:00.0", 4.0 + 0.1 * r) for r in range(8)], 8)
    t = omc_dist.check_rank_table(rows, 8)
    assert [x["rank"] for x in t] == list(range(8)) and t[3]["pci_bus_id"] == "0000:13:00.0" and t[3]["device"] == 3
    assert max(t, key=lambda x: x["ms_per_step"])["rank"] == 7
    assert len({x["pci_bus_id"] for x in t}) == 8


def test_rank_table_rejects_two_ranks_on_one_card():
    rows = _table([(0, 0, "0000:c1:00.0", 4.0), (1, 0, "0000:c1:00.0", 4.1)], 2)
    with pytest.raises(ValueError, match="both run on the card at PCI 0000:c1:00.0"):
        omc_dist.check_rank_table(rows, 2)
    t = omc_dist.check_rank_table(rows, 2, allow_shared_device=True)  # the one-GPU rehearsal (--single-device)
    assert [x["pci_bus_id"] for x in t] == ["0000:c1:00.0"] * 2


def test_rank_table_rejects_a_communicator_without_n_distinct_ranks():
    # a rank that never contributed (a communicator of fewer ranks), and one rank number claimed twice
    with pytest.raises(ValueError, match="rank 1 contributed 0 rows"):
        omc_dist.check_rank_table(_table([(0, 0, "0000:c1:00.0", 4.0), None], 2), 2)
    with pytest.raises(ValueError, match="contributed 2 rows"):
        omc_dist.check_rank_table(_table([(0, 0, "0000:c1:00.0", 4.0), (0, 1, "0000:c2:00.0", 4.0)], 2), 2)
    with pytest.raises(ValueError, match="expected 4"):
        omc_dist.check_rank_table(_table([(0, 0, "0000:c1:00.0", 4.0), (1, 1, "0000:c2:00.0", 4.0)], 2), 4)


def _table_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def all_reduce_sum(vals):
            t = torch.tensor(vals, dtype=torch.float64)
            td.all_reduce(t)
            return t.tolist()

        rows = omc_dist.gather_rank_table(all_reduce_sum, rank, world, rank, f"0000:{0xc1 + rank:02x}:00.0", 0.5 + rank)
        t = omc_dist.check_rank_table(rows, world)
        np.save(os.path.join(out_dir, f"table{rank}.npy"), np.array([[x["rank"], x["device"], x["ms_per_step"]] for x in t]))
        with open(os.path.join(out_dir, f"pci{rank}.txt"), "w") as f:
            f.write(",".join(x["pci_bus_id"] for x in t))
    finally:
        td.destroy_process_group()


def test_rank_table_over_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_table_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "table0.npy"), np.load(tmp_path / "table1.npy")
    assert np.array_equal(t0, t1) and np.array_equal(t0, [[0, 0, 0.5], [1, 1, 1.5]])  # every rank holds the whole table
    assert (tmp_path / "pci0.txt").read_text() == (tmp_path / "pci1.txt").read_text() == "0000:c1:00.0,0000:c2:00.0"


# ---- bench.py's `config3` block at the sizes no one-GPU rehearsal reaches: 8 ranks x 8M paths ---
def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_c3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


class _FakeFfi:
    """Stands where options_model_amd._ffi stands in bench.config3_block: records what each context was asked to price."""

    class OmcError(RuntimeError):
        pass

    def __init__(self, ms_per_path=1e-6, fail_at=None):
        self.calls, self.ms_per_path, self.fail_at = [], ms_per_path, fail_at
        outer = self

        class Context:
            def __init__(self, device):
                self.device = device

            def price_american_seq(self, plist):
                import time as _t
                if outer.fail_at is not None and plist[0]["n_paths"] >= outer.fail_at:
                    raise outer.OmcError("out of memory (test)")
                outer.calls.append(("plain", [(p["n_paths"], p["pair_offset"], p["stream"]) for p in plist]))
                _t.sleep(1e-3 * outer.ms_per_path * plist[0]["n_paths"] * len(plist) / 1e3)
                return [dict(price=7.0 + 1e-3 * p["stream"]) for p in plist]

            def sync(self):
                pass

            def close(self):
                pass

            def set_option(self, key, value):
                outer.calls.append(("option", key, value))

        self.Context = Context

    @staticmethod
    def make_params(**kw):
        return dict(kw)


class _FakePricer:
    def __init__(self, ffi, rank, world, others_ms=0.0, rank0_one=None):
        self.ffi, self.rank, self.world, self.others_ms, self.rank0_one = ffi, rank, world, others_ms, rank0_one

    def price_american_seq(self, n_global, streams, **kw):
        n_local, off = omc_dist.shard(n_global, self.world, self.rank)
        self.ffi.calls.append(("sharded", [(n_global, n_local, off, s) for s in streams]))
        return [dict(price=7.0 + 1e-3 * s, n_paths=n_global, n_exercised=1, n_zero=2, sum_nitm=3) for s in streams]

    def allreduce_max(self, x):
        return x

    def allreduce_sum(self, v):
        v = list(v)
        if len(v) == self.world:      # the shard-alone table: the other ranks' entries
            return [x if i == self.rank else self.others_ms for i, x in enumerate(v)]
        if self.rank != 0 and self.rank0_one is not None:  # rank 0's one-GPU figures reach everybody
            return list(self.rank0_one)
        return v


def _args(**kw):
    import argparse
    d = dict(config3_paths=None, config3_steps=4, paths_per_gpu=None, storage="folded")
    d.update(kw)
    return argparse.Namespace(**d)


def test_config3_block_is_baseline_configs2_at_eight_ranks():
    bench = _bench_module()
    ffi = _FakeFfi(ms_per_path=0.0)
    kw = dict(model="gbm", is_put=True, semantics="two_pass", n_steps=252, seed=42, heston_scheme="reference")
    ctx = ffi.Context(5)
    pr = _FakePricer(ffi, 5, 8, others_ms=4.0, rank0_one=[32.0, 12.003, 1.0])
    b = bench.config3_block(_args(), ffi, ctx, pr, 5, 8, 5, 252, kw, lambda: None, "rccl-native (test)", "on")
    assert (b["total_paths"], b["paths_per_gpu"], b["n_gpus"], b["n_steps"]) == (64_000_000, 8_000_000, 8, 252)
    sharded = [c for c in ffi.calls if c[0] == "sharded"]
    # warm-up then the timed pricings, the whole 64M-path problem, rank 5's shard at pair offset 5 x 4M
    assert sharded[0][1] == [(64_000_000, 8_000_000, 20_000_000, s) for s in (4900, 4901, 4902)]
    assert sharded[1][1] == [(64_000_000, 8_000_000, 20_000_000, s) for s in (5000, 5001, 5002, 5003)]
    plain = [c for c in ffi.calls if c[0] == "plain"]
    # ... the same shard, same streams, through a communicator-free context; and no one-GPU leg on a rank other than 0
    assert plain[1][1] == [(8_000_000, 20_000_000, s) for s in (5000, 5001, 5002, 5003)] and len(plain) == 2
    # the side contexts price with the storage the job's own context uses
    assert [c for c in ffi.calls if c[0] == "option"] == [("option", "fold_antithetic", 1)] and b["storage"] == "folded"
    assert b["one_gpu"] == {"ms_per_step": 32.0, "rank": 0, "price": 12.003}
    assert b["speedup_vs_one_gpu"] == pytest.approx(32.0 / b["ms_per_step"])
    assert b["efficiency_vs_one_gpu"] == pytest.approx(b["speedup_vs_one_gpu"] / 8)
    assert b["price"] == 12.003 and b["price_equals_one_gpu"] is True
    assert len(b["shard_alone_ms"]) == 8 and b["shard_alone_ms_max"] >= 4.0
    assert b["scaling_efficiency"] == pytest.approx(b["shard_alone_ms_mean"] / b["ms_per_step"])


def test_config3_block_rank0_runs_the_whole_problem_and_survives_a_card_without_room():
    bench = _bench_module()
    kw = dict(model="gbm", is_put=True, semantics="two_pass", n_steps=252, seed=42, heston_scheme="reference")
    ffi = _FakeFfi(ms_per_path=0.0)
    b = bench.config3_block(_args(config3_steps=7), ffi, ffi.Context(0), _FakePricer(ffi, 0, 4), 0, 4, 0, 252, kw,
                            lambda: None, "c", "on")
    plain = [c for c in ffi.calls if c[0] == "plain"]
    # rank 0: its shard alone (16M paths at offset 0), then ALL 64M paths on its card: 2 warm-ups, the LAST 5 streams timed,
    # so that its last price is the sharded job's last price (stream 5006)
    assert plain[1][1][0][:2] == (16_000_000, 0)
    assert plain[2][1] == [(64_000_000, 0, 4900), (64_000_000, 0, 4901)]
    assert plain[3][1] == [(64_000_000, 0, s) for s in (5002, 5003, 5004, 5005, 5006)]
    assert b["one_gpu"]["price"] == b["price"] == 7.0 + 1e-3 * 5006 and b["price_equals_one_gpu"] is True
    # a card without room for 65 GB: the one-GPU leg is skipped, the block and the job go on
    ffi = _FakeFfi(ms_per_path=0.0, fail_at=64_000_000)
    b = bench.config3_block(_args(), ffi, ffi.Context(0), _FakePricer(ffi, 0, 4), 0, 4, 0, 252, kw, lambda: None, "c", "on")
    assert b["one_gpu"] is None and b["speedup_vs_one_gpu"] is None and b["scaling_efficiency"] > 0
    # sizes: 64 x the headline's paths per GPU unless given; rounded down to whole groups of 4 pairs per rank
    b = bench.config3_block(_args(paths_per_gpu=1000), ffi, ffi.Context(0), _FakePricer(ffi, 2, 3), 2, 3, 0, 50, kw,
                            lambda: None, "c", "on")
    assert b["total_paths"] == 64_000 // 24 * 24 and b["paths_per_gpu"] == b["total_paths"] // 3
    # one rank (the N = 1 line of a scaling run): the whole problem through the line's own context
    ffi = _FakeFfi(ms_per_path=0.0)
    b = bench.config3_block(_args(), ffi, ffi.Context(0), None, 0, 1, 0, 252, kw, lambda: None, "none", "n/a")
    assert b["total_paths"] == 64_000_000 and b["speedup_vs_one_gpu"] == 1.0 and b["n_gpus"] == 1
    assert [c[1][0][0] for c in ffi.calls] == [64_000_000, 64_000_000]
