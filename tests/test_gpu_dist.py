"""GPU tests of the multi-GPU plumbing that can run on ONE card: the C-ABI all-reduce hook, the
torch tensor aliasing of the device moment table, stream ordering, and the two-shard algebra
(shard A's moments injected into shard B's all-reduce and vice versa == unsharded pricing).
The real 8-GPU RCCL run is the driver's; tests/test_dist_cpu.py covers the gloo side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0


@pytest.fixture(scope="module")
def tctx():
    import torch

    from options_model_amd import _ffi
    from options_model_amd.dist import _DevPtr

    assert torch.cuda.is_available()
    stream = torch.cuda.Stream()
    c = _ffi.Context(0, stream=stream.cuda_stream)
    yield torch, stream, c, _DevPtr
    c.close()


def test_identity_hook_matches_hookless_run(tctx, ctx):
    """hook = in-place no-op through a torch alias: exercises the external-moments code path
    of every flow (per-step flows read gmom[t] instead of the launch partials)."""
    torch, stream, c, DevPtr = tctx
    from options_model_amd import _ffi
    calls = []

    def hook(dptr, count):
        t = torch.as_tensor(DevPtr(dptr, count), device="cuda")
        t.add_(0.0)
        calls.append(count)

    # moments: once (two-pass) or once per step t = N-1..1 (per-step flows); + the 8 result sums
    for sem, ncalls in (("two_pass", 2), ("reference", 30), ("textbook", 30)):
        p = _ffi.make_params(semantics=sem, n_paths=20000, n_steps=30, seed=5)
        base = ctx.price_american(p)
        calls.clear()
        c.set_allreduce_hook(hook)
        with torch.cuda.stream(stream):
            out = c.price_american(p)
        c.set_allreduce_hook(None)
        assert len(calls) == ncalls and set(calls) == ({8 * 31, 8} if sem == "two_pass" else {8})
        assert out["price"] == pytest.approx(base["price"], rel=1e-12)
        assert (out["n_exercised"], out["n_zero"], out["sum_nitm"]) == (
            base["n_exercised"], base["n_zero"], base["sum_nitm"])


def test_two_shard_emulation_equals_unsharded(tctx, ctx):
    torch, stream, c, DevPtr = tctx
    from options_model_amd import _ffi
    from options_model_amd import dist as omc_dist
    M, N = 16384, 24
    shards = [omc_dist.shard(M, 2, r) for r in range(2)]
    mom = {}

    def capture(key):
        def h(dptr, count):
            if count == 8 * (N + 1):  # the moment table (the 8 result sums stay local here)
                mom[key] = torch.as_tensor(DevPtr(dptr, count), device="cuda").clone()
        return h

    def inject(other):
        def h(dptr, count):
            if count == 8 * (N + 1):
                torch.as_tensor(DevPtr(dptr, count), device="cuda").add_(mom[other])
        return h

    def run(rank, hook):
        n, off = shards[rank]
        c.set_allreduce_hook(hook)
        with torch.cuda.stream(stream):
            out = c.price_american(_ffi.make_params(semantics="two_pass", n_paths=n, n_steps=N, seed=9,
                                                    pair_offset=off))
        c.set_allreduce_hook(None)
        return out

    run(0, capture(0))
    run(1, capture(1))
    a = run(0, inject(1))
    b = run(1, inject(0))
    merged = omc_dist.merge(a, lambda v: [x + y for x, y in zip(v, [float(b[k]) for k in omc_dist.SUM_KEYS])])
    full = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=9))
    assert merged["n_paths"] == M
    assert merged["price"] == pytest.approx(full["price"], rel=1e-12)
    assert merged["n_exercised"] == full["n_exercised"] and merged["n_zero"] == full["n_zero"]
    # each emulated rank saw the GLOBAL moment table: its regression-set total is the unsharded one
    assert a["sum_nitm"] == full["sum_nitm"] and b["sum_nitm"] == full["sum_nitm"]


def test_world_size_normalisation_with_doubling_hook(tctx, ctx):
    """Two identical 'ranks': an all-reduce that doubles every buffer.  Moments x2 give the same
    fits; the library's own result sums go through the hook too and are normalised by
    n_local * world_size, so the price is unchanged and the counts double."""
    torch, stream, c, DevPtr = tctx
    from options_model_amd import _ffi

    def hook(dptr, count):
        torch.as_tensor(DevPtr(dptr, count), device="cuda").mul_(2.0)

    for sem in ("two_pass", "reference"):
        p = _ffi.make_params(semantics=sem, n_paths=30000, n_steps=20, seed=8)
        base = ctx.price_american(p)
        c.set_allreduce_hook(hook)
        c.set_option("world_size", 2)
        with torch.cuda.stream(stream):
            out = c.price_american(p)
            # a sequence defers the result sums of all its pricings to ONE hook call of 8n doubles
            calls = []
            c.set_allreduce_hook(lambda dptr, count: (calls.append(count), hook(dptr, count)))
            seq = c.price_american_seq([p, p, p])
        c.set_allreduce_hook(None)
        c.set_option("world_size", 1)
        # two_pass: 3 moment tables + ONE collective of 24 result sums; reference: the three pricings share their
        # launches, so every time step sends ONE collective of 3 x 8 moments (19 of them), then the 24 result sums
        assert calls[-1] == 24 and calls.count(24) == (1 if sem == "two_pass" else 20)
        for o in seq:
            assert (o["price"], o["n_exercised"], o["sum_nitm"], o["n_paths"]) == (
                out["price"], out["n_exercised"], out["sum_nitm"], out["n_paths"])
        assert out["n_paths"] == 60000 and out["n_exercised"] == 2 * base["n_exercised"]
        # the regression-set sizes come from the moment table, which is global on every rank BEFORE the
        # result sums are all-reduced: they must come out once per rank, not world_size times each
        assert out["sum_nitm"] == 2 * base["sum_nitm"]
        assert out["price"] == pytest.approx(base["price"], rel=1e-12)
        assert out["std"] == pytest.approx(base["std"], rel=1e-9)


def test_sharded_pricer_world_size_one():
    """ShardedPricer end to end with a 1-rank gloo+nccl-free group is not possible; with
    world_size 1 over nccl it must reduce to the plain pricing."""
    import os

    import torch
    import torch.distributed as td

    from options_model_amd import _ffi
    from options_model_amd import dist as omc_dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sp = omc_dist.ShardedPricer(0)
        out = sp.price_american(50_000, semantics="two_pass", n_steps=40, seed=3)
        ref = _ffi.Context(0)
        base = ref.price_american(_ffi.make_params(semantics="two_pass", n_paths=50_000, n_steps=40, seed=3))
        ref.close()
        assert out["price"] == pytest.approx(base["price"], rel=1e-12)
        assert out["n_paths"] == 50_000 and out["stderr"] > 0
        sp.close()
        # the real collective on the library's own device buffers (torch alias of a hipMalloc'ed
        # pointer handed to RCCL), for every flow: 1-rank all-reduce is the identity
        sp = omc_dist.ShardedPricer(0, force_hook=True)
        for sem in ("two_pass", "reference", "textbook"):
            out = sp.price_american(20_000, semantics=sem, n_steps=12, seed=4)
            base = _ffi.default_context(0).price_american(
                _ffi.make_params(semantics=sem, n_paths=20_000, n_steps=12, seed=4))
            assert out["price"] == pytest.approx(base["price"], rel=1e-12)
            assert out["n_exercised"] == base["n_exercised"]
        sp.close()
    finally:
        td.destroy_process_group()


def test_two_ranks_through_bench_equal_the_unsharded_price(ctx, tmp_path):
    """world_size 2 end to end (torch.distributed.run, two processes sharing this GPU, gloo standing in
    for RCCL): the sharded pricing returns the price of the unsharded 2x-path problem."""
    import json
    import os
    import subprocess
    import sys
    from options_model_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--paths-per-gpu", "100000", "--n-steps", "50", "--backend", "gloo", "--single-device",
           "--no-variants", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "path-sharded x2"
    ref = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=200000, n_steps=50, seed=42, stream=1))
    assert d["price"] == pytest.approx(ref["price"], rel=1e-10)


def _bench_json(cmd, root, timeout=420):
    import json
    import os
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks(ctx):
    """`python bench.py --gpus 2` WITHOUT torchrun: bench.py itself starts two rank processes (before any
    GPU call in the parent), they form a 2-rank group (gloo here, both on this one GPU) and the sharded
    pricing equals the unsharded 2x-path problem."""
    import os
    import sys
    from options_model_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                     "--paths-per-gpu", "100000", "--n-steps", "50", "--backend", "gloo", "--single-device",
                     "--no-variants", "--no-cpu-baseline", "--no-sustained"], root)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["parallelism"] == "path-sharded x2"
    ref = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=200000, n_steps=50, seed=42, stream=1))
    assert d["price"] == pytest.approx(ref["price"], rel=1e-10)
    assert d["price_check"]["rel_err"] < 1e-3 and d["price_check"]["same_stream"]
    # the config3 block through the torch.distributed hook transport too: 64 x the headline's paths, sharded two ways
    c3 = d["config3"]
    assert (c3["total_paths"], c3["paths_per_gpu"], c3["n_gpus"]) == (6_400_000, 3_200_000, 2) and c3["comm"].startswith("torch.distributed gloo")
    assert c3["price_equals_one_gpu"] is True and len(c3["shard_alone_ms"]) == 2 and c3["speedup_vs_one_gpu"] > 0
    # the per-step flow went through the 2-rank exchange too
    ps = d["roofline_per_step"]  # (several pricings per launch: K moment vectors per collective)
    ref2 = ctx.price_american(_ffi.make_params(semantics="reference", n_paths=200000, n_steps=50, seed=42,
                                               stream=ps["price_stream"]))
    assert ps["price"] == pytest.approx(ref2["price"], rel=1e-10) and ps["pricings_per_launch"] >= 4


def test_bench_gpus_mismatch_fails_loudly():
    """--gpus must equal the launcher's world size: a silent single-rank run is what round 1 did."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=120, cwd=root)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_native_rccl_communicator_world_size_one(ctx):
    """omc_comm_init / ncclAllReduce from inside libomc.so (no torch.distributed): with one rank every
    all-reduce is the identity, so each flow must reproduce the plain pricing -- through the external-
    moments code path and real RCCL calls on the library's own stream and buffers."""
    from options_model_amd import _ffi
    from options_model_amd import dist as omc_dist
    sp = omc_dist.RcclPricer(0, 0, 1, tag=f"pytest_{__import__('os').getpid()}")
    try:
        assert sp.ctx.comm_info() == (0, 1) and sp.comm_ranks() == 1
        assert sp.allreduce_max(3.5) == 3.5
        sp.barrier()
        for sem in ("two_pass", "reference", "textbook"):
            out = sp.price_american(20_000, semantics=sem, n_steps=12, seed=4)
            base = ctx.price_american(_ffi.make_params(semantics=sem, n_paths=20_000, n_steps=12, seed=4))
            assert out["price"] == pytest.approx(base["price"], rel=1e-12)
            assert (out["n_exercised"], out["n_zero"], out["sum_nitm"]) == (
                base["n_exercised"], base["n_zero"], base["sum_nitm"])
        outs = sp.price_american_seq(20_000, [1, 2, 3], semantics="two_pass", n_steps=12, seed=4)
        for s, o in zip((1, 2, 3), outs):
            base = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=20_000, n_steps=12, seed=4,
                                                       stream=s))
            assert o["price"] == pytest.approx(base["price"], rel=1e-12)
        # the overlapped sequence (collective of pricing k under the path generation of pricing k+1, two path
        # buffers, one result collective at the end) against one pricing after the other: the same bits
        streams = list(range(1, 8))
        sp.ctx.set_option("seq_overlap", 1)   # (default: only with more than one rank)
        on = sp.price_american_seq(300_000, streams, semantics="two_pass", n_steps=40, seed=9)
        sp.ctx.set_option("seq_overlap", 0)
        off = sp.price_american_seq(300_000, streams, semantics="two_pass", n_steps=40, seed=9)
        sp.ctx.set_option("seq_overlap", -1)
        for a, b, s in zip(on, off, streams):
            assert (a["price"], a["sumsq"], a["n_exercised"], a["n_zero"], a["sum_nitm"]) == (
                b["price"], b["sumsq"], b["n_exercised"], b["n_zero"], b["sum_nitm"])
            base = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=300_000, n_steps=40, seed=9, stream=s))
            assert a["price"] == base["price"] and a["sum_nitm"] == base["sum_nitm"]
        assert len({o["price"] for o in on}) == len(streams)
    finally:
        sp.close()


def test_bench_single_gpu_line_carries_config3(ctx):
    """The N = 1 line of a scaling run: the same headline as ever plus the `config3` block -- BASELINE configs[2]'s whole
    problem (64 x the headline's paths per GPU) on this one card, the denominator of the strong-scaling curve the N > 1
    lines continue (tests/test_gpu_multirank.py checks those through the stand-in)."""
    import os
    import sys
    from options_model_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                     "--paths-per-gpu", "50000", "--n-steps", "40", "--no-variants", "--no-cpu-baseline", "--no-sustained",
                     "--config3-steps", "2"], root)
    c3 = d["config3"]
    assert (c3["total_paths"], c3["n_gpus"], c3["paths_per_gpu"], c3["scaling"]) == (3_200_000, 1, 3_200_000, "strong")
    ref = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=3_200_000, n_steps=40, seed=42, stream=5001))
    assert c3["price"] == ref["price"] and c3["speedup_vs_one_gpu"] == 1.0
    assert c3["value"] == pytest.approx(3_200_000 * 40 / (c3["ms_per_step"] * 1e-3))
    assert d["config"]["paths_per_gpu"] == 50000 and d["scaling"] == "weak"  # the headline is untouched
    d = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                     "--paths-per-gpu", "50000", "--n-steps", "40", "--no-variants", "--no-cpu-baseline", "--no-sustained",
                     "--no-config3"], root)
    assert "config3" not in d


def test_bench_force_dist_uses_native_rccl():
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                     "--paths-per-gpu", "100000", "--n-steps", "50", "--force-dist", "--no-variants",
                     "--no-cpu-baseline", "--no-sustained"], root)
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["comm"].startswith("rccl-native")
    assert d["price_check"]["rel_err"] < 1e-3 and d["seq_overlap"] == "off (one rank)"
    # the multi-rank start-up self-check (overlapped sequence == sequential one, bit for bit), rehearsed with one rank
    os.environ["OMC_BENCH_SELFCHECK"] = "1"
    try:
        d2 = _bench_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--paths-per-gpu", "100000", "--n-steps", "50", "--force-dist", "--no-variants",
                          "--no-cpu-baseline", "--no-sustained"], root)
    finally:
        del os.environ["OMC_BENCH_SELFCHECK"]
    assert d2["seq_overlap"] == "on" and d2["price_check"]["rel_err"] < 1e-3
