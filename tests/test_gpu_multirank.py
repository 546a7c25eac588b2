"""The N > 1 native-communicator path of libomc.so with 2 and 4 ranks -- on ONE GPU.

Real RCCL refuses two ranks on one device, so csrc/omc_comm.hip is pointed (through its OMC_RCCL_LIB hook) at
tests/rccl_standin: the seven ncclXxx entry points over POSIX shared memory, all-reduces enqueued on the given
stream like RCCL's.  Everything above that library is the product's own code, run exactly as the driver's 8-GPU
job runs it: bench.py's launcher, the unique-id rendezvous, the collective transport decision, omc_comm_init, the
251 per-step all-reduces enqueued from C, the overlapped two-stream sequence and its one result collective.
(Ranks are limited to 4 here: a GPU box admits at most 6 processes on its card, and pytest is one of them.)
"""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M_PER_GPU, N = 100_000, 50


@pytest.fixture(scope="module")
def standin():
    import rccl_standin
    return rccl_standin.build()


def _bench(args, standin, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMC_RCCL_LIB=standin, MASTER_ADDR="127.0.0.1")
    env.update(extra_env or {})
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=timeout, cwd=ROOT)
    return out, time.monotonic() - t0


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("sem", ["two_pass", "reference", "textbook"])
def test_ranks_sharing_one_gpu_equal_the_unsharded_pricing(ctx, standin, world, sem):
    from options_model_amd import _ffi
    out, _ = _bench(["--gpus", str(world), "--single-device", "--backend", "rccl", "--steps", "4", "--warmup", "2",
                     "--paths-per-gpu", str(M_PER_GPU), "--n-steps", str(N), "--semantics", sem, "--group", "4",
                     "--min-warmup-seconds", "0.05", "--no-variants", "--no-cpu-baseline", "--no-sustained",
                     "--config3-steps", "3"], standin)
    d = _line(out)
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["comm"].startswith("rccl-native")
    # the line says who ran where (gathered through the communicator): `world` distinct ranks, here all on this one card
    # (--single-device; without that flag two ranks on one PCI id end the job: tests/test_dist_cpu.py)
    assert [t["rank"] for t in d["ranks"]] == list(range(world)) and d["distinct_gpus"] == 1
    assert len({t["pci_bus_id"] for t in d["ranks"]}) == 1 and all(t["ms_per_step"] > 0 for t in d["ranks"])
    assert d["ranks"][d["slowest_rank"]]["ms_per_step"] <= d["ms_per_step"] * 1.0001
    # the sharded job == ONE pricing of world x paths: same Philox pairs (global pair index), sums in another order
    ref = ctx.price_american(_ffi.make_params(semantics=sem, n_paths=world * M_PER_GPU, n_steps=N, seed=42, stream=3))
    assert d["price"] == pytest.approx(ref["price"], rel=1e-12)
    lp = d["last_pricing"]
    assert (lp["n_paths"], lp["n_exercised"], lp["n_zero"], lp["sum_nitm"]) == (
        ref["n_paths"], ref["n_exercised"], ref["n_zero"], ref["sum_nitm"])
    # the two-pass job runs on antithetic-folded storage (the rule looks at the JOB's 200k / 400k paths, not at a rank's share),
    # the unsharded pricing it is compared with likewise; the per-step flows keep the full matrix
    assert ref["folded"] == (1 if sem == "two_pass" else 0)
    assert d["config"]["storage"].startswith("antithetic-folded" if sem == "two_pass" else "full")
    if sem == "two_pass":
        # before timing, every rank priced three streams overlapped (moment all-reduce on a second stream under the
        # next pricing's paths + pass 1) and one after the other, and all ranks saw the same bits
        assert d["seq_overlap"] == "on"
        ps = d["roofline_per_step"]  # ... and the per-step reference flow went through 49 all-reduces per pricing
        ref2 = ctx.price_american(_ffi.make_params(semantics="reference", n_paths=world * M_PER_GPU, n_steps=N,
                                                   seed=42, stream=ps["price_stream"]))
        assert ps["price"] == pytest.approx(ref2["price"], rel=1e-12)
        assert ps["last_pricing"]["sum_nitm"] == ref2["sum_nitm"]
        assert ps["last_pricing"]["n_exercised"] == ref2["n_exercised"]
        # the `config3` block: BASELINE configs[2]'s FIXED problem (64 x the headline's paths per GPU: 64M paths when the
        # headline is config 2's 1M) sharded over this job's ranks = strong scaling, with every rank's communicator-free
        # time for its own shard and rank 0's one-GPU time of the whole problem beside it
        c3 = d["config3"]
        total = 64 * M_PER_GPU
        assert (c3["total_paths"], c3["paths_per_gpu"], c3["n_gpus"], c3["scaling"]) == (total, total // world, world, "strong")
        assert c3["comm"].startswith("rccl-native") and c3["seq_overlap"] == "on" and c3["steps"] == 3
        ref3 = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=total, n_steps=N, seed=42, stream=5002))
        assert c3["price"] == pytest.approx(ref3["price"], rel=1e-12)
        assert c3["one_gpu"]["price"] == ref3["price"] and c3["price_equals_one_gpu"] is True
        assert c3["last_pricing"]["sum_nitm"] == ref3["sum_nitm"] and c3["last_pricing"]["n_paths"] == total
        assert len(c3["shard_alone_ms"]) == world and all(t > 0 for t in c3["shard_alone_ms"])
        assert c3["scaling_efficiency"] == pytest.approx(c3["shard_alone_ms_mean"] / c3["ms_per_step"])
        assert c3["speedup_vs_one_gpu"] == pytest.approx(c3["one_gpu"]["ms_per_step"] / c3["ms_per_step"])
        assert c3["value"] == pytest.approx(total * N / (c3["ms_per_step"] * 1e-3))
    else:
        assert "config3" not in d  # the block belongs to the headline flow


def test_a_rank_whose_init_fails_ends_the_job_within_the_deadline(standin):
    """Rank 1's ncclCommInitRank fails at once; rank 0 sits inside its own (a dead peer, as real RCCL would see it:
    the stand-in's own bound is set far away).  The launcher's deadline must end the job, non-zero, naming the rank."""
    out, took = _bench(["--gpus", "2", "--single-device", "--backend", "rccl", "--steps", "2", "--warmup", "1",
                        "--paths-per-gpu", "20000", "--n-steps", "20", "--rank-timeout", "25", "--no-variants",
                        "--no-cpu-baseline", "--no-sustained"], standin,
                       {"OMC_STANDIN_FAIL_RANK": "1", "OMC_STANDIN_TIMEOUT_S": "600"})
    assert out.returncode != 0 and took < 90
    assert out.stdout.strip() == ""
    assert "did not finish within --rank-timeout" in out.stderr or "exited with" in out.stderr


def test_failed_init_is_a_collective_fallback(standin):
    """Rank 1's init fails, rank 0's gives up after 3 s: both vote, both learn that the native communicator is not
    usable for this JOB, and both switch to the torch.distributed transport together (gloo here: one device)."""
    out, _ = _bench(["--gpus", "2", "--single-device", "--backend", "rccl", "--steps", "2", "--warmup", "1",
                     "--paths-per-gpu", "20000", "--n-steps", "20", "--no-variants", "--no-cpu-baseline",
                     "--no-sustained", "--min-warmup-seconds", "0.05"], standin,
                    {"OMC_STANDIN_FAIL_RANK": "1", "OMC_STANDIN_TIMEOUT_S": "3"})
    d = _line(out)
    assert d["rccl_ranks"] == 2 and d["comm"].startswith("torch.distributed gloo")
    assert out.stderr.count("native RCCL unavailable for this job") == 2


_FACADE = r"""
import json, os, sys
sys.path.insert(0, %r)
from options_model_amd import price_american_option
out = {}
for sem in ("two_pass", "per_step", "textbook"):
    r = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, model="GBM", option_type="put",
                              semantics=sem, seed=5, stream=2, n_gpus=2, device=0)
    out[sem] = [r.price, r.n_paths, r.n_exercised, r.sum_nitm, r.stderr, r.info["transport"]]
r = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, model="Heston", option_type="call",
                          heston_scheme="full_truncation", seed=5, n_gpus=2, device=0)
out["heston"] = [r.price, r.n_paths, r.n_exercised, r.sum_nitm, r.stderr, r.info["transport"]]
print("RESULT" + os.environ["RANK"] + " " + json.dumps(out))
"""


def test_facade_n_gpus_2_shards_through_the_native_communicator(ctx, standin, tmp_path):
    """North-star facade, n_gpus=2: two rank processes (sharing this GPU through the stand-in) call
    price_american_option(..., n_gpus=2); both get the price of the unsharded pricing."""
    from options_model_amd import price_american_option
    script = tmp_path / "facade2.py"
    script.write_text(_FACADE % ROOT)
    port = 29700 + os.getpid() % 200
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"facade{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    got = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT")][0]
        got.append(json.loads(line.split(" ", 1)[1]))
    assert got[0] == got[1]                      # every rank returns the same global result
    for sem in ("two_pass", "per_step", "textbook"):
        one = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, semantics=sem, seed=5, stream=2, ctx=ctx)
        price, n, nex, nitm, stderr, transport = got[0][sem]
        assert transport == "rccl-native" and n == 200_000
        assert price == pytest.approx(one.price, rel=1e-12) and stderr == pytest.approx(one.stderr, rel=1e-9)
        assert (nex, nitm) == (one.n_exercised, one.sum_nitm)
    one = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, model="Heston", option_type="call",
                                heston_scheme="full_truncation", seed=5, ctx=ctx)
    assert got[0]["heston"][0] == pytest.approx(one.price, rel=1e-12)


@pytest.mark.parametrize("world", [2, 4])
def test_direct_peer_exchange_equals_the_collective_path(ctx, standin, world):
    """SURVEY 5.8(b) / VERDICT r2 item 8: the per-step flows' moments by direct writes into every peer's mailbox
    (hipIpc-mapped fine-grained memory, system-scope stores and polls, bounded) instead of an all-reduce per step.
    Ranks share this GPU; both paths add the ranks' contributions in rank order, so the results are the same BITS:
    compared against the unsharded pricing exactly as the collective path is, for one pricing per launch and for K."""
    from options_model_amd import _ffi
    out, _ = _bench(["--gpus", str(world), "--single-device", "--backend", "rccl", "--steps", "4", "--warmup", "2",
                     "--paths-per-gpu", str(M_PER_GPU), "--n-steps", str(N), "--group", "4", "--p2p-exchange",
                     "--min-warmup-seconds", "0.05", "--no-variants", "--no-cpu-baseline", "--no-sustained"], standin)
    d = _line(out)
    ps = d["roofline_per_step"]
    assert ps["exchange_across_ranks"].startswith("direct writes"), ps["exchange_across_ranks"]
    ref = ctx.price_american(_ffi.make_params(semantics="reference", n_paths=world * M_PER_GPU, n_steps=N, seed=42,
                                              stream=ps["price_stream"]))
    assert ps["price"] == pytest.approx(ref["price"], rel=1e-12) and ps["pricings_per_launch"] >= 4
    assert ps["last_pricing"]["sum_nitm"] == ref["sum_nitm"] and ps["last_pricing"]["n_exercised"] == ref["n_exercised"]


_P2P = r"""
import json, os, sys
sys.path.insert(0, %r)
from options_model_amd import _ffi
from options_model_amd.dist import RcclPricer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
sp = RcclPricer(0, rank, world)
kw = dict(semantics="reference", n_steps=30, seed=9)
coll1 = sp.price_american(120_000 * world, stream=1, **kw)
collK = sp.price_american_seq(120_000 * world, [2, 3, 4, 5, 6], **kw)
assert sp.enable_p2p()
assert sp.ctx.p2p_status() == (True, world, 0)
p2p1 = sp.price_american(120_000 * world, stream=1, **kw)
p2pK = sp.price_american_seq(120_000 * world, [2, 3, 4, 5, 6], **kw)
tb = sp.price_american(120_000 * world, stream=1, **dict(kw, semantics="textbook"))
sp.ctx.set_option("p2p_exchange", 0)
tb0 = sp.price_american(120_000 * world, stream=1, **dict(kw, semantics="textbook"))
keys = ("price", "sumsq", "n_exercised", "sum_nitm")
same = all(p2p1[k] == coll1[k] for k in keys) and all(a[k] == b[k] for a, b in zip(p2pK, collK) for k in keys) \
    and all(tb[k] == tb0[k] for k in keys)
print("RESULT" + str(rank) + " " + json.dumps(dict(same=same, price=p2p1["price"], status=list(sp.ctx.p2p_status()))))
sp.close()
"""


def test_direct_exchange_bitwise_equals_collective_in_process(standin, tmp_path):
    """The same comparison without bench.py: two rank processes price through the collective and through the direct
    exchange (single pricing, a sequence of five sharing their launches, the textbook flow) -- identical bits."""
    script = tmp_path / "p2p2.py"
    script.write_text(_P2P % ROOT)
    port = 29900 + os.getpid() % 90
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"p2p{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    got = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2500:]
        got.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT")][0].split(" ", 1)[1]))
    assert got[0]["same"] and got[1]["same"] and got[0]["price"] == got[1]["price"]
    assert got[0]["status"] == [True, 2, 0]


_TIGHT = r"""
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == 1:
    os.environ["OMC_SEQ_STEP_BYTES"] = "1"       # this rank's card "has no room": its own K comes out as 1
sys.path.insert(0, %r)
from options_model_amd import _ffi
kw = dict(semantics="reference", n_steps=30, seed=9)
if os.environ.get("TIGHT_TRANSPORT") == "hook":
    import torch, torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from options_model_amd.dist import ShardedPricer
    sp = ShardedPricer(0)
else:
    from options_model_amd.dist import RcclPricer
    sp = RcclPricer(0, rank, world)
p1 = _ffi.make_params(n_paths=120_000, n_steps=30, semantics="reference", seed=9)
local_width = sp.ctx.lib.omc_seq_step_width(sp.ctx.handle, (type(p1) * 5)(*[p1] * 5), 5)
seq = sp.price_american_seq(120_000 * world, [2, 3, 4, 5, 6], **kw)
if os.environ.get("TIGHT_TRANSPORT") != "hook":
    assert sp.enable_p2p()
    seq_p2p = sp.price_american_seq(120_000 * world, [2, 3, 4, 5, 6], **kw)
else:
    seq_p2p = seq
one = [sp.price_american(120_000 * world, stream=s, **kw) for s in (2, 3, 4, 5, 6)]
keys = ("price", "sumsq", "n_exercised", "sum_nitm")
same = all(a[k] == b[k] == c[k] for a, b, c in zip(seq, one, seq_p2p) for k in keys)
print("RESULT" + str(rank) + " " + json.dumps(dict(same=same, width=local_width, prices=[r["price"] for r in seq])))
sp.close()
"""


@pytest.mark.parametrize("transport", ["comm", "hook"])
def test_a_memory_tight_rank_votes_the_whole_job_down(ctx, standin, tmp_path, transport):
    """the number of pricings per launch is bounded by each rank's free memory, so it is
    rank-dependent.  Rank 1 here has a budget of one byte (its own estimate: 1 pricing per launch), rank 0 plenty
    (5).  Every rank votes -- also the one whose K is 1 -- through the context's generic all-reduce, native
    communicator or hook alike, and the job runs at the smallest K: no rank skips a collective its peer enters (the
    hang this test used to be), the per-step collectives carry the same 8K doubles everywhere, the direct exchange the
    same K jobs, and the results are the bits of one pricing after the other."""
    from options_model_amd import _ffi
    script = tmp_path / "tight.py"
    script.write_text(_TIGHT % ROOT)
    port = 29800 + os.getpid() % 90 + (7 if transport == "hook" else 0)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"tight{transport}{os.getpid()}",
                   TIGHT_TRANSPORT=transport)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=200) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    got = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2500:]
        got.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT")][0].split(" ", 1)[1]))
    assert got[0]["same"] and got[1]["same"] and got[0]["prices"] == got[1]["prices"]
    assert got[0]["width"] == 5 and got[1]["width"] == 1          # the ranks' own estimates DO differ
    ref = ctx.price_american(_ffi.make_params(semantics="reference", n_paths=240_000, n_steps=30, seed=9, stream=2))
    assert got[0]["prices"][0] == pytest.approx(ref["price"], rel=1e-12)


_OLS7_TIGHT = r"""
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
sys.path.insert(0, %r)
from options_model_amd import _ffi
from options_model_amd.dist import RcclPricer
sp = RcclPricer(0, rank, world)
kw = dict(semantics="two_pass", n_steps=40, seed=11, stream=2)
if rank == 1:
    sp.ctx.set_option("alloc_limit", 16 << 20)   # this rank's card "has no room" for its 32.8 MB path matrix
try:
    sp.price_american_ols7(400_000, **kw)
    err = None
except _ffi.OmcError as e:
    err = str(e)
sp.ctx.set_option("alloc_limit", 0)
out = sp.price_american_ols7(400_000, **kw)      # the communicator is still in step: the next collective call works
print("RESULT" + str(rank) + " " + json.dumps(dict(err=err, price=out["price"], sum_nitm=out["sum_nitm"])))
sp.close()
"""


def test_ols7_a_rank_without_room_for_its_paths_fails_the_call_on_every_rank(ctx, standin, tmp_path):
    """omc_price_american_ols7 is collective on a distributed context; rank 1's path matrix does not fit
    (option "alloc_limit").  It used to return before any collective and leave rank 0 inside the first all-reduce for
    ever.  Now the failure travels as a flag: rank 1 reports its own error, rank 0 "another rank ..." (3103), both AFTER the
    all-reduce -- and the very next collective call of the two ranks prices the unsharded problem."""
    from options_model_amd import _ffi
    script = tmp_path / "ols7_tight.py"
    script.write_text(_OLS7_TIGHT % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(29890 + os.getpid() % 60), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"ols7tight{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=200) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    got = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2500:]
        got.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT")][0].split(" ", 1)[1]))
    assert "another rank of the job" in got[0]["err"] and "alloc_limit" in got[1]["err"]
    one = ctx.price_american_ols7(_ffi.make_params(semantics="two_pass", n_paths=400_000, n_steps=40, seed=11, stream=2))
    assert got[0]["price"] == got[1]["price"] == pytest.approx(one["price"], rel=1e-12)
    assert got[0]["sum_nitm"] == one["sum_nitm"]


def test_the_drivers_launch_form_torchrun_native_communicator(ctx, standin):
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`: the ranks are torchrun's children (common parent = its agent, which names
    the rendezvous files together with MASTER_PORT and TORCHELASTIC_RUN_ID), bench.py must NOT start ranks of its own,
    and the default transport is the native communicator."""
    from options_model_amd import _ffi
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMC_RCCL_LIB=standin, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29650 + os.getpid() % 40), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "4", "--warmup", "2", "--paths-per-gpu", str(M_PER_GPU), "--n-steps", str(N), "--single-device",
           "--backend", "rccl", "--group", "4", "--min-warmup-seconds", "0.05", "--no-variants", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    d = _line(out)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["comm"].startswith("rccl-native") and d["seq_overlap"] == "on"
    ref = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=2 * M_PER_GPU, n_steps=N, seed=42, stream=3))
    assert d["price"] == pytest.approx(ref["price"], rel=1e-12)
    assert d["sustained"]["pricings"] > 0 and d["clock_settled"] in (True, False)


_P2P_LOST = r"""
import os, sys, time
sys.path.insert(0, %r)
from options_model_amd import _ffi
from options_model_amd.dist import RcclPricer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
sp = RcclPricer(0, rank, world)
assert sp.enable_p2p()
sp.ctx.set_option("p2p_first_deadline_ms", 2000)            # (default 30 s: a call's first exchange absorbs start-up skew)
kw = dict(semantics="reference", n_steps=12, seed=9)
sp.price_american(40_000 * world, stream=1, **kw)          # one good pricing: the mailboxes work
if rank == 1:
    time.sleep(20)                                           # rank 1 goes missing (it never sends another contribution)
    os._exit(0)
t0 = time.monotonic()
try:
    sp.price_american(40_000 * world, stream=2, **kw)
    print("RESULT no error")
except _ffi.OmcError as e:
    print("RESULT error after %%.1f s: %%s" %% (time.monotonic() - t0, e))
    print("STATUS", sp.ctx.p2p_status())
os._exit(0)
"""


def test_a_lost_peer_ends_the_direct_exchange_with_an_error_not_a_hang(standin, tmp_path):
    """Bounded waits: when a peer's contribution never arrives, the exchange kernel gives up at its deadline, marks the
    mailbox (sticky: later exchanges of the call return at once) and the pricing call FAILS with error 3100 -- it does
    not hang and it does not return a price built on missing data."""
    script = tmp_path / "p2p_lost.py"
    script.write_text(_P2P_LOST % ROOT)
    port = 29950 + os.getpid() % 40
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"lost{os.getpid()}",
                   OMC_STANDIN_TIMEOUT_S="4")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    t0 = time.monotonic()
    so, se = procs[0].communicate(timeout=120)
    took = time.monotonic() - t0
    procs[1].communicate(timeout=120)
    assert procs[0].returncode == 0, se[-2000:]
    line = [ln for ln in so.splitlines() if ln.startswith("RESULT")][0]
    assert "error after" in line and "direct peer exchange" in line, (line, se[-1500:])
    status = [ln for ln in so.splitlines() if ln.startswith("STATUS")][0]
    assert status.endswith(", 1)"), status       # the sticky error word
    assert took < 60

_P2P_SLOW = r"""
import os, sys, time
sys.path.insert(0, %r)
from options_model_amd import _ffi
from options_model_amd.dist import RcclPricer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
sp = RcclPricer(0, rank, world)
assert sp.enable_p2p()
sp.ctx.set_option("p2p_deadline_ms", 300)
sp.ctx.set_option("p2p_first_deadline_ms", 300)
kw = dict(semantics="reference", n_steps=12, seed=9)
good = sp.price_american(40_000 * world, stream=1, **kw)   # one good pricing: the mailboxes work
if rank == 1:
    time.sleep(2.0)                                          # SLOW, not dead: 1.7 s past rank 0's deadline, then it joins
try:
    out = sp.price_american(40_000 * world, stream=2, **kw)
    print("RESULT price %%r" %% out["price"])
except _ffi.OmcError as e:
    print("RESULT error: %%s" %% e)
print("STATUS", sp.ctx.p2p_status())
os._exit(0)
"""


def test_a_slow_peer_fails_the_exchange_on_every_rank(standin, tmp_path):
    """rank 1 is two seconds late for a pricing whose exchange deadline is 0.3 s -- alive, not
    lost.  Rank 0 gives up (error word 1) and from then on publishes poison instead of its epoch; rank 1, arriving late,
    must NOT assemble a finite price from rank 0's abandoned contributions: it reads the poison, sets its own error word
    (2) and fails too.  Both calls return 3100; neither returns a price."""
    script = tmp_path / "p2p_slow.py"
    script.write_text(_P2P_SLOW % ROOT)
    port = 29990 + os.getpid() % 9
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMC_RCCL_LIB=standin, OMC_RDZV_NONCE=f"slow{os.getpid()}",
                   OMC_STANDIN_TIMEOUT_S="30")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    lines, status = [], []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        lines.append([ln for ln in so.splitlines() if ln.startswith("RESULT")][0])
        status.append([ln for ln in so.splitlines() if ln.startswith("STATUS")][0])
    assert "error" in lines[0] and "timed out" in lines[0], lines
    assert "error" in lines[1] and "another rank gave up" in lines[1], lines  # the slow rank fails as well
    assert status[0].endswith(", 1)") and status[1].endswith(", 2)"), status


def test_config3_shard_size_through_the_native_communicator(ctx, standin):
    """BASELINE configs[2] is 64M paths over 8 GPUs = 8M-path shards.  Four such shards (the test box admits four rank
    processes on its card; 65 GB of path matrices incl. the overlapped sequence's second buffers) priced through the
    native-communicator path must equal the ONE-GPU pricing of the same 32M paths: global pair offsets, the
    decision-independent moment all-reduce and the result collective at config 3's per-GPU size."""
    from options_model_amd import _ffi
    out, took = _bench(["--gpus", "4", "--single-device", "--backend", "rccl", "--config", "c3", "--steps", "2", "--warmup",
                        "1", "--min-warmup-seconds", "0.05", "--no-variants", "--no-cpu-baseline", "--no-sustained"],
                       standin, timeout=600)
    d = _line(out)
    assert d["rccl_ranks"] == 4 and d["config"]["paths_per_gpu"] == 8_000_000 and d["seq_overlap"] == "on"
    ref = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=32_000_000, n_steps=252, seed=42, stream=1))
    assert d["price"] == pytest.approx(ref["price"], rel=1e-12)
    lp = d["last_pricing"]
    assert (lp["n_paths"], lp["n_exercised"], lp["sum_nitm"]) == (32_000_000, ref["n_exercised"], ref["sum_nitm"])
    assert d["price_check"]["rel_err"] < 1e-3
